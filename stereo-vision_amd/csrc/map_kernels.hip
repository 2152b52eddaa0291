// Map fusion on the device (include/svh_map.h): 3-D reprojection of D1 and the
// frame-to-frame association of stereomapper, kernels + the small host engine around them.
//   StereoThread::createCurrentMap                  stereomapper/stereothread.cpp:180-255
//   StereoThread::addDisparityMapToReconstruction   stereomapper/stereothread.cpp:290-437
//
// What is sequential in the reference and how it is kept:
//  * the association scans the previous map column by column and READ-MODIFY-WRITES the current
//    map at the projected pixel, so several previous points landing on one pixel see each
//    other's effect in scan order.  Here every previous point first files itself under its target
//    pixel (one atomicExch per point: per-target linked lists), then one thread per target pixel
//    replays its list in scan order -- different targets never interact;
//  * the two point lists are push_back'ed in scan order (columns left to right, top to bottom):
//    ordered stream compaction over that order (count per 1024 elements, scan, scatter).
// fp32 with IEEE division and no contraction (the file is compiled with -ffp-contract=off): the
// reference's float expressions operation by operation; the double-typed sub-expressions
// (x / 255.0, the gain ramp, (a + b) / 2.0) are evaluated in double and narrowed once.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/matrix.h"
#include "../../include/svh.h"
#include "../../include/svh_map.h"
#include "svh_config.h"

namespace svh {
int fail(int code, const std::string& msg);   // elas_engine.cpp: sets svh_last_error()
bool fi_armed();                                    // elas_engine.cpp: fault injection (svh_internal.h)
bool fi_hit(const char* expr_text);
void report_hip_failure(const char* entry);
}

namespace {

struct MapCoef {
    float hcf[12];      // rows 0..2 of H_total
    float hfc[4];       // row 2 of inv(H_total)
    float pfc[12];      // K * inv(H_total)[0:3, 0:4]
    float f, cu, cv, base, max_dist;
    float gain_inv;
    int32_t margin;
};

struct Planes {
    float *I, *D, *X, *Y, *Z;
};

// (int32_t)float as x86's cvttss2si does it: out of range / NaN -> INT_MIN
__device__ __forceinline__ int32_t f2i_x86(float f) {
    return (f >= -2147483648.f && f < 2147483648.f) ? (int32_t)f : (int32_t)0x80000000;
}

// ---- createCurrentMap ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_map_create(const float* __restrict__ D1, const uint8_t* __restrict__ I1,
                                                    int w, int h, int step, MapCoef c, Planes cur) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= w || v >= h) return;
    const int a = v * w + u;
    float I = (float)((double)(float)I1[(size_t)v * step + u] / 255.0);
    // gain ramp over the image border (:232-252): every pixel is touched by at most one (i, side)
    const int m = c.margin;
    int i = -1;
    if (u >= m && u < w - m) {
        if (v < m) i = v;
        else if (v >= h - m) i = h - 1 - v;
    } else if (v >= m && v < h - m) {
        if (u < m) i = u;
        else if (u >= w - m) i = w - 1 - u;
    }
    if (i >= 0) {
        const float g = (float)(((double)__fmul_rn((float)(m - i), c.gain_inv) + (double)(float)i * 1.0) / (double)(float)m);
        const float t = __fmul_rn(g, I);
        I = fminf(fmaxf(t, 0.f), 1.f);
    }
    float d = D1[a], X = 0.f, Y = 0.f, Z = 0.f;
    if (d > 0) {
        const float z = __fdiv_rn(__fmul_rn(c.f, c.base), d);
        if ((double)z > 0.1 && z < c.max_dist) {
            const float x = __fdiv_rn(__fmul_rn(__fsub_rn((float)u, c.cu), c.base), d);
            const float y = __fdiv_rn(__fmul_rn(__fsub_rn((float)v, c.cv), c.base), d);
            X = c.hcf[0] * x + c.hcf[1] * y + c.hcf[2] * z + c.hcf[3];
            Y = c.hcf[4] * x + c.hcf[5] * y + c.hcf[6] * z + c.hcf[7];
            Z = c.hcf[8] * x + c.hcf[9] * y + c.hcf[10] * z + c.hcf[11];
        } else {
            d = -1.f;
        }
    }
    cur.I[a] = I;
    cur.D[a] = d;
    cur.X[a] = X;
    cur.Y[a] = Y;
    cur.Z[a] = Z;
}

// ---- association, step 1: every valid previous point finds its target pixel -----------------
// state: 0 = no point here, 1 = point that stays in the previous list, 2 = filed under a target
__global__ __launch_bounds__(256) void k_map_project(Planes prev, int pw, int ph, int cw, int chh, MapCoef c,
                                                     int32_t* __restrict__ head, int32_t* __restrict__ next,
                                                     uint8_t* __restrict__ state) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= pw * ph) return;
    uint8_t st = 0;
    if (prev.D[a] > 0) {
        st = 1;
        const float x = prev.X[a], y = prev.Y[a], z = prev.Z[a];
        const float z2 = c.hfc[0] * x + c.hfc[1] * y + c.hfc[2] * z + c.hfc[3];
        if ((double)z2 > 0.1 && z2 < c.max_dist) {
            const float w2 = c.pfc[8] * x + c.pfc[9] * y + c.pfc[10] * z + c.pfc[11];
            const int32_t u2 = f2i_x86(__fdiv_rn(c.pfc[0] * x + c.pfc[1] * y + c.pfc[2] * z + c.pfc[3], w2));
            const int32_t v2 = f2i_x86(__fdiv_rn(c.pfc[4] * x + c.pfc[5] * y + c.pfc[6] * z + c.pfc[7], w2));
            if (u2 >= 0 && u2 < cw && v2 >= 0 && v2 < chh) {
                st = 2;
                next[a] = atomicExch(&head[v2 * cw + u2], a);
            }
        }
    }
    state[a] = st;
}

// ---- association, step 2: one thread per target pixel replays its points in scan order -------
__global__ __launch_bounds__(256) void k_map_fuse(Planes prev, int pw, int ph, Planes cur, int cn,
                                                  const int32_t* __restrict__ head,
                                                  const int32_t* __restrict__ next, uint8_t* __restrict__ state) {
    const int a2 = blockIdx.x * 256 + threadIdx.x;
    if (a2 >= cn) return;
    int first = head[a2];
    if (first < 0) return;
    float D = cur.D[a2], X = cur.X[a2], Y = cur.Y[a2], Z = cur.Z[a2], I = cur.I[a2];
    // scan order of a previous pixel a = v * pw + u is u * ph + v
    long long last = -1;
    for (;;) {
        int pick = -1;
        long long best = 0x7FFFFFFFFFFFFFFFll;
        for (int s = first; s >= 0; s = next[s]) {
            const int v = s / pw, u = s - v * pw;
            const long long key = (long long)u * ph + v;
            if (key > last && key < best) {
                best = key;
                pick = s;
            }
        }
        if (pick < 0) break;
        last = best;
        const float x = prev.X[pick], y = prev.Y[pick], z = prev.Z[pick], pi = prev.I[pick];
        bool added = false;
        if (D > 0) {
            // fabs(float) + fabs(float) + fabs(float) < 0.2 (:355)
            const float dist = __fadd_rn(__fadd_rn(fabsf(__fsub_rn(x, X)), fabsf(__fsub_rn(y, Y))), fabsf(__fsub_rn(z, Z)));
            if ((double)dist < 0.2) {
                X = (float)((double)__fadd_rn(X, x) / 2.0);
                Y = (float)((double)__fadd_rn(Y, y) / 2.0);
                Z = (float)((double)__fadd_rn(Z, z) / 2.0);
                I = (float)((double)__fadd_rn(I, pi) / 2.0);
                added = true;
            }
        } else {
            X = x;
            Y = y;
            Z = z;
            I = pi;
            D = 1.f;
            added = true;
        }
        state[pick] = added ? 0 : 1;
    }
    cur.D[a2] = D;
    cur.X[a2] = X;
    cur.Y[a2] = Y;
    cur.Z[a2] = Z;
    cur.I[a2] = I;
}

// ---- ordered compaction over the scan order e = u * h + v -------------------------------------
// kFromState: element taken iff state == 1 (previous list); else iff D > 0 (current list)
template <bool kFromState>
__device__ __forceinline__ bool map_taken(const uint8_t* state, const float* D, int a) {
    return kFromState ? state[a] == 1 : D[a] > 0;
}

template <bool kFromState>
__global__ __launch_bounds__(256) void k_map_count(const uint8_t* __restrict__ state, Planes pl, int w, int h,
                                                   int32_t* __restrict__ blockcnt) {
    __shared__ int s_sum[4];
    const int n = w * h;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int e = blockIdx.x * 1024 + threadIdx.x * 4 + k;
        if (e < n) {
            const int u = e / h, v = e - u * h;
            mine += map_taken<kFromState>(state, pl.D, v * w + u) ? 1 : 0;
        }
    }
    for (int off = 32; off; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// exclusive scan of the block counts in place (one workgroup); total -> *total
__global__ __launch_bounds__(1024) void k_map_scan(int32_t* __restrict__ blockcnt, int nb, int64_t* __restrict__ total) {
    __shared__ int s[1024];
    int carry = 0;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int x = i < nb ? blockcnt[i] : 0;
        s[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nb) blockcnt[i] = carry + s[threadIdx.x] - x;
        const int chunk = s[1023];
        __syncthreads();
        carry += chunk;
    }
    if (threadIdx.x == 0) *total = carry;
}

template <bool kFromState>
__global__ __launch_bounds__(256) void k_map_scatter(const uint8_t* __restrict__ state, Planes pl, int w, int h,
                                                     const int32_t* __restrict__ blockoff,
                                                     float4* __restrict__ out) {
    __shared__ int s[256];
    const int n = w * h;
    int addr[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int e = blockIdx.x * 1024 + threadIdx.x * 4 + k;
        addr[k] = -1;
        if (e < n) {
            const int u = e / h, v = e - u * h;
            const int a = v * w + u;
            if (map_taken<kFromState>(state, pl.D, a)) {
                addr[k] = a;
                mine++;
            }
        }
    }
    s[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += add;
        __syncthreads();
    }
    int pos = blockoff[blockIdx.x] + s[threadIdx.x] - mine;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (addr[k] >= 0) out[pos++] = make_float4(pl.X[addr[k]], pl.Y[addr[k]], pl.Z[addr[k]], pl.I[addr[k]]);
}

// ---- colour-coded disparity (stereothread.cpp:117-147) -------------------------------------------
__global__ __launch_bounds__(256) void k_disp_color(const float* __restrict__ D, long long n, float* __restrict__ rgb) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float val = fminf(__fdiv_rn(D[i], 200.f), 1.0f);
    float r = 0.f, g = 0.f, b = 0.f;
    if (val > 0) {
        const float h2 = (float)(6.0 * (1.0 - (double)val));
        const float x = (float)(1.0 * (1.0 - fabs((double)fmodf(h2, 2.0f) - 1.0)));
        if (0 <= h2 && h2 < 1)       { r = 1; g = x; b = 0; }
        else if (1 <= h2 && h2 < 2)  { r = x; g = 1; b = 0; }
        else if (2 <= h2 && h2 < 3)  { r = 0; g = 1; b = x; }
        else if (3 <= h2 && h2 < 4)  { r = 0; g = x; b = 1; }
        else if (4 <= h2 && h2 < 5)  { r = x; g = 0; b = 1; }
        else if (5 <= h2 && h2 <= 6) { r = 1; g = 0; b = x; }
    }
    rgb[3 * i + 0] = r;
    rgb[3 * i + 1] = g;
    rgb[3 * i + 2] = b;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host engine
// ---------------------------------------------------------------------------------------------
struct svh_map {
    svh_map_params p{};
    int device = 0;
    hipStream_t stream = nullptr;
    int32_t w = 0, h = 0;            // geometry the buffers are sized for
    Planes pl[2] = {};               // [cur], [prev] planes, swapped every frame
    int cur = 0;
    bool have_prev = false;
    int32_t pw = 0, ph = 0;
    float* dD1 = nullptr;            // staged disparity map when the caller's is on the host
    uint8_t* dI1 = nullptr;
    uint8_t* h_stage = nullptr;      // pinned: packed I1 rows, then D1
    int32_t *head = nullptr, *next = nullptr, *blockcnt = nullptr;
    uint8_t* state = nullptr;
    float4* pts[2] = {nullptr, nullptr};
    int64_t* h_total = nullptr;      // pinned: point counts of the two lists
    int64_t npts[2] = {0, 0};

    void release() {
        for (int k = 0; k < 2; k++) {
            (void)hipFree(pl[k].I); (void)hipFree(pl[k].D); (void)hipFree(pl[k].X);
            (void)hipFree(pl[k].Y); (void)hipFree(pl[k].Z);
            pl[k] = Planes{};
            (void)hipFree(pts[k]);
            pts[k] = nullptr;
        }
        (void)hipFree(dD1); (void)hipFree(dI1); (void)hipFree(head); (void)hipFree(next);
        (void)hipFree(blockcnt); (void)hipFree(state);
        (void)hipHostFree(h_stage);
        dD1 = nullptr; dI1 = nullptr; head = next = blockcnt = nullptr; state = nullptr; h_stage = nullptr;
        w = h = 0;
    }
};

static int map_hip_failed(const char* expr, bool injected, hipError_t e) {
    const int rc = svh::fail(SVH_ERR_HIP, std::string(expr) + ": " + (injected ? "injected failure (SVH_TEST_FAIL_AT)" : hipGetErrorString(e)));
    svh::report_hip_failure("map");
    return rc;
}
#define MAP_TRY(expr)                                                                      \
    do {                                                                                   \
        const bool inj_ = svh::fi_armed() && svh::fi_hit(#expr);   /* svh_internal.h: fault injection */ \
        hipError_t e_ = inj_ ? hipErrorUnknown : (expr);                                   \
        if (e_ != hipSuccess) return map_hip_failed(#expr, inj_, e_);                      \
    } while (0)

static int32_t map_ensure(svh_map* m, int32_t w, int32_t h) {
    if (m->w == w && m->h == h) return SVH_OK;
    // a change of geometry starts a new reconstruction: the previous map cannot be addressed
    // with the new dimensions' buffers
    m->release();
    const size_t n = (size_t)w * h;
    for (int k = 0; k < 2; k++) {
        MAP_TRY(hipMalloc(&m->pl[k].I, n * 4)); MAP_TRY(hipMalloc(&m->pl[k].D, n * 4));
        MAP_TRY(hipMalloc(&m->pl[k].X, n * 4)); MAP_TRY(hipMalloc(&m->pl[k].Y, n * 4));
        MAP_TRY(hipMalloc(&m->pl[k].Z, n * 4));
        MAP_TRY(hipMalloc(&m->pts[k], n * sizeof(float4)));
    }
    MAP_TRY(hipMalloc(&m->dD1, n * 4));
    MAP_TRY(hipMalloc(&m->dI1, n));
    MAP_TRY(hipMalloc(&m->head, n * 4));
    MAP_TRY(hipMalloc(&m->next, n * 4));
    MAP_TRY(hipMalloc(&m->state, n));
    MAP_TRY(hipMalloc(&m->blockcnt, ((n + 1023) / 1024 + 1) * 4));
    MAP_TRY(hipHostMalloc(&m->h_stage, n * 5));
    m->w = w;
    m->h = h;
    m->have_prev = false;
    m->npts[0] = m->npts[1] = 0;
    return SVH_OK;
}

extern "C" {

svh_map* svh_map_create(const svh_map_params* p) {
    svh::ensure_init();
    if (!p) return nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        svh::fail(SVH_ERR_NO_DEVICE, "no HIP device visible: libsvhip has no CPU fallback");
        return nullptr;
    }
    svh_map* m = new svh_map();
    m->p = *p;
    (void)hipGetDevice(&m->device);
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc(&m->h_total, 2 * sizeof(int64_t)) != hipSuccess) {
        delete m;
        return nullptr;
    }
    return m;
}

void svh_map_destroy(svh_map* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    m->release();
    (void)hipHostFree(m->h_total);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

void svh_map_clear(svh_map* m) {
    if (!m) return;
    m->have_prev = false;
    m->npts[0] = m->npts[1] = 0;
}

int32_t svh_map_add(svh_map* m, const float* D1, int32_t d1_on_device, const uint8_t* I1, const int32_t* dims,
                    const double* H_total, float gain) {
    if (!m || !D1 || !I1 || !dims || !H_total) return svh::fail(SVH_ERR_BAD_ARG, "null argument");
    const int32_t w = dims[0], h = dims[1], step = dims[2];
    if (w < 1 || h < 1 || step < w || (int64_t)w * h > (1 << 28)) return svh::fail(SVH_ERR_BAD_ARG, "bad dimensions");
    MAP_TRY(hipSetDevice(m->device));
    int32_t rc = map_ensure(m, w, h);
    if (rc) return rc;
    hipStream_t s = m->stream;
    const size_t n = (size_t)w * h;
    // inputs: the image rows are packed on the way into pinned memory
    for (int32_t v = 0; v < h; v++) memcpy(m->h_stage + (size_t)v * w, I1 + (size_t)v * step, w);
    MAP_TRY(hipMemcpyAsync(m->dI1, m->h_stage, n, hipMemcpyHostToDevice, s));
    const float* dD = D1;
    if (!d1_on_device) {
        memcpy(m->h_stage + n, D1, n * 4);
        MAP_TRY(hipMemcpyAsync(m->dD1, m->h_stage + n, n * 4, hipMemcpyHostToDevice, s));
        dD = m->dD1;
    }
    // coefficients (stereothread.cpp:196-199, 306-314, 450-455) with the Matrix class of the boundary
    MapCoef c;
    {
        Matrix Ht(4, 4, H_total);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 4; j++) c.hcf[4 * i + j] = (float)Ht._val[i][j];
        Matrix Hi = Matrix::inv(Ht);
        const bool ok = Hi._m == 4;
        for (int j = 0; j < 4; j++) c.hfc[j] = ok ? (float)Hi._val[2][j] : 0.f;
        Matrix K(3, 3);
        K._val[0][0] = m->p.f; K._val[1][1] = m->p.f; K._val[0][2] = m->p.cu; K._val[1][2] = m->p.cv; K._val[2][2] = 1;
        if (ok) {
            Matrix top(3, 4);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 4; j++) top._val[i][j] = Hi._val[i][j];
            Matrix P = K * top;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 4; j++) c.pfc[4 * i + j] = (float)P._val[i][j];
        } else {
            for (int i = 0; i < 12; i++) c.pfc[i] = 0.f;
        }
    }
    c.f = m->p.f; c.cu = m->p.cu; c.cv = m->p.cv; c.base = m->p.base; c.max_dist = m->p.max_dist;
    c.margin = std::min(std::min(200, w / 2), h / 2);
    c.gain_inv = 1;
    if (gain) c.gain_inv = 1.0 / gain;

    const Planes cur = m->pl[m->cur], prev = m->pl[1 - m->cur];
    hipLaunchKernelGGL(k_map_create, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, dD, m->dI1, w, h, w, c, cur);
    const int nb = (int)((n + 1023) / 1024);
    if (m->have_prev) {
        const int pn = m->pw * m->ph;   // == n: a geometry change resets the reconstruction
        MAP_TRY(hipMemsetAsync(m->head, 0xFF, n * 4, s));
        hipLaunchKernelGGL(k_map_project, dim3((pn + 255) / 256), dim3(256), 0, s, prev, m->pw, m->ph, w, h, c,
                           m->head, m->next, m->state);
        hipLaunchKernelGGL(k_map_fuse, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, prev, m->pw, m->ph, cur,
                           (int)n, m->head, m->next, m->state);
        const int pb = (pn + 1023) / 1024;
        hipLaunchKernelGGL(k_map_count<true>, dim3(pb), dim3(256), 0, s, m->state, prev, m->pw, m->ph, m->blockcnt);
        hipLaunchKernelGGL(k_map_scan, dim3(1), dim3(1024), 0, s, m->blockcnt, pb, m->h_total);
        hipLaunchKernelGGL(k_map_scatter<true>, dim3(pb), dim3(256), 0, s, m->state, prev, m->pw, m->ph, m->blockcnt,
                           m->pts[0]);
    } else {
        m->h_total[0] = 0;
    }
    hipLaunchKernelGGL(k_map_count<false>, dim3(nb), dim3(256), 0, s, m->state, cur, w, h, m->blockcnt);
    hipLaunchKernelGGL(k_map_scan, dim3(1), dim3(1024), 0, s, m->blockcnt, nb, m->h_total + 1);
    hipLaunchKernelGGL(k_map_scatter<false>, dim3(nb), dim3(256), 0, s, m->state, cur, w, h, m->blockcnt, m->pts[1]);
    MAP_TRY(hipStreamSynchronize(s));
    MAP_TRY(hipGetLastError());
    m->npts[0] = m->h_total[0];
    m->npts[1] = m->h_total[1];
    // the current map becomes the previous one (the intended ":432")
    m->cur = 1 - m->cur;
    m->pw = w;
    m->ph = h;
    m->have_prev = true;
    return SVH_OK;
}

int64_t svh_map_points(svh_map* m, int32_t which, float* xyzv, int64_t cap) {
    if (!m) return 0;
    const int k = which ? 1 : 0;
    const int64_t n = m->npts[k];
    if (xyzv && n > 0 && cap > 0) {
        (void)hipSetDevice(m->device);
        (void)hipMemcpy(xyzv, m->pts[k], (size_t)std::min(n, cap) * sizeof(float4), hipMemcpyDeviceToHost);
    }
    return n;
}

int32_t svh_disparity_colormap(const float* D, int32_t d_on_device, int64_t n, float* rgb) {
    if (!D || !rgb || n < 0) return svh::fail(SVH_ERR_BAD_ARG, "null argument");
    if (n == 0) return SVH_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
        return svh::fail(SVH_ERR_NO_DEVICE, "no HIP device visible: libsvhip has no CPU fallback");
    float *dD = nullptr, *dC = nullptr;
    struct FreeBoth {       // (the error returns below must not leak the two temporaries)
        float*& a;
        float*& b;
        ~FreeBoth() { (void)hipFree(a); (void)hipFree(b); }
    } free_both_{dD, dC};
    MAP_TRY(hipMalloc(&dC, (size_t)n * 12));
    const float* src = D;
    if (!d_on_device) {
        MAP_TRY(hipMalloc(&dD, (size_t)n * 4));
        MAP_TRY(hipMemcpy(dD, D, (size_t)n * 4, hipMemcpyHostToDevice));
        src = dD;
    }
    hipLaunchKernelGGL(k_disp_color, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, src, (long long)n, dC);
    const hipError_t e = hipMemcpy(rgb, dC, (size_t)n * 12, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return svh::fail(SVH_ERR_HIP, std::string("colormap: ") + hipGetErrorString(e));
    return SVH_OK;
}

int32_t svh_map_planes(svh_map* m, float* out5, size_t cap_floats) {
    if (!m || !out5 || !m->have_prev) return svh::fail(SVH_ERR_BAD_ARG, "no map yet");
    const size_t n = (size_t)m->w * m->h;
    if (cap_floats < 5 * n) return svh::fail(SVH_ERR_BAD_ARG, "buffer too small");
    MAP_TRY(hipSetDevice(m->device));
    const Planes& p = m->pl[1 - m->cur];   // the map of the last frame
    float* src[5] = {p.I, p.D, p.X, p.Y, p.Z};
    for (int k = 0; k < 5; k++) MAP_TRY(hipMemcpy(out5 + k * n, src[k], n * 4, hipMemcpyDeviceToHost));
    return SVH_OK;
}

}  // extern "C"
