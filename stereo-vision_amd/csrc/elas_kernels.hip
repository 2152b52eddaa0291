// HIP kernels (gfx950 / CDNA4) of the ELAS dense-disparity path.
//
// All of this is integer SAD / rank / stencil work bounded by memory traffic,
// not by arithmetic: no MFMA.  Design rules followed throughout:
//   * wavefront = 64 lanes; every per-pixel kernel maps lanes to consecutive u
//     so descriptor traffic is 16 B/lane = 1 KiB per wave instruction;
//   * 8-bit tiles are staged through LDS once and gathered from there;
//   * SAD uses v_sad_u8 (__builtin_amdgcn_sad_u8), arg-min over disparities is
//     a packed (cost<<16|d) wave reduction with __shfl_xor;
//   * float expressions that decide pixel ownership or d_plane are written with
//     __fmul_rn/__fadd_rn so they are never contracted into FMAs (the reference
//     is built -msse3 without FMA).
// Each kernel cites the reference lines whose result it reproduces.
#include <hip/hip_runtime.h>

#include "svh_internal.h"

namespace svh {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t other = (uint32_t)__shfl_xor((int)v, o, kWave);
        v = other < v ? other : v;
    }
    return v;
}

// sum |a.byte - b.byte| over 16 bytes (== psadbw lanes 0+4, elas.cpp:406-414)
__device__ __forceinline__ uint32_t sad16(const uint4& a, const uint4& b) {
    uint32_t s = __builtin_amdgcn_sad_u8(a.x, b.x, 0u);
    s = __builtin_amdgcn_sad_u8(a.y, b.y, s);
    s = __builtin_amdgcn_sad_u8(a.z, b.z, s);
    s = __builtin_amdgcn_sad_u8(a.w, b.w, s);
    return s;
}

// descriptor texture: sum |byte - 128| (elas.cpp:358-362, 851-855)
__device__ __forceinline__ uint32_t texture16(const uint4& a) {
    const uint4 mid = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    return sad16(a, mid);
}

__device__ __forceinline__ int32_t sat_u8(int32_t x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

// (uint32_t)f assigned to int32_t with x86 cvttss2si semantics (elas.cpp:1081-1082)
__device__ __forceinline__ int32_t f2u2i(float f) { return (int32_t)(uint32_t)(long long)f; }

// ---------------------------------------------------------------------------
// E1+E2  3x3 Sobel + 16-byte descriptor, fused
//   filter::sobel3x3          libelas/src/filter.cpp:408-416 (+372-405, 227-267, 176-222)
//   Descriptor::createDescriptor   libelas/src/descriptor.cpp:48-121
// One block = 64x16 pixel tile of one image.  The 8-bit tile (+3 halo) goes to
// LDS once, du/dv (+2 halo) are built in LDS, then every thread gathers its 16
// bytes and issues one 16-byte store; a wave writes 1 KiB contiguous.
// Border descriptors (and odd rows when half) are written as zero.
// ---------------------------------------------------------------------------
constexpr int TX = 64, TY = 16;

__global__ __launch_bounds__(256) void k_descriptor(const uint8_t* __restrict__ I1,
                                                    const uint8_t* __restrict__ I2, int pitch1,
                                                    int pitch2, int W, int H, int half,
                                                    uint8_t* __restrict__ desc1,
                                                    uint8_t* __restrict__ desc2) {
    __shared__ uint8_t sI[TY + 6][TX + 8];
    __shared__ uint8_t sU[TY + 4][TX + 4];
    __shared__ uint8_t sV[TY + 4][TX + 4];

    const uint8_t* I = blockIdx.z ? I2 : I1;
    const int pitch = blockIdx.z ? pitch2 : pitch1;
    uint8_t* desc = blockIdx.z ? desc2 : desc1;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int tid = threadIdx.y * TX + threadIdx.x;

    for (int i = tid; i < (TY + 6) * (TX + 6); i += 256) {
        int r = i / (TX + 6), c = i - r * (TX + 6);
        int y = y0 - 3 + r, x = x0 - 3 + c;
        uint8_t val = 0;
        if (x >= 0 && x < W && y >= 0 && y < H) val = I[(size_t)y * pitch + x];
        sI[r][c] = val;
    }
    __syncthreads();
    for (int i = tid; i < (TY + 4) * (TX + 4); i += 256) {
        int r = i / (TX + 4), c = i - r * (TX + 4);
        // (r,c) in sU is image (y0-2+r, x0-2+c) == sI[r+1][c+1]
        int a0 = sI[r][c], a1 = sI[r][c + 1], a2 = sI[r][c + 2];
        int b0 = sI[r + 1][c], b1 = sI[r + 1][c + 1], b2 = sI[r + 1][c + 2];
        int c0 = sI[r + 2][c], c1 = sI[r + 2][c + 1], c2 = sI[r + 2][c + 2];
        (void)b1;
        int Sl = a0 + 2 * b0 + c0, Sr = a2 + 2 * b2 + c2;          // vertical 1 2 1
        int Tl = a0 - c0, Tm = a1 - c1, Tr = a2 - c2;              // vertical 1 0 -1
        sU[r][c] = (uint8_t)sat_u8(((Sl - Sr) >> 2) + 128);        // horizontal 1 0 -1
        sV[r][c] = (uint8_t)sat_u8(((Tl + 2 * Tm + Tr) >> 2) + 128);  // horizontal 1 2 1
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= W) return;
    const int lx = threadIdx.x + 2;
#pragma unroll
    for (int k = 0; k < TY / 4; k++) {
        const int ty = threadIdx.y + 4 * k;
        const int y = y0 + ty;
        if (y >= H) break;
        uint4 out = make_uint4(0, 0, 0, 0);
        bool inside = x >= 3 && x < W - 3 && y >= 3 && y < H - 3;
        if (half) inside = inside && y >= 4 && (y & 1) == 0;
        if (inside) {
            const int ly = ty + 2;
            uint32_t b0 = sU[ly - 2][lx], b1 = sU[ly - 1][lx - 2], b2 = sU[ly - 1][lx],
                     b3 = sU[ly - 1][lx + 2];
            uint32_t b4 = sU[ly][lx - 1], b5 = sU[ly][lx], b7 = sU[ly][lx + 1];
            uint32_t b8 = sU[ly + 1][lx - 2], b9 = sU[ly + 1][lx], b10 = sU[ly + 1][lx + 2],
                     b11 = sU[ly + 2][lx];
            uint32_t b12 = sV[ly - 1][lx], b13 = sV[ly][lx - 1], b14 = sV[ly][lx + 1],
                     b15 = sV[ly + 1][lx];
            out.x = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            out.y = b4 | (b5 << 8) | (b5 << 16) | (b7 << 24);
            out.z = b8 | (b9 << 8) | (b10 << 16) | (b11 << 24);
            out.w = b12 | (b13 << 8) | (b14 << 16) | (b15 << 24);
        }
        *reinterpret_cast<uint4*>(desc + ((size_t)y * W + x) * 16) = out;
    }
}

// ---------------------------------------------------------------------------
// E3+E4  support candidate matching
//   Elas::computeMatchingDisparity   libelas/src/elas.cpp:322-445
//   Elas::computeSupportMatches      libelas/src/elas.cpp:449-493
// One wave per lattice candidate; lanes = disparities (up to 4 rounds of 64).
// Energy = SAD over the 4 descriptors at (+-2,+-2) (64 bytes).  The reference
// keeps best and second best while scanning d upwards with strict "<": that is
// the smallest and second smallest of the keys (E<<16 | d).
// ---------------------------------------------------------------------------
struct SupportParams {
    int W, H, Wc, Hc, step;
    int disp_min, disp_max, support_texture, lr_threshold;
    float support_threshold;
};

__device__ __forceinline__ int support_match(const uint8_t* __restrict__ own,
                                             const uint8_t* __restrict__ oth, int u, int v,
                                             bool right, const SupportParams& P, int lane) {
    if (!(u >= 5 && u <= P.W - 6 && v >= 5 && v <= P.H - 6)) return -1;
    const size_t row = (size_t)P.W;  // in 16-byte units
    const uint4* ownq = reinterpret_cast<const uint4*>(own);
    const uint4* othq = reinterpret_cast<const uint4*>(oth);
    const size_t centre = (size_t)v * row + u;
    if ((int)texture16(ownq[centre]) < P.support_texture) return -1;

    const int dmin = P.disp_min > 0 ? P.disp_min : 0;
    int dmax = right ? P.W - u - 5 : u - 5;
    dmax = dmax < P.disp_max ? dmax : P.disp_max;
    if (dmax - dmin < 10) return -1;

    const size_t o0 = centre - 2 * row - 2, o1 = centre - 2 * row + 2;
    const size_t o2 = centre + 2 * row - 2, o3 = centre + 2 * row + 2;
    const uint4 r0 = ownq[o0], r1 = ownq[o1], r2 = ownq[o2], r3 = ownq[o3];

    uint32_t best1 = 0xFFFFFFFFu, best2 = 0xFFFFFFFFu;
    for (int d0 = dmin; d0 <= dmax; d0 += kWave) {
        const int d = d0 + lane;
        if (d <= dmax) {
            const ptrdiff_t sh = right ? d : -d;
            uint32_t e = sad16(r0, othq[o0 + sh]);
            e += sad16(r1, othq[o1 + sh]);
            e += sad16(r2, othq[o2 + sh]);
            e += sad16(r3, othq[o3 + sh]);
            const uint32_t key = (e << 16) | (uint32_t)d;
            if (key < best1) {
                best2 = best1;
                best1 = key;
            } else if (key < best2) {
                best2 = key;
            }
        }
    }
    const uint32_t m1 = wave_min_u32(best1);
    const uint32_t m2 = wave_min_u32(best1 == m1 ? best2 : best1);
    if (m1 == 0xFFFFFFFFu || m2 == 0xFFFFFFFFu) return -1;
    const float e1 = (float)(m1 >> 16), e2 = (float)(m2 >> 16);
    if (e1 < __fmul_rn(P.support_threshold, e2)) return (int)(m1 & 0xFFFFu);
    return -1;
}

__global__ __launch_bounds__(256) void k_support(const uint8_t* __restrict__ desc1,
                                                 const uint8_t* __restrict__ desc2,
                                                 int16_t* __restrict__ dcan, SupportParams P) {
    const int cand = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    const int lane = threadIdx.x & 63;
    if (cand >= P.Wc * P.Hc) return;
    const int vc = cand / P.Wc, uc = cand - vc * P.Wc;
    int out = 0;  // row 0 / column 0 stay at calloc's 0 (elas.cpp:464, 471-479)
    if (uc > 0 && vc > 0) {
        out = -1;
        const int u = uc * P.step, v = vc * P.step;
        const int d = support_match(desc1, desc2, u, v, false, P, lane);
        if (d >= 0) {
            const int d2 = support_match(desc2, desc1, u - d, v, true, P, lane);
            const int diff = d > d2 ? d - d2 : d2 - d;
            if (d2 >= 0 && diff <= P.lr_threshold) out = d;
        }
    }
    if (lane == 0) dcan[cand] = (int16_t)out;
}

// ---------------------------------------------------------------------------
// E10 (ownership)  Elas::computeDisparity triangle rasterisation
//   libelas/src/elas.cpp:1003-1115
// The reference walks triangles in list order and lets a later triangle
// overwrite an earlier one on the few pixels both cover; findMatch's early-outs
// depend on the pixel only, so "owner = highest triangle index covering the
// pixel" is exact.  One wave per triangle, lanes = columns.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_owner(const TriRaster* __restrict__ r1, int n1,
                                               const TriRaster* __restrict__ r2, int n2, int W,
                                               int H, int sub, int32_t* __restrict__ owner1,
                                               int32_t* __restrict__ owner2) {
    const int side = blockIdx.y;
    const TriRaster* r = side ? r2 : r1;
    const int n = side ? n2 : n1;
    int32_t* owner = side ? owner2 : owner1;
    const int t = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    const int lane = threadIdx.x & 63;
    if (t >= n) return;
    const TriRaster tr = r[t];
#pragma unroll
    for (int part = 0; part < 2; part++) {
        const int lo = part ? tr.uB : tr.uA, hi = part ? tr.uC : tr.uB;
        if (lo == hi) continue;
        const float ea = part ? tr.BCa : tr.ABa, eb = part ? tr.BCb : tr.ABb;
        const int ulo = lo > 0 ? lo : 0, uhi = hi < W ? hi : W;
        for (int u = ulo + lane; u < uhi; u += kWave) {
            if (sub && (u & 1)) continue;
            const float fu = (float)u;
            const int v1 = f2u2i(__fadd_rn(__fmul_rn(tr.ACa, fu), tr.ACb));
            const int v2 = f2u2i(__fadd_rn(__fmul_rn(ea, fu), eb));
            int va = v1 < v2 ? v1 : v2, vb = v1 < v2 ? v2 : v1;
            va = va > 0 ? va : 0;
            vb = vb < H ? vb : H;
            for (int v = va; v < vb; v++) {
                if (sub && (v & 1)) continue;
                atomicMax(&owner[(size_t)v * W + u], t);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// E11  Elas::findMatch + updatePosteriorMinimum   libelas/src/elas.cpp:784-955
// One thread per disparity-map pixel; lanes are consecutive u, so for a common
// candidate disparity the 64 descriptor loads of a wave are one contiguous
// 1 KiB segment of the other image's descriptor row.
// ---------------------------------------------------------------------------
struct MatchParams {
    int W, H, DW, DH, gw, grid_size, sub;
    int disp_max, match_texture, plane_radius;
};

__global__ __launch_bounds__(256) void k_match(MatchArgs a, MatchParams P) {
    const int side = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= P.DW || y >= P.DH) return;
    const int u = P.sub ? 2 * x : x, v = P.sub ? 2 * y : y;
    float out = -10.f;
    const int t = a.owner[side][(size_t)v * P.W + u];
    if (t >= 0 && u >= 2 && u < P.W - 2) {
        int line = v < P.H - 3 ? v : P.H - 3;
        line = line > 2 ? line : 2;
        const uint4* own_line = reinterpret_cast<const uint4*>(a.desc[side]) + (size_t)line * P.W;
        const uint4* oth_line = reinterpret_cast<const uint4*>(a.desc[1 - side]) + (size_t)line * P.W;
        const uint4 own = own_line[u];
        if ((int)texture16(own) >= P.match_texture) {
            const TriRaster* tr = a.raster[side] + t;
            const float pa = tr->pa, pb = tr->pb, pc = tr->pc;
            const int valid = tr->valid;
            const int d_plane = (int)__fadd_rn(
                __fadd_rn(__fmul_rn(pa, (float)u), __fmul_rn(pb, (float)v)), pc);
            int dlo = d_plane - P.plane_radius;
            dlo = dlo > 0 ? dlo : 0;
            int dhi = d_plane + P.plane_radius;
            dhi = dhi < P.disp_max ? dhi : P.disp_max;
            const int cell = (v / P.grid_size) * P.gw + u / P.grid_size;
            const int cb = a.cell_off[side][cell], ce = a.cell_off[side][cell + 1];
            const uint16_t* cd = a.cell_d[side];
            int min_val = 10000, min_d = -1;
            for (int i = cb; i < ce; i++) {
                const int dc = cd[i];
                if (dc < dlo || dc > dhi) {
                    const int uw = side ? u + dc : u - dc;
                    if (uw < 2 || uw >= P.W - 2) continue;
                    const int val = (int)sad16(own, oth_line[uw]);
                    if (val < min_val) {
                        min_val = val;
                        min_d = dc;
                    }
                }
            }
            for (int dc = dlo; dc <= dhi; dc++) {
                const int uw = side ? u + dc : u - dc;
                if (uw < 2 || uw >= P.W - 2) continue;
                int dd = dc - d_plane;
                dd = dd < 0 ? -dd : dd;
                const int val = (int)sad16(own, oth_line[uw]) + (valid ? a.P[dd] : 0);
                if (val < min_val) {
                    min_val = val;
                    min_d = dc;
                }
            }
            out = min_d >= 0 ? (float)min_d : -1.f;
        }
    }
    a.D[side][(size_t)y * P.DW + x] = out;
}

// ---------------------------------------------------------------------------
// E12  Elas::leftRightConsistencyCheck   libelas/src/elas.cpp:1122-1204
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lr(const float* __restrict__ R1,
                                            const float* __restrict__ R2, float* __restrict__ D1,
                                            float* __restrict__ D2, int DW, int DH, int sub,
                                            float lr_threshold) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= DW || y >= DH) return;
    const size_t rowo = (size_t)y * DW;
    const float d1 = R1[rowo + x], d2 = R2[rowo + x];
    const float fx = (float)x;
    const float uw1 = sub ? fx - d1 / 2 : fx - d1;
    const float uw2 = sub ? fx + d2 / 2 : fx + d2;
    float o1 = -10.f, o2 = -10.f;
    if (d1 >= 0 && uw1 >= 0 && uw1 < (float)DW)
        if (!(fabsf(R2[rowo + (int)uw1] - d1) > lr_threshold)) o1 = d1;
    if (d2 >= 0 && uw2 >= 0 && uw2 < (float)DW)
        if (!(fabsf(R1[rowo + (int)uw2] - d2) > lr_threshold)) o2 = d2;
    D1[rowo + x] = o1;
    D2[rowo + x] = o2;
}

// ---------------------------------------------------------------------------
// E13  Elas::removeSmallSegments   libelas/src/elas.cpp:1208-1326
// The flood fill joins 4-neighbours that are both valid and differ by at most
// speckle_sim_threshold: a symmetric relation, so segments are the connected
// components of that graph and a parallel union-find gives the same sets.
//
// Run-based labelling keeps the number of atomic unions near the number of
// run-to-run contacts instead of two per pixel:
//   k_seg_runs  : every wave (64 consecutive pixels of one row) cuts its span
//                 into horizontally connected runs with one ballot; L[p] = index
//                 of the run's first pixel, RL[first] = run length
//   k_seg_link  : unions between runs -- across wave borders, and to the row
//                 above, skipping a vertical contact when the pixel to the left
//                 already makes the same union
//   k_seg_count : one atomicAdd of the run length per run; roots compressed
//   k_seg_mask  : pixels of components below speckle_size become -10
// ---------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int32_t* L, int x) {
    int p = L[x];
    while (p != x) {
        x = p;
        p = L[x];
    }
    return x;
}

__device__ __forceinline__ void uf_union(int32_t* L, int a, int b) {
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a > b) {
            int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[b], a);  // hang the larger root under the smaller
        if (old == b) return;
        b = old;
    }
}

__device__ __forceinline__ bool seg_joined(float a, float b, float thr) {
    return a >= 0 && b >= 0 && fabsf(a - b) <= thr;
}

__global__ __launch_bounds__(256) void k_seg_runs(const float* __restrict__ D,
                                                  int32_t* __restrict__ L,
                                                  int32_t* __restrict__ RL,
                                                  int32_t* __restrict__ cnt, int DW, int DH,
                                                  float thr) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;   // wave-uniform
    if (y >= DH) return;
    const int lane = threadIdx.x;
    const int i = y * DW + x;
    const bool inside = x < DW;
    const float d = inside ? D[i] : -10.f;
    const bool valid = d >= 0;
    float dl = __shfl_up(d, 1, kWave);
    if (lane == 0) dl = (x > 0 && inside) ? D[i - 1] : -10.f;
    const bool cl = seg_joined(d, dl, thr);
    const bool start = valid && (lane == 0 || !cl);
    const unsigned long long starts = __ballot(start);
    const unsigned long long invalid = __ballot(!valid);
    if (!inside) return;
    int label = -1, len = 0;
    if (valid) {
        const unsigned long long upto = starts & (~0ull >> (63 - lane));
        const int first = 63 - __clzll((long long)upto);  // a start at or before me always exists
        label = i - lane + first;
        if (start) {
            const unsigned long long stops = ((starts | invalid) >> lane) >> 1;
            len = stops ? __ffsll((long long)stops) : (kWave - lane);
        }
    }
    L[i] = label;
    RL[i] = len;
    cnt[i] = 0;
}

__global__ __launch_bounds__(256) void k_seg_link(const float* __restrict__ D,
                                                  int32_t* __restrict__ L, int DW, int DH,
                                                  float thr) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= DW || y >= DH) return;
    const int i = y * DW + x;
    const float d = D[i];
    if (!(d >= 0)) return;
    const float dl = x > 0 ? D[i - 1] : -10.f;
    const bool cl = seg_joined(d, dl, thr);
    // a run that continues across the wave border
    if (threadIdx.x == 0 && cl) uf_union(L, i, L[i - 1]);
    if (y > 0) {
        const float du = D[i - DW];
        if (seg_joined(d, du, thr)) {
            // p-1 ~ p, p-1 ~ q-1 and q-1 ~ q already put p and q in one set
            bool redundant = false;
            if (cl) {
                const float dul = D[i - DW - 1];
                redundant = seg_joined(dl, dul, thr) && seg_joined(du, dul, thr);
            }
            if (!redundant) uf_union(L, L[i], L[i - DW]);
        }
    }
}

__global__ __launch_bounds__(256) void k_seg_count(int32_t* __restrict__ L,
                                                   const int32_t* __restrict__ RL,
                                                   int32_t* __restrict__ cnt, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int len = RL[i];
    if (len > 0) {
        const int root = uf_find(L, i);
        L[i] = root;
        atomicAdd(&cnt[root], len);
    }
}

__global__ __launch_bounds__(256) void k_seg_mask(float* __restrict__ D,
                                                  const int32_t* __restrict__ L,
                                                  const int32_t* __restrict__ cnt, int n,
                                                  int min_size) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = L[i];
    if (s < 0) {
        // an invalid pixel is a segment of one pixel (elas.cpp:1244-1317)
        if (1 < min_size) D[i] = -10.f;
    } else if (cnt[L[s]] < min_size) {
        D[i] = -10.f;
    }
}

// ---------------------------------------------------------------------------
// E14  Elas::gapInterpolation   libelas/src/elas.cpp:1330-1530
// A run of invalid pixels is filled iff its length is in [1,gap] and it has a
// valid pixel on both sides inside the line.  Both ends are original valid
// pixels, so every line is a pure function of its input: the row pass and the
// column pass are two out-of-place per-pixel kernels that look at most `gap`
// pixels each way.  (Sequential per-line kernels below handle large gaps and the
// add_corners extrapolation.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float gap_value(float d1, float d2) {
    return fabsf(d1 - d2) < 3.0f ? (d1 + d2) / 2 : (d1 < d2 ? d1 : d2);  // min(d1,d2): std::min
}

template <bool kCols>
__global__ __launch_bounds__(256) void k_gap_local(const float* __restrict__ in,
                                                   float* __restrict__ out, int DW, int DH,
                                                   int gap) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= DW || y >= DH) return;
    const int i = y * DW + x;
    const int stride = kCols ? DW : 1;
    const int pos = kCols ? y : x, len = kCols ? DH : DW;
    float val = in[i];
    if (!(val >= 0)) {
        int l = 0, r = 0;
        for (int k = 1; k <= gap && pos - k >= 0; k++)
            if (in[i - k * stride] >= 0) {
                l = k;
                break;
            }
        if (l) {
            for (int k = 1; k <= gap - l + 1 && pos + k < len; k++)
                if (in[i + k * stride] >= 0) {
                    r = k;
                    break;
                }
            if (r) val = gap_value(in[i - l * stride], in[i + r * stride]);
        }
    }
    out[i] = val;
}

// general per-line version (any gap width, add_corners extrapolation), in place
template <bool kCols>
__global__ void k_gap_lines(float* __restrict__ D, int DW, int DH, int gap, int add_corners) {
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    const int nlines = kCols ? DW : DH;
    if (line >= nlines) return;
    const int len = kCols ? DH : DW;
    const int stride = kCols ? DW : 1;
    float* base = D + (kCols ? line : (size_t)line * DW);
    int count = 0;
    for (int p = 0; p < len; p++) {
        if (base[(size_t)p * stride] >= 0) {
            if (count >= 1 && count <= gap) {
                const int first = p - count, last = p - 1;
                if (first > 0 && last < len - 1) {
                    const float di = gap_value(base[(size_t)(first - 1) * stride],
                                               base[(size_t)(last + 1) * stride]);
                    for (int q = first; q <= last; q++) base[(size_t)q * stride] = di;
                }
            }
            count = 0;
        } else {
            count++;
        }
    }
    if (add_corners) {
        for (int p = 0; p < len; p++)
            if (base[(size_t)p * stride] >= 0) {
                const float val = base[(size_t)p * stride];
                for (int q = (p - gap > 0 ? p - gap : 0); q < p; q++) base[(size_t)q * stride] = val;
                break;
            }
        for (int p = len - 1; p >= 0; p--)
            if (base[(size_t)p * stride] >= 0) {
                const float val = base[(size_t)p * stride];
                const int qe = p + gap < len - 1 ? p + gap : len - 1;
                for (int q = p; q <= qe; q++) base[(size_t)q * stride] = val;
                break;
            }
    }
}

// ---------------------------------------------------------------------------
// E15  Elas::adaptiveMean   libelas/src/elas.cpp:1535-1754
// weight = max(0, 4 - float_from_bits(bits(val - centre) & 0x4F000000)): the
// reference's "abs mask" is an int->float conversion of 0x7FFFFFFF
// (elas.cpp:1571).  Lanes of the SSE registers are ring slots (position % taps);
// the 8-tap branch first adds slot j and j+4, then sums ((l0+l1)+l2)+l3.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float am_weight(float val, float centre) {
    const float m = __uint_as_float(__float_as_uint(__fsub_rn(val, centre)) & 0x4F000000u);
    const float w = __fsub_rn(4.0f, m);
    return w > 0.0f ? w : 0.0f;
}

template <bool kCols, int kTaps>
__global__ __launch_bounds__(256) void k_adaptive_mean(const float* in, const float* keep,
                                                       float* out, int DW, int DH) {
    // in   : filter input (negative values read as -10)
    // keep : value written where the filter does not fire
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= DW || y >= DH) return;
    const int i = y * DW + x;
    constexpr int back = kTaps == 8 ? 3 : 1;      // centre = newest - back
    constexpr int lead = kTaps - 1;
    const int stride = kCols ? DW : 1;
    const int pos = kCols ? y : x, len = kCols ? DH : DW;
    const int other = kCols ? x : y, olen = kCols ? DW : DH;
    float res = keep[i];
    if (!kCols && res < 0) res = -10.f;  // D_copy/D_tmp initialisation (elas.cpp:1553-1560)
    // lines 3..olen-4; centres lead-back .. len-1-back
    if (other >= 3 && other < olen - 3 && pos >= lead - back && pos <= len - 1 - back) {
        const int first = pos + back - lead;  // oldest tap position
        float ring[kTaps];
#pragma unroll
        for (int k = 0; k < kTaps; k++) {
            const int p = first + k;
            float t = in[i + (p - pos) * stride];
            if (!kCols && t < 0) t = -10.f;
            ring[k] = t;
        }
        float centre = in[i];
        if (!kCols && centre < 0) centre = -10.f;
        // slot s holds the tap whose position % kTaps == s
        float wl[4], fl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            wl[j] = 0.f;
            fl[j] = 0.f;
        }
        float ws[kTaps], fs[kTaps];
#pragma unroll
        for (int s = 0; s < kTaps; s++) {
            const int k = ((s - first) % kTaps + kTaps) % kTaps;  // tap index in ring[]
            float t = ring[0];
#pragma unroll
            for (int q = 1; q < kTaps; q++) t = (k == q) ? ring[q] : t;
            const float w = am_weight(t, centre);
            ws[s] = w;
            fs[s] = __fmul_rn(t, w);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (kTaps == 8) {
                wl[j] = __fadd_rn(ws[j], ws[j + 4]);
                fl[j] = __fadd_rn(fs[j], fs[j + 4]);
            } else {
                wl[j] = ws[j];
                fl[j] = fs[j];
            }
        }
        const float wsum = __fadd_rn(__fadd_rn(__fadd_rn(wl[0], wl[1]), wl[2]), wl[3]);
        const float fsum = __fadd_rn(__fadd_rn(__fadd_rn(fl[0], fl[1]), fl[2]), fl[3]);
        if (wsum > 0) {
            const float dv = __fdiv_rn(fsum, wsum);
            if (dv >= 0) res = dv;
        }
    }
    out[i] = res;
}

// ---------------------------------------------------------------------------
// E16  Elas::median   libelas/src/elas.cpp:1758-1838 (separable 7-tap)
// ---------------------------------------------------------------------------
template <bool kCols>
__global__ __launch_bounds__(256) void k_median(const float* gate, const float* in, float* out,
                                                int DW, int DH) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= DW || y >= DH) return;
    const int i = y * DW + x;
    const bool interior = x >= 3 && x < DW - 3 && y >= 3 && y < DH - 3;
    // first pass writes D_temp (calloc'ed: 0 outside the interior); second pass
    // leaves D untouched outside the interior
    float res = kCols ? gate[i] : 0.f;
    if (interior) {
        if (gate[i] >= 0) {
            const int stride = kCols ? DW : 1;
            float v[7];
#pragma unroll
            for (int k = 0; k < 7; k++) v[k] = in[i + (k - 3) * stride];
            // insertion sort of 7 values
            for (int a = 1; a < 7; a++) {
                float key = v[a];
                int b = a - 1;
                while (b >= 0 && v[b] > key) {
                    v[b + 1] = v[b];
                    b--;
                }
                v[b + 1] = key;
            }
            res = v[3];
        } else {
            res = gate[i];
        }
    }
    out[i] = res;
}

inline dim3 grid2d(int w, int h, int z = 1) { return dim3((w + 63) / 64, (h + 3) / 4, z); }

// brackets one kernel launch with the context's timer, if any
struct Timed {
    Profiler* p;
    Timed(const LaunchCtx& cx, const char* name) : p(cx.prof) {
        if (p) p->begin(name);
    }
    ~Timed() {
        if (p) p->end();
    }
};

#define LAUNCH(name, kernel, grid, block, ...)                                                  \
    do {                                                                                        \
        Timed timed_(cx, name);                                                                 \
        hipLaunchKernelGGL(kernel, grid, block, 0, (hipStream_t)cx.stream, __VA_ARGS__);        \
    } while (0)

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_descriptor(const LaunchCtx& cx, const DevImages& img, int32_t W, int32_t H,
                       int32_t half, uint8_t* desc1, uint8_t* desc2) {
    dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY, 2), block(TX, 4);
    LAUNCH("k_descriptor", k_descriptor, grid, block, img.I[0], img.I[1], img.pitch[0],
           img.pitch[1], W, H, half, desc1, desc2);
}

void launch_support(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d,
                    const uint8_t* desc1, const uint8_t* desc2, int16_t* dcan) {
    SupportParams P;
    P.W = d.W; P.H = d.H; P.Wc = d.Wc; P.Hc = d.Hc; P.step = d.step;
    P.disp_min = p.disp_min; P.disp_max = p.disp_max;
    P.support_texture = p.support_texture; P.lr_threshold = p.lr_threshold;
    P.support_threshold = p.support_threshold;
    const int cands = d.Wc * d.Hc;
    LAUNCH("k_support", k_support, dim3((cands + 3) / 4), dim3(256), desc1, desc2, dcan, P);
}

void launch_owner(const LaunchCtx& cx, const Dims& d, const TriRaster* r1, int32_t n1,
                  const TriRaster* r2, int32_t n2, int32_t subsampling, int32_t* owner1,
                  int32_t* owner2) {
    hipStream_t s = (hipStream_t)cx.stream;
    const size_t bytes = (size_t)d.W * d.H * sizeof(int32_t);
    if (owner2 == owner1 + (size_t)d.W * d.H) {
        (void)hipMemsetAsync(owner1, 0xFF, 2 * bytes, s);
    } else {
        (void)hipMemsetAsync(owner1, 0xFF, bytes, s);
        (void)hipMemsetAsync(owner2, 0xFF, bytes, s);
    }
    const int nmax = n1 > n2 ? n1 : n2;
    if (nmax == 0) return;
    LAUNCH("k_owner", k_owner, dim3((nmax + 3) / 4, 2), dim3(256), r1, n1, r2, n2, d.W, d.H,
           subsampling, owner1, owner2);
}

void launch_match(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d,
                  const MatchArgs& a) {
    MatchParams P;
    P.W = d.W; P.H = d.H; P.DW = d.DW; P.DH = d.DH; P.gw = d.gw; P.grid_size = p.grid_size;
    P.sub = p.subsampling; P.disp_max = p.disp_max; P.match_texture = p.match_texture;
    P.plane_radius = a.plane_radius;
    LAUNCH("k_match", k_match, grid2d(d.DW, d.DH, 2), dim3(64, 4), a, P);
}

void launch_lr(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, const float* D1raw,
               const float* D2raw, float* D1, float* D2) {
    LAUNCH("k_lr", k_lr, grid2d(d.DW, d.DH), dim3(64, 4), D1raw, D2raw, D1, D2, d.DW, d.DH,
           p.subsampling, (float)p.lr_threshold);
}

void launch_segments(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, float* D,
                     int32_t* labels, int32_t* runlen, int32_t* counts) {
    const int n = d.DW * d.DH;
    int min_size = p.speckle_size;
    if (p.subsampling) min_size = (int)(sqrtf((float)p.speckle_size) * 2);  // elas.cpp:1218
    const dim3 lin((n + 255) / 256), b256(256), g2 = grid2d(d.DW, d.DH), b2(64, 4);
    LAUNCH("k_seg_runs", k_seg_runs, g2, b2, D, labels, runlen, counts, d.DW, d.DH,
           p.speckle_sim_threshold);
    LAUNCH("k_seg_link", k_seg_link, g2, b2, D, labels, d.DW, d.DH, p.speckle_sim_threshold);
    LAUNCH("k_seg_count", k_seg_count, lin, b256, labels, runlen, counts, n);
    LAUNCH("k_seg_mask", k_seg_mask, lin, b256, D, labels, counts, n, min_size);
}

void launch_gap(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, float* D,
                float* tmp) {
    int gap = p.ipol_gap_width;
    if (p.subsampling) gap = p.ipol_gap_width / 2 + 1;  // elas.cpp:1340
    if (gap <= 16 && !p.add_corners) {
        LAUNCH("k_gap_rows", k_gap_local<false>, grid2d(d.DW, d.DH), dim3(64, 4), D, tmp, d.DW, d.DH,
               gap);
        LAUNCH("k_gap_cols", k_gap_local<true>, grid2d(d.DW, d.DH), dim3(64, 4), tmp, D, d.DW, d.DH,
               gap);
    } else {
        LAUNCH("k_gap_rows_seq", k_gap_lines<false>, dim3((d.DH + 63) / 64), dim3(64), D, d.DW, d.DH,
               gap, p.add_corners);
        LAUNCH("k_gap_cols_seq", k_gap_lines<true>, dim3((d.DW + 63) / 64), dim3(64), D, d.DW, d.DH,
               gap, p.add_corners);
    }
}

void launch_adaptive_mean(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, float* D,
                          float* tmp) {
    const dim3 g = grid2d(d.DW, d.DH), b(64, 4);
    if (p.subsampling) {
        LAUNCH("k_mean_h", (k_adaptive_mean<false, 4>), g, b, D, D, tmp, d.DW, d.DH);
        LAUNCH("k_mean_v", (k_adaptive_mean<true, 4>), g, b, tmp, D, D, d.DW, d.DH);
    } else {
        LAUNCH("k_mean_h", (k_adaptive_mean<false, 8>), g, b, D, D, tmp, d.DW, d.DH);
        LAUNCH("k_mean_v", (k_adaptive_mean<true, 8>), g, b, tmp, D, D, d.DW, d.DH);
    }
}

void launch_median(const LaunchCtx& cx, const Dims& d, float* D, float* tmp) {
    const dim3 g = grid2d(d.DW, d.DH), b(64, 4);
    // horizontal pass D -> tmp, vertical pass gates on D, reads tmp, writes D
    LAUNCH("k_median_h", k_median<false>, g, b, D, D, tmp, d.DW, d.DH);
    LAUNCH("k_median_v", k_median<true>, g, b, D, tmp, D, d.DW, d.DH);
}

}  // namespace svh
