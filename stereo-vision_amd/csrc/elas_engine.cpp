// Engine behind the svh_elas_* C-ABI (include/svh.h).
//
// Shape of the pipeline for one stereo pair (reference: Elas::process,
// libelas/src/elas.cpp:32-170):
//
//   device phase A   Sobel+descriptor (both images, one launch), support
//                    candidate matching  ->  D_can (37 KB) to pinned host memory
//   host             in-place lattice filters, Delaunay x2, planes, grid, prior
//                    table  (~1 ms; serial in the reference as well)
//   device phase B   triangle ownership, dense matching (both sides, one
//                    launch), L/R check, speckle removal, gap interpolation,
//                    adaptive mean / median  ->  D1, D2
//
// A "lane" owns one HIP stream, its device buffers (sized once per image
// geometry and reused: callers construct an Elas per frame,
// stereomapper/stereothread.cpp:113) and its pinned staging.  A single
// svh_elas_process() borrows a lane on the calling thread; a batch spreads its
// pairs over all lanes with one host worker per lane, so the host section of one
// pair overlaps the device phases of the others.  No default-stream work, no
// process-global mutable state besides the lane pool (mutex protected), so
// objects may be used concurrently from different threads like the reference's
// (maindialog.cpp:456-465, 514-518).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "svh_internal.h"

namespace svh {

static thread_local std::string t_error;
static thread_local int t_device = 0;

static int fail(int code, const std::string& msg) {
    t_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(SVH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------
// per-kernel timing with HIP events recorded on the lane's own stream
// (svh_profile_*): totals are accumulated per kernel name across lanes.
// ---------------------------------------------------------------------------
struct KernelStat {
    double total_ms = 0;
    int64_t launches = 0;
};
static std::mutex g_prof_mu;
static std::map<std::string, KernelStat> g_prof;
static std::atomic<int> g_prof_on{0};

struct EventProfiler : Profiler {
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;          // 2 per launch
    std::vector<const char*> names;
    size_t used = 0;
    void begin(const char* kernel) override {
        if (ev.size() < 2 * (used + 1)) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            ev.push_back(a);
            ev.push_back(b);
            names.push_back(kernel);
        }
        names[used] = kernel;
        (void)hipEventRecord(ev[2 * used], stream);
    }
    void end() override {
        if (ev.size() < 2 * (used + 1)) return;
        (void)hipEventRecord(ev[2 * used + 1], stream);
        used++;
    }
    // call after the stream has been synchronised
    void collect() {
        if (!used) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        for (size_t i = 0; i < used; i++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess) {
                KernelStat& k = g_prof[names[i]];
                k.total_ms += ms;
                k.launches++;
            }
        }
        used = 0;
    }
};

// ---------------------------------------------------------------------------
// lane: stream + buffers
// ---------------------------------------------------------------------------
struct Lane {
    EventProfiler prof;
    int device = 0;
    hipStream_t stream = nullptr;
    // geometry the buffers were sized for
    int32_t W = 0, H = 0, disp_max = -1, step = 0, grid_size = 0, sub = -1;
    Dims d{};
    // device
    uint8_t* img[2] = {nullptr, nullptr};
    uint8_t* desc[2] = {nullptr, nullptr};
    int16_t* dcan = nullptr;
    int32_t* owner[2] = {nullptr, nullptr};
    uint8_t* prior_dev = nullptr;   // packed upload: rasters, cell lists, P
    float* Draw[2] = {nullptr, nullptr};
    float* D[2] = {nullptr, nullptr};
    float* tmp = nullptr;
    int32_t* labels = nullptr;
    int32_t* runlen = nullptr;
    int32_t* counts = nullptr;
    // pinned host
    int16_t* h_dcan = nullptr;
    uint8_t* h_img = nullptr;      // packed copy of both input images
    uint8_t* h_prior = nullptr;
    size_t prior_cap = 0;
    HostPrior hp;
    std::vector<int16_t> dcan_work;

    void release() {
        if (!stream) return;
        (void)hipSetDevice(device);
        for (int k = 0; k < 2; k++) {
            (void)hipFree(img[k]); (void)hipFree(desc[k]);
            (void)hipFree(Draw[k]); (void)hipFree(D[k]);
            img[k] = desc[k] = nullptr; owner[k] = nullptr; Draw[k] = D[k] = nullptr;
        }
        (void)hipFree(owner[0]);
        (void)hipFree(dcan); (void)hipFree(prior_dev); (void)hipFree(tmp); (void)hipFree(labels);
        (void)hipFree(counts); (void)hipFree(runlen);
        dcan = nullptr; prior_dev = nullptr; tmp = nullptr; labels = counts = runlen = nullptr;
        (void)hipHostFree(h_dcan); (void)hipHostFree(h_prior); (void)hipHostFree(h_img);
        h_dcan = nullptr; h_prior = nullptr; h_img = nullptr;
        W = H = 0;
    }

    int ensure(const svh_elas_params& p, int32_t w, int32_t h) {
        if (!stream) {
            HIP_TRY(hipSetDevice(device));
            HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        }
        const int32_t st = p.candidate_stepsize + (p.subsampling ? p.candidate_stepsize % 2 : 0);
        if (w == W && h == H && p.disp_max == disp_max && st == step && p.grid_size == grid_size &&
            p.subsampling == sub)
            return SVH_OK;
        hipStream_t keep = stream;
        release();
        stream = keep;
        d = make_dims(p, w, h);
        const size_t N = (size_t)w * h, DN = (size_t)d.DW * d.DH;
        for (int k = 0; k < 2; k++) {
            HIP_TRY(hipMalloc(&img[k], N));
            HIP_TRY(hipMalloc(&desc[k], N * 16));
            HIP_TRY(hipMalloc(&Draw[k], DN * sizeof(float)));
            HIP_TRY(hipMalloc(&D[k], DN * sizeof(float)));
        }
        HIP_TRY(hipMalloc(&owner[0], 2 * N * sizeof(int32_t)));   // both sides, one memset
        owner[1] = owner[0] + N;
        HIP_TRY(hipMalloc(&tmp, DN * sizeof(float)));
        HIP_TRY(hipMalloc(&labels, DN * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&counts, DN * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&runlen, DN * sizeof(int32_t)));
        const size_t nc = (size_t)d.Wc * d.Hc;
        HIP_TRY(hipMalloc(&dcan, nc * sizeof(int16_t)));
        HIP_TRY(hipHostMalloc(&h_dcan, nc * sizeof(int16_t)));
        HIP_TRY(hipHostMalloc(&h_img, 2 * N));
        // worst case prior: 2*(2n+8) triangles, full cell lists
        const size_t nsup = nc + 6, ntri = 2 * nsup + 8, cells = (size_t)d.gw * d.gh;
        prior_cap = 2 * (ntri * sizeof(TriRaster) + (cells + 1) * sizeof(int32_t) +
                         cells * (size_t)(p.disp_max + 1) * sizeof(uint16_t)) +
                    (size_t)(p.disp_max + 1) * sizeof(int32_t) + 256;
        HIP_TRY(hipMalloc(&prior_dev, prior_cap));
        HIP_TRY(hipHostMalloc(&h_prior, prior_cap));
        W = w; H = h; disp_max = p.disp_max; step = st; grid_size = p.grid_size; sub = p.subsampling;
        return SVH_OK;
    }
};

// ---------------------------------------------------------------------------
// per-device lane pool
// ---------------------------------------------------------------------------
struct Pool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Lane*> lanes;
    std::vector<Lane*> free_list;
    int max_lanes = 4;
};

static std::mutex g_mu;
static std::map<int, Pool*> g_pools;
static std::atomic<int> g_lanes{4};

static Pool* pool_for(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_pools.find(device);
    if (it != g_pools.end()) return it->second;
    Pool* p = new Pool();
    g_pools[device] = p;
    return p;
}

static Lane* acquire_lane(int device) {
    Pool* p = pool_for(device);
    std::unique_lock<std::mutex> lk(p->mu);
    p->max_lanes = g_lanes.load();
    for (;;) {
        if (!p->free_list.empty()) {
            Lane* l = p->free_list.back();
            p->free_list.pop_back();
            return l;
        }
        if ((int)p->lanes.size() < p->max_lanes) {
            Lane* l = new Lane();
            l->device = device;
            p->lanes.push_back(l);
            return l;
        }
        p->cv.wait(lk);
    }
}

static void release_lane(Lane* l) {
    Pool* p = pool_for(l->device);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->free_list.push_back(l);
    }
    p->cv.notify_one();
}

// ---------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------
struct Taps {
    bool enabled = false;
    std::vector<uint8_t> data[SVH_ELAS_STAGE_COUNT];
};

}  // namespace svh

struct svh_elas {
    svh_elas_params p;
    int device;
    svh::Taps taps;
    std::vector<std::string> tnames;
    std::vector<float> tms;
};

namespace svh {

struct PairIO {
    const uint8_t* I[2];
    int32_t pitch;
    bool in_device;
    float* D[2];
    bool out_device;
};

static int check_params(const svh_elas_params& p, int32_t W, int32_t H) {
    if (W < 16 || H < 16) return fail(SVH_ERR_BAD_ARG, "image smaller than 16x16");
    if (p.disp_max < 1 || p.disp_max > 4095) return fail(SVH_ERR_BAD_ARG, "disp_max out of range");
    if (p.grid_size < 1 || p.candidate_stepsize < 1) return fail(SVH_ERR_BAD_ARG, "bad grid/step");
    return SVH_OK;
}

template <typename T>
static int tap_dev(Lane& L, Taps* taps, int stage, const T* dev, size_t count) {
    if (!taps || !taps->enabled) return SVH_OK;
    taps->data[stage].resize(count * sizeof(T));
    HIP_TRY(hipMemcpyAsync(taps->data[stage].data(), dev, count * sizeof(T), hipMemcpyDeviceToHost,
                           L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    return SVH_OK;
}

template <typename T>
static void tap_host(Taps* taps, int stage, const T* src, size_t count) {
    if (!taps || !taps->enabled) return;
    taps->data[stage].resize(count * sizeof(T));
    if (count) memcpy(taps->data[stage].data(), src, count * sizeof(T));
}

// one stereo pair through one lane
static int run_pair(Lane& L, const svh_elas_params& p, const int32_t* dims, const PairIO& io,
                    Taps* taps, svh_elas* timing) {
    const int32_t W = dims[0], H = dims[1];
    int rc = check_params(p, W, H);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(L.device));
    rc = L.ensure(p, W, H);
    if (rc) return rc;
    const Dims& d = L.d;
    hipStream_t s = L.stream;
    L.prof.stream = s;
    const LaunchCtx cx = {s, g_prof_on.load() ? &L.prof : nullptr};
    const size_t N = (size_t)W * H, DN = (size_t)d.DW * d.DH;
    double t0 = now_ms();

    // ---- phase A ---------------------------------------------------------
    DevImages img;
    if (io.in_device) {
        img.I[0] = io.I[0]; img.I[1] = io.I[1];
        img.pitch[0] = img.pitch[1] = io.pitch;
    } else {
        // the caller's rows (any stride, pageable) are packed into pinned staging
        // on the host, then each image goes up in one linear DMA
        for (int k = 0; k < 2; k++) {
            uint8_t* dst = L.h_img + (size_t)k * N;
            if (io.pitch == W) memcpy(dst, io.I[k], N);
            else
                for (int32_t v = 0; v < H; v++) memcpy(dst + (size_t)v * W, io.I[k] + (size_t)v * io.pitch, W);
            HIP_TRY(hipMemcpyAsync(L.img[k], dst, N, hipMemcpyHostToDevice, s));
        }
        img.I[0] = L.img[0]; img.I[1] = L.img[1];
        img.pitch[0] = img.pitch[1] = W;
    }
    launch_descriptor(cx, img, W, H, p.subsampling, L.desc[0], L.desc[1]);
    launch_support(cx, p, d, L.desc[0], L.desc[1], L.dcan);
    const size_t nc = (size_t)d.Wc * d.Hc;
    HIP_TRY(hipMemcpyAsync(L.h_dcan, L.dcan, nc * sizeof(int16_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    L.prof.collect();
    double t1 = now_ms();
    if (taps && taps->enabled) {
        rc = tap_dev(L, taps, SVH_ELAS_DESC1, L.desc[0], N * 16); if (rc) return rc;
        rc = tap_dev(L, taps, SVH_ELAS_DESC2, L.desc[1], N * 16); if (rc) return rc;
        tap_host(taps, SVH_ELAS_DCAN_RAW, L.h_dcan, nc);
    }

    // ---- host ------------------------------------------------------------
    HostPrior& hp = L.hp;
    L.dcan_work.assign(L.h_dcan, L.h_dcan + nc);
    support_from_candidates(p, d, L.dcan_work.data(), hp.support);
    tap_host(taps, SVH_ELAS_SUPPORT, hp.support.data(), hp.support.size());
    if (hp.support.size() / 3 < 3) {
        // elas.cpp:69-75: message on stdout, outputs untouched
        printf("ERROR: Need at least 3 support points!\n");
        fflush(stdout);
        return SVH_ERR_FEW_SUPPORT;
    }
    if (!build_prior(p, d, hp)) return fail(SVH_ERR_UNSUPPORTED, "triangulation failed");
    if (taps && taps->enabled) {
        for (int k = 0; k < 2; k++) {
            tap_host(taps, SVH_ELAS_TRI1 + k, hp.tri[k].data(), hp.tri[k].size());
            tap_host(taps, SVH_ELAS_PLANES1 + k, hp.planes[k].data(), hp.planes[k].size());
            std::vector<int32_t> g;
            expand_grid(p, d, hp, k, g);
            tap_host(taps, SVH_ELAS_GRID1 + k, g.data(), g.size());
        }
    }
    // pack the prior into one pinned block -> one H2D copy
    size_t off = 0;
    auto put = [&](const void* src, size_t bytes) {
        size_t at = off;
        if (bytes) memcpy(L.h_prior + at, src, bytes);
        off = (off + bytes + 63) & ~(size_t)63;
        return at;
    };
    size_t o_r[2], o_off[2], o_d[2];
    for (int k = 0; k < 2; k++) {
        o_r[k] = put(hp.raster[k].data(), hp.raster[k].size() * sizeof(TriRaster));
        o_off[k] = put(hp.cell_off[k].data(), hp.cell_off[k].size() * sizeof(int32_t));
        o_d[k] = put(hp.cell_d[k].data(), hp.cell_d[k].size() * sizeof(uint16_t));
    }
    size_t o_P = put(hp.P.data(), hp.P.size() * sizeof(int32_t));
    if (off > L.prior_cap) return fail(SVH_ERR_BAD_ARG, "prior exceeds staging capacity");
    double t2 = now_ms();

    // ---- phase B ---------------------------------------------------------
    HIP_TRY(hipMemcpyAsync(L.prior_dev, L.h_prior, off, hipMemcpyHostToDevice, s));
    const TriRaster* r_dev[2] = {(const TriRaster*)(L.prior_dev + o_r[0]),
                                 (const TriRaster*)(L.prior_dev + o_r[1])};
    launch_owner(cx, d, r_dev[0], (int32_t)hp.raster[0].size(), r_dev[1],
                 (int32_t)hp.raster[1].size(), p.subsampling, L.owner[0], L.owner[1]);
    MatchArgs ma;
    for (int k = 0; k < 2; k++) {
        ma.desc[k] = L.desc[k];
        ma.owner[k] = L.owner[k];
        ma.raster[k] = r_dev[k];
        ma.cell_off[k] = (const int32_t*)(L.prior_dev + o_off[k]);
        ma.cell_d[k] = (const uint16_t*)(L.prior_dev + o_d[k]);
        ma.D[k] = L.Draw[k];
    }
    ma.P = (const int32_t*)(L.prior_dev + o_P);
    ma.plane_radius = hp.plane_radius;
    launch_match(cx, p, d, ma);
    if (taps && taps->enabled) {
        rc = tap_dev(L, taps, SVH_ELAS_D1_RAW, L.Draw[0], DN); if (rc) return rc;
        rc = tap_dev(L, taps, SVH_ELAS_D2_RAW, L.Draw[1], DN); if (rc) return rc;
    }
    // when the caller's maps live on the device the post-processing chain runs in
    // place on them: no final copy
    float* D1 = io.out_device ? io.D[0] : L.D[0];
    float* D2 = io.out_device ? io.D[1] : L.D[1];
    launch_lr(cx, p, d, L.Draw[0], L.Draw[1], D1, D2);
    if (taps && taps->enabled) {
        rc = tap_dev(L, taps, SVH_ELAS_D1_LR, D1, DN); if (rc) return rc;
        rc = tap_dev(L, taps, SVH_ELAS_D2_LR, D2, DN); if (rc) return rc;
    }
    const int nside = p.postprocess_only_left ? 1 : 2;
    for (int k = 0; k < nside; k++) launch_segments(cx, p, d, k ? D2 : D1, L.labels, L.runlen, L.counts);
    if (taps && taps->enabled) {
        rc = tap_dev(L, taps, SVH_ELAS_D1_SEG, D1, DN); if (rc) return rc;
        rc = tap_dev(L, taps, SVH_ELAS_D2_SEG, D2, DN); if (rc) return rc;
    }
    for (int k = 0; k < nside; k++) launch_gap(cx, p, d, k ? D2 : D1, L.tmp);
    if (taps && taps->enabled) {
        rc = tap_dev(L, taps, SVH_ELAS_D1_GAP, D1, DN); if (rc) return rc;
        rc = tap_dev(L, taps, SVH_ELAS_D2_GAP, D2, DN); if (rc) return rc;
    }
    if (p.filter_adaptive_mean)
        for (int k = 0; k < nside; k++) launch_adaptive_mean(cx, p, d, k ? D2 : D1, L.tmp);
    if (p.filter_median)
        for (int k = 0; k < nside; k++) launch_median(cx, d, k ? D2 : D1, L.tmp);

    if (!io.out_device) {
        HIP_TRY(hipMemcpyAsync(io.D[0], D1, DN * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(io.D[1], D2, DN * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    L.prof.collect();
    double t3 = now_ms();
    if (timing) {
        timing->tnames = {"Descriptor+Support Matches (device)", "Filters+Delaunay+Planes+Grid (host)",
                          "Matching+L/R+Segments+Gap+Mean (device)"};
        timing->tms = {(float)(t1 - t0), (float)(t2 - t1), (float)(t3 - t2)};
    }
    return SVH_OK;
}

}  // namespace svh

// ===========================================================================
// C-ABI
// ===========================================================================
using namespace svh;

extern "C" {

const char* svh_version(void) { return "svhip 0.1 (gfx950)"; }
const char* svh_last_error(void) { return t_error.c_str(); }

int32_t svh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t svh_set_device(int32_t device) {
    if (device < 0 || device >= svh_device_count())
        return fail(SVH_ERR_NO_DEVICE, "no such HIP device");
    t_device = device;
    return SVH_OK;
}

int32_t svh_profile_enable(int32_t on) {
    g_prof_on.store(on ? 1 : 0);
    return SVH_OK;
}

void svh_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.clear();
}

int32_t svh_profile_get(int32_t index, const char** name, double* total_ms, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (index < 0) return (int32_t)g_prof.size();
    int32_t i = 0;
    for (auto& kv : g_prof) {
        if (i++ == index) {
            if (name) *name = kv.first.c_str();
            if (total_ms) *total_ms = kv.second.total_ms;
            if (launches) *launches = kv.second.launches;
            return SVH_OK;
        }
    }
    return SVH_ERR_BAD_ARG;
}

int32_t svh_elas_set_lanes(int32_t lanes) {
    if (lanes < 1) lanes = 1;
    if (lanes > 64) lanes = 64;
    g_lanes.store(lanes);
    return lanes;
}

void svh_elas_params_default(svh_elas_params* p, int32_t setting) {
    // Elas::parameters::parameters(setting)  libelas/src/elas.h:86-147
    const bool rob = setting == SVH_ELAS_ROBOTICS;
    p->disp_min = 0;
    p->disp_max = 255;
    p->support_threshold = rob ? 0.85f : 0.95f;
    p->support_texture = 10;
    p->candidate_stepsize = 5;
    p->incon_window_size = 5;
    p->incon_threshold = 5;
    p->incon_min_support = 5;
    p->add_corners = rob ? 0 : 1;
    p->grid_size = 20;
    p->beta = 0.02f;
    p->gamma = rob ? 3.f : 5.f;
    p->sigma = 1.f;
    p->sradius = rob ? 2.f : 3.f;
    p->match_texture = rob ? 1 : 0;
    p->lr_threshold = 2;
    p->speckle_sim_threshold = 1.f;
    p->speckle_size = 200;
    p->ipol_gap_width = rob ? 3 : 5000;
    p->filter_median = rob ? 0 : 1;
    p->filter_adaptive_mean = rob ? 1 : 0;
    p->postprocess_only_left = rob ? 1 : 0;
    p->subsampling = 0;
}

svh_elas* svh_elas_create(const svh_elas_params* p) {
    if (!p) return nullptr;
    svh_elas* e = new svh_elas();
    e->p = *p;
    e->device = t_device;
    return e;
}

void svh_elas_destroy(svh_elas* e) { delete e; }

int32_t svh_elas_set_taps(svh_elas* e, int32_t enable) {
    if (!e) return SVH_ERR_BAD_ARG;
    e->taps.enabled = enable != 0;
    return SVH_OK;
}

int32_t svh_elas_get_stage(svh_elas* e, int32_t stage, void* buf, size_t cap, size_t* size) {
    if (!e || stage < 0 || stage >= SVH_ELAS_STAGE_COUNT) return SVH_ERR_BAD_ARG;
    const std::vector<uint8_t>& v = e->taps.data[stage];
    if (size) *size = v.size();
    if (!buf) return SVH_OK;
    if (cap < v.size()) return fail(SVH_ERR_BAD_ARG, "tap buffer too small");
    if (!v.empty()) memcpy(buf, v.data(), v.size());
    return SVH_OK;
}

int32_t svh_elas_last_timing(svh_elas* e, const char** names, float* ms, int32_t cap) {
    if (!e) return 0;
    int32_t n = (int32_t)e->tms.size();
    if (n > cap) n = cap;
    for (int32_t i = 0; i < n; i++) {
        names[i] = e->tnames[i].c_str();
        ms[i] = e->tms[i];
    }
    return n;
}

static int32_t require_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SVH_ERR_NO_DEVICE, "no HIP device visible: libsvhip has no CPU fallback");
    if (device >= n) return fail(SVH_ERR_NO_DEVICE, "HIP device index out of range");
    return SVH_OK;
}

int32_t svh_elas_process(svh_elas* e, const uint8_t* I1, const uint8_t* I2, float* D1, float* D2,
                         const int32_t* dims) {
    if (!e || !I1 || !I2 || !D1 || !D2 || !dims) return fail(SVH_ERR_BAD_ARG, "null argument");
    int32_t rc = require_device(e->device);
    if (rc) return rc;
    Lane* L = acquire_lane(e->device);
    PairIO io;
    io.I[0] = I1; io.I[1] = I2; io.pitch = dims[2]; io.in_device = false;
    io.D[0] = D1; io.D[1] = D2; io.out_device = false;
    rc = run_pair(*L, e->p, dims, io, &e->taps, e);
    release_lane(L);
    return rc;
}

static int32_t batch_impl(svh_elas* e, int32_t n, const int32_t* dims, int32_t* status,
                          const std::function<PairIO(int32_t)>& io_of) {
    int32_t rc = require_device(e->device);
    if (rc) return rc;
    const int lanes = std::min<int>(g_lanes.load(), n);
    std::atomic<int32_t> next{0};
    std::vector<int32_t> st(n, SVH_OK);
    std::vector<std::string> errs(lanes);
    auto worker = [&](int w) {
        Lane* L = acquire_lane(e->device);
        for (;;) {
            int32_t i = next.fetch_add(1);
            if (i >= n) break;
            PairIO io = io_of(i);
            st[i] = run_pair(*L, e->p, dims, io, nullptr, nullptr);
            if (st[i] < 0) errs[w] = t_error;
        }
        release_lane(L);
    };
    if (lanes <= 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (int w = 0; w < lanes; w++) th.emplace_back(worker, w);
        for (auto& t : th) t.join();
    }
    int32_t first = SVH_OK;
    for (int32_t i = 0; i < n; i++) {
        if (status) status[i] = st[i];
        if (first == SVH_OK && st[i] != SVH_OK) first = st[i];
    }
    for (auto& m : errs)
        if (!m.empty()) t_error = m;
    return first;
}

int32_t svh_elas_process_batch(svh_elas* e, int32_t n, const uint8_t* const* I1,
                               const uint8_t* const* I2, float* const* D1, float* const* D2,
                               const int32_t* dims, int32_t* status) {
    if (!e || n < 0 || !I1 || !I2 || !D1 || !D2 || !dims) return fail(SVH_ERR_BAD_ARG, "null argument");
    if (n == 0) return SVH_OK;
    return batch_impl(e, n, dims, status, [&](int32_t i) {
        PairIO io;
        io.I[0] = I1[i]; io.I[1] = I2[i]; io.pitch = dims[2]; io.in_device = false;
        io.D[0] = D1[i]; io.D[1] = D2[i]; io.out_device = false;
        return io;
    });
}

int32_t svh_elas_process_batch_device(svh_elas* e, int32_t n, const uint8_t* dI1, const uint8_t* dI2,
                                      size_t in_stride, float* dD1, float* dD2, size_t out_stride,
                                      const int32_t* dims, int32_t* status) {
    if (!e || n < 0 || !dI1 || !dI2 || !dD1 || !dD2 || !dims) return fail(SVH_ERR_BAD_ARG, "null argument");
    if (n == 0) return SVH_OK;
    return batch_impl(e, n, dims, status, [&](int32_t i) {
        PairIO io;
        io.I[0] = dI1 + (size_t)i * in_stride; io.I[1] = dI2 + (size_t)i * in_stride;
        io.pitch = dims[2]; io.in_device = true;
        io.D[0] = (float*)((uint8_t*)dD1 + (size_t)i * out_stride);
        io.D[1] = (float*)((uint8_t*)dD2 + (size_t)i * out_stride);
        io.out_device = true;
        return io;
    });
}

}  // extern "C"
