// Engine behind the svh_elas_* C-ABI (include/svh.h).
//
// Shape of the pipeline for one stereo pair (reference: Elas::process,
// libelas/src/elas.cpp:32-170):
//
//   device phase A   Sobel+descriptor (both images, one launch), support
//                    candidate matching  ->  D_can (37 KB) to pinned host memory
//   host             in-place lattice filters, Delaunay x2  (~0.4 ms; serial in the
//                    reference as well), one packed upload (support, triangles, prior table)
//   device phase B   planes + raster records, grid bit sets, triangle ownership, dense
//                    matching of both maps with the L/R check (one launch), speckle
//                    labelling, gap interpolation + adaptive mean tile kernels (or the
//                    unfused kernels / median)  ->  D1, D2
//
// A "lane" owns one HIP stream, its device buffers (sized once per image
// geometry and reused: callers construct an Elas per frame,
// stereomapper/stereothread.cpp:113) and its pinned staging.  A single
// svh_elas_process() borrows a lane on the calling thread.  A batch runs G pairs
// per launch ("group") on double-buffered workers: each worker owns two lanes and
// overlaps the host section of group i with phase A of group i+1 (and the tail of
// group i-1); workers sleep-poll instead of spinning.  No default-stream work, no
// process-global mutable state besides the lane pool (mutex protected), so
// objects may be used concurrently from different threads like the reference's
// (maindialog.cpp:456-465, 514-518).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "svh_internal.h"

namespace svh {

static thread_local std::string t_error;
static thread_local int t_device = 0;

// shared by the Elas and Matcher engines: text behind svh_last_error()
int fail(int code, const std::string& msg) {
    t_error = msg;
    return code;
}

// ---- fault injection (svh_internal.h) --------------------------------------------------------------------------
namespace {
enum { FI_MALLOC, FI_LAUNCH, FI_COPY, FI_WAIT, FI_KINDS };
std::atomic<int> g_fi_kind{-1};          // -1: nothing armed
std::atomic<int64_t> g_fi_first{0}, g_fi_count{1};
std::atomic<int64_t> g_fi_seen[FI_KINDS];
int fi_kind_of(const char* t) {
    if (strstr(t, "hipMalloc") || strstr(t, "hipHostMalloc")) return FI_MALLOC;
    if (strstr(t, "hipMemcpy") || strstr(t, "hipMemset")) return FI_COPY;
    if (strstr(t, "hipGetLastError")) return FI_LAUNCH;
    if (strstr(t, "_wait") || strstr(t, "Synchronize") || strstr(t, "hipEventQuery")) return FI_WAIT;
    return -1;
}
bool fi_parse(const char* spec) {
    static const char* names[FI_KINDS] = {"malloc", "launch", "copy", "wait"};
    g_fi_kind.store(-1);
    if (!spec || !*spec) return true;
    const char* colon = strchr(spec, ':');
    if (!colon) return false;
    int kind = -1;
    for (int k = 0; k < FI_KINDS; k++)
        if (strlen(names[k]) == (size_t)(colon - spec) && !strncmp(spec, names[k], colon - spec)) kind = k;
    if (kind < 0) return false;
    char* end = nullptr;
    const long long n = strtoll(colon + 1, &end, 10);
    long long cnt = 1;
    if (end && *end == ':') cnt = strtoll(end + 1, nullptr, 10);
    if (n < 1 || cnt < 0) return false;
    for (auto& c : g_fi_seen) c.store(0);
    g_fi_first.store(n);
    g_fi_count.store(cnt);
    g_fi_kind.store(kind);
    return true;
}
}   // namespace
bool fi_armed() { return g_fi_kind.load(std::memory_order_relaxed) >= 0; }
bool fi_hit(const char* expr_text) {
    const int kind = g_fi_kind.load(std::memory_order_relaxed);
    if (kind < 0 || fi_kind_of(expr_text) != kind) return false;
    const int64_t i = g_fi_seen[kind].fetch_add(1) + 1, first = g_fi_first.load(), cnt = g_fi_count.load();
    return i >= first && (cnt == 0 || i < first + cnt);
}
void report_hip_failure(const char* entry) { fprintf(stderr, "svhip: %s: %s\n", entry, t_error.c_str()); }

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        const bool inj_ = fi_armed() && fi_hit(#expr);                                      \
        hipError_t e_ = inj_ ? hipErrorUnknown : (expr);                                    \
        if (e_ != hipSuccess)                                                               \
            return fail(SVH_ERR_HIP, std::string(#expr) + ": " +                            \
                                     (inj_ ? "injected failure (SVH_TEST_FAIL_AT)" : hipGetErrorString(e_))); \
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
static char g_prof_only[64] = "";   // when non-empty only this kernel is timed

struct EventProfiler : Profiler {
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;          // 2 per launch
    std::vector<const char*> names;
    size_t used = 0;
    bool skipping = false;
    void begin(const char* kernel) override {
        skipping = g_prof_only[0] && strcmp(kernel, g_prof_only) != 0;
        if (skipping) return;
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
        if (skipping || ev.size() < 2 * (used + 1)) return;
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
// lane: stream + buffers for a group of up to `gcap` pairs
// ---------------------------------------------------------------------------
struct Lane {
    EventProfiler prof;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t wait_ev = nullptr;   // completion marker polled by lane_wait
    hipStream_t copy_stream = nullptr;   // early D2H of maps that are final before the stream is done
    hipEvent_t match_ev = nullptr, copy_ev = nullptr;
    bool poll_wait = false;         // batch workers sleep-poll; single calls spin (lowest latency)
    bool parallel_host = true;      // host stage may use helper threads (off when every core is a worker)
    bool lone_group = true;         // the lane's group is the only one of its call (single call, batch of one group): latency-bound
    // geometry the buffers were sized for
    int32_t W = 0, H = 0, disp_max = -1, step = 0, grid_size = 0, sub = -1, gcap = 0;
    Dims d{};
    // device
    uint8_t* img = nullptr;        // [gcap][2][N]       staged host images
    uint8_t* desc = nullptr;       // [gcap][2][N*16]
    bool desc_fly = false;         // the group in flight left only the Sobel planes there (descriptors_on_the_fly)
    int16_t* dcan = nullptr;       // [gcap][nc]
    int32_t* owner = nullptr;      // [gcap][2][N]
    int64_t owner_hi = 0;          // every value stored in owner[] so far is <= owner_hi
    uint8_t* prior_dev = nullptr;  // packed upload: header, P, support, triangles
    TriRaster* raster = nullptr;   // [gcap*2*ntri_max]
    float* planes = nullptr;       // 6 per triangle
    uint32_t* seed = nullptr;      // [gcap][2][cells][gwords]
    uint32_t* mask = nullptr;
    uint16_t* lists = nullptr;     // [gcap][2][cells][32] candidate records (k_grid_list)
    float* Draw = nullptr;         // [gcap][2][DN]
    float* D = nullptr;            // [gcap][2][DN]  (host-output mode)
    float* tmp = nullptr;          // [gcap][2][DN]
    int32_t* labels = nullptr;
    int32_t* counts = nullptr;
    int32_t* seg_nroots = nullptr;   // [2 * gcap][tiles] tile-local roots per 64 x 16 tile (k_seg_tile -> k_seg_sum)
    // pinned host
    int16_t* h_dcan = nullptr;
    uint8_t* h_img = nullptr;
    uint8_t* h_prior = nullptr;
    size_t prior_cap = 0;
    size_t ntri_max = 0;
    std::vector<HostPrior> hp;
    std::vector<int32_t> P;
    // device-side E5-E7 (elas_stage_kernels.hip): scratch, counts read back through pinned memory
    StageDev stg{};
    bool stage_ok = false;          // the geometry fits the device stage and its scratch exists
    void* stage_blob = nullptr;     // one allocation behind the StageDev arrays
    StageCounts* h_counts = nullptr;
    hipEvent_t stage_ev = nullptr;  // the counts have arrived
    bool resident = false;          // the group on this lane runs the device stage
    bool force_host = false;        // rerun of a group the device stage handed back
    float P_key[5] = {-1, -1, -1, -1, -1};   // parameters the prior table on the device was built for
    size_t o_P = 0, o_sup = 0, o_tri = 0;    // fixed layout of prior_dev in resident mode

    void release() {
        if (!stream) return;
        (void)hipSetDevice(device);
        (void)hipFree(img); (void)hipFree(desc); (void)hipFree(dcan); (void)hipFree(owner);
        (void)hipFree(prior_dev); (void)hipFree(raster); (void)hipFree(planes); (void)hipFree(seed);
        (void)hipFree(mask); (void)hipFree(lists); lists = nullptr; (void)hipFree(Draw); (void)hipFree(D); (void)hipFree(tmp);
        (void)hipFree(labels); (void)hipFree(counts); (void)hipFree(seg_nroots); seg_nroots = nullptr;
        img = desc = prior_dev = nullptr; dcan = nullptr; owner = nullptr; raster = nullptr;
        planes = Draw = D = tmp = nullptr; seed = mask = nullptr; labels = counts = nullptr;
        (void)hipHostFree(h_dcan); (void)hipHostFree(h_prior); (void)hipHostFree(h_img);
        h_dcan = nullptr; h_prior = nullptr; h_img = nullptr;
        (void)hipFree(stage_blob); (void)hipHostFree(h_counts);
        stage_blob = nullptr; h_counts = nullptr; stage_ok = false; stg = StageDev{};
        P_key[0] = -1;
        W = H = 0;
    }

    // everything the lane holds, streams and events included (svh_elas_trim)
    void destroy() {
        if (!stream) return;
        (void)hipSetDevice(device);
        release();
        for (hipEvent_t* e : {&wait_ev, &match_ev, &copy_ev, &stage_ev})
            if (*e) {
                (void)hipEventDestroy(*e);
                *e = nullptr;
            }
        for (hipEvent_t e : prof.ev) (void)hipEventDestroy(e);
        prof.ev.clear();
        prof.names.clear();
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        (void)hipStreamDestroy(stream);
        copy_stream = stream = nullptr;
    }

    int ensure(const svh_elas_params& p, int32_t w, int32_t h, int32_t g) {
        if (!stream) {
            HIP_TRY(hipSetDevice(device));
            HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        }
        const int32_t st = p.candidate_stepsize + (p.subsampling ? p.candidate_stepsize % 2 : 0);
        if (w == W && h == H && p.disp_max == disp_max && st == step && p.grid_size == grid_size &&
            p.subsampling == sub && g <= gcap)
            return SVH_OK;
        hipStream_t keep = stream;
        release();
        stream = keep;
        d = make_dims(p, w, h);
        const size_t N = (size_t)w * h, DN = (size_t)d.DW * d.DH, G2 = (size_t)2 * g;
        HIP_TRY(hipMalloc(&img, G2 * N));
        HIP_TRY(hipMalloc(&desc, G2 * N * 16));
        HIP_TRY(hipMalloc(&owner, G2 * N * sizeof(int32_t)));
        HIP_TRY(hipMemset(owner, 0, G2 * N * sizeof(int32_t)));
        // SVH_TEST_OWNER_HI: start the moving base just below the int32 limit so that a test
        // reaches the (otherwise once-per-40 000-groups) re-clear path with its first groups
        owner_hi = svh::env("SVH_TEST_OWNER_HI") ? atoll(svh::env("SVH_TEST_OWNER_HI")) : 0;
        HIP_TRY(hipMalloc(&Draw, G2 * DN * sizeof(float)));
        HIP_TRY(hipMalloc(&D, G2 * DN * sizeof(float)));
        HIP_TRY(hipMalloc(&tmp, G2 * DN * sizeof(float)));
        HIP_TRY(hipMalloc(&labels, G2 * DN * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&counts, G2 * DN * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&seg_nroots, G2 * (size_t)((d.DW + 63) / 64) * ((d.DH + 15) / 16) * sizeof(int32_t)));   // one count per 64 x 16 tile
        const size_t nc = (size_t)d.Wc * d.Hc;
        HIP_TRY(hipMalloc(&dcan, g * nc * sizeof(int16_t)));
        HIP_TRY(hipHostMalloc(&h_dcan, g * nc * sizeof(int16_t)));
        HIP_TRY(hipHostMalloc(&h_img, G2 * N));
        // worst case per pair: nc+6 support points, 2n+8 triangles per side
        const size_t nsup = nc + 6;
        ntri_max = 2 * nsup + 8;
        prior_cap = sizeof(GroupHdr) + (size_t)(p.disp_max + 1) * sizeof(int32_t) + 512 +
                    (size_t)g * (nsup * 3 + 2 * ntri_max * 3) * sizeof(int32_t);
        HIP_TRY(hipMalloc(&prior_dev, prior_cap));
        HIP_TRY(hipHostMalloc(&h_prior, prior_cap));
        HIP_TRY(hipMalloc(&raster, G2 * ntri_max * sizeof(TriRaster)));
        HIP_TRY(hipMalloc(&planes, G2 * ntri_max * 6 * sizeof(float)));
        const size_t gw_bytes = G2 * d.gw * d.gh * d.gwords * sizeof(uint32_t);
        HIP_TRY(hipMalloc(&seed, gw_bytes));
        HIP_TRY(hipMalloc(&mask, gw_bytes));
        if (d.gwords <= 8) HIP_TRY(hipMalloc(&lists, G2 * d.gw * d.gh * 32 * sizeof(uint16_t)));
        hp.resize(g);
        // fixed layout of the packed lists when the device builds them
        o_P = (sizeof(GroupHdr) + 63) & ~(size_t)63;
        o_sup = (o_P + (size_t)(p.disp_max + 1) * sizeof(int32_t) + 63) & ~(size_t)63;
        o_tri = (o_sup + (size_t)g * nsup * 3 * sizeof(int32_t) + 63) & ~(size_t)63;
        if (stage_device_ok(p, d) && o_tri + (size_t)g * 2 * ntri_max * 3 * sizeof(int32_t) <= prior_cap) {
            // scratch of the device stage, one allocation: see StageDev
            const size_t cap = std::min<size_t>(nsup, 65535), rec = 2 * cap + 2, S2 = (size_t)2 * g;
            size_t off = 0;
            auto take = [&](size_t bytes) { size_t at = off; off = (off + bytes + 255) & ~(size_t)255; return at; };
            const size_t a_sup = take((size_t)g * 3 * cap * 4), a_cnt = take(sizeof(StageCounts));
            const size_t a_ids = take(S2 * 4 * rec * 4), a_xys = take(S2 * 4 * rec * 4), a_nbr = take(S2 * 4 * rec * 4);
            size_t a_arr[11];
            for (int k = 0; k < 11; k++) a_arr[k] = take(S2 * cap * 4);
            const size_t a_fl = take(S2 * 2 * cap * 4), a_fr = take(S2 * 2 * cap * 4);
            const size_t a_wl = take((size_t)g * 3 * nc * 4), a_cw = take((size_t)g * (nc / 4 + 1) * 4);
            HIP_TRY(hipMalloc(&stage_blob, off));
            HIP_TRY(hipMemset(stage_blob, 0, off));
            HIP_TRY(hipHostMalloc(&h_counts, sizeof(StageCounts)));
            uint8_t* b = static_cast<uint8_t*>(stage_blob);
            stg.dcan = dcan;
            stg.sup_raw = reinterpret_cast<int32_t*>(b + a_sup);
            stg.counts = reinterpret_cast<StageCounts*>(b + a_cnt);
            stg.ids = reinterpret_cast<int32_t*>(b + a_ids);
            stg.xys = reinterpret_cast<int32_t*>(b + a_xys);
            stg.nbr = reinterpret_cast<uint32_t*>(b + a_nbr);
            int32_t** ia[6] = {&stg.pxy, &stg.buck, &stg.buck2, &stg.byx, &stg.order, &stg.oxy};
            uint32_t** ua[4] = {&stg.lx, &stg.ly, &stg.tmp, &stg.P};
            for (int k = 0; k < 6; k++) *ia[k] = reinterpret_cast<int32_t*>(b + a_arr[k]);
            for (int k = 0; k < 4; k++) *ua[k] = reinterpret_cast<uint32_t*>(b + a_arr[6 + k]);
            stg.dmap = reinterpret_cast<int32_t*>(b + a_arr[10]);
            stg.fl = reinterpret_cast<uint32_t*>(b + a_fl);
            stg.fr = reinterpret_cast<uint32_t*>(b + a_fr);
            stg.wl = reinterpret_cast<int32_t*>(b + a_wl);
            stg.cntw = reinterpret_cast<uint32_t*>(b + a_cw);
            stg.sup_cap = (int32_t)cap;
            stg.rec_cap = (int32_t)rec;
            stage_ok = true;
        }
        // the memsets above run on the null stream, which the lanes' non-blocking streams do not
        // wait for: they must have landed before the first kernel of this lane
        HIP_TRY(hipDeviceSynchronize());
        W = w; H = h; disp_max = p.disp_max; step = st; grid_size = p.grid_size; sub = p.subsampling;
        gcap = g;
        return SVH_OK;
    }
};

// Wait for everything enqueued on the lane's stream.  A single svh_elas_process call spins in
// hipStreamSynchronize (lowest latency).  Batch workers must NOT spin: a batch keeps as many
// workers as cores busy, containers often cap the CPU time of the whole process, and the
// Delaunay / lattice-filter work of the other workers needs those cycles (measured on the
// 16-core-quota MI355X box: same throughput at 10 instead of 16 cores).  They poll a completion
// event and sleep in between (SVH_WAIT_US, default 40; 0 = spin).
static std::atomic<int> g_wait_us{40};   // (SVH_WAIT_US / svh_config::wait_us: apply_config below)
static std::atomic<bool> g_hostprof{false};   // SVH_HOST_PROF (apply_config)
static hipError_t lane_wait(Lane& L) {
    if (!L.poll_wait || g_wait_us <= 0) return hipStreamSynchronize(L.stream);
    if (!L.wait_ev) {
        hipError_t e = hipEventCreateWithFlags(&L.wait_ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipEventRecord(L.wait_ev, L.stream);
    if (e != hipSuccess) return e;
    for (;;) {
        e = hipEventQuery(L.wait_ev);
        if (e != hipErrorNotReady) return e;
        std::this_thread::sleep_for(std::chrono::microseconds(g_wait_us.load(std::memory_order_relaxed)));
    }
}

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
// (the pools live as long as the process -- batch crew and stream workers may still be parked on them at exit -- and
// stay reachable through this never-destroyed map, so that leak checkers do not report them)
// (hardware queues: the ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues, default 4, and the
// kernels of one queue run one after the other -- a worker's stream then waits behind another worker's k_delaunay.  The
// count the engine wants, 20, is asked for by svh_init / the implicit initialisation, see svh_init.cpp; round 5 set it
// from a load-time constructor here.)

static std::map<int, Pool*>& g_pools = *new std::map<int, Pool*>();
// Defaults = the settings the headline number is measured with (bench.py used to set them itself; a C++ caller
// of the batch entries got 27-31 k pairs/s instead of 33.6-33.9 k with round 4's 8 workers x 16 pairs and 4 queues):
// six workers (each double-buffered; more only stretches every kernel's in-run duration), 32 pairs per launch
// for KITTI-size images, 16 hardware queues (below).
static std::atomic<int> g_lanes{6};
static std::atomic<int> g_group{32};
static std::atomic<bool> g_group_set{false};   // svh_elas_set_group was called: take the value as is
static std::atomic<int> g_stage_mode{-1};   // where E5-E7 run, see stage_mode_from_env() below
static int stage_mode_from_env();
// svh_init / the implicit initialisation (svh_init.cpp): an explicit call applies its fields, then -- in either case, while
// the environment is honoured -- the switches that used to be read when the library was loaded
static void apply_config(const svh_config& c, int explicit_call) {
    if (explicit_call) {
        if (c.elas_workers > 0) svh_elas_set_lanes(c.elas_workers);
        if (c.elas_pairs_per_launch > 0) svh_elas_set_group(c.elas_pairs_per_launch);
        svh_elas_set_stage(c.elas_stage);
        if (c.wait_us >= 0) g_wait_us = c.wait_us;
    }
    if (const char* e = svh::env("SVH_WAIT_US")) g_wait_us = atoi(e);
    if (svh::env("SVH_STAGE")) g_stage_mode.store(stage_mode_from_env());
    g_hostprof = svh::env("SVH_HOST_PROF") != nullptr;
}
namespace { struct ConfigHook { ConfigHook() { svh::on_config(apply_config); } } g_config_hook; }
// pairs per launch for an image of N pixels: the default (32) is meant for KITTI-size pairs, whose
// group holds ~1.5 GB of lane buffers; much larger images get proportionally smaller groups (1920x1080:
// 16 -- round 5 measured 6.5 k pairs/s at 8 per launch, 7.1-7.3 k at 16, a collapse to 4.5 k at 32)
// The automatic value is also bounded by the device's FREE memory (round 6; the defaults were tuned on one 288 GB
// part): every worker holds two buffer sets of ~120 bytes per pixel and pair (maps, descriptors / planes, ownership,
// labels, stage scratch), and all the workers' sets together take at most half of what hipMemGetInfo reports free on
// the calling thread's device -- a smaller device, or many ranks on one, get smaller groups instead of a failing
// hipMalloc.  An explicit svh_elas_set_group / svh_config::elas_pairs_per_launch is taken as is.
static int group_for(size_t N) {
    const int g = std::max(1, std::min(g_group.load(), kMaxGroup));
    if (g_group_set.load()) return g;
    size_t a = std::max<size_t>(1, std::min<size_t>(g, (size_t)32 * 1024 * 1024 / std::max<size_t>(N, 1)));
    size_t free_b = 0, total_b = 0;
    if (a > 1 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
        const size_t per_pair = 120 * std::max<size_t>(N, 1), sets = (size_t)2 * std::max(1, g_lanes.load());
        a = std::max<size_t>(1, std::min<size_t>(a, free_b / 2 / (per_pair * sets)));
    } else {
        (void)hipGetLastError();
    }
    return (int)a;
}

static Pool* pool_for(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_pools.find(device);
    if (it != g_pools.end()) return it->second;
    Pool* p = new Pool();
    g_pools[device] = p;
    return p;
}

// takes `count` (1 or 2) lanes at once -- all or nothing, so two callers can never hold one
// lane each while waiting for a second.  A batch worker is double-buffered (2 lanes), hence
// the pool may grow to twice the configured number of workers.
static void acquire_lanes(int device, int count, Lane** out) {
    Pool* p = pool_for(device);
    std::unique_lock<std::mutex> lk(p->mu);
    p->max_lanes = 2 * g_lanes.load() + 2;   // (+2: single calls still find a lane while a stream holds its workers' lanes)
    for (;;) {
        const int can_make = p->max_lanes - (int)p->lanes.size();
        if ((int)p->free_list.size() + std::max(can_make, 0) >= count) {
            for (int i = 0; i < count; i++) {
                if (!p->free_list.empty()) {
                    out[i] = p->free_list.back();
                    p->free_list.pop_back();
                } else {
                    Lane* l = new Lane();
                    l->device = device;
                    p->lanes.push_back(l);
                    out[i] = l;
                }
            }
            return;
        }
        p->cv.wait(lk);
    }
}

static Lane* acquire_lane(int device) {
    Lane* l = nullptr;
    acquire_lanes(device, 1, &l);
    return l;
}

static void release_lane(Lane* l) {
    Pool* p = pool_for(l->device);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->free_list.push_back(l);
    }
    p->cv.notify_all();
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

// inputs / outputs of one group of g consecutive pairs
struct GroupIO {
    int32_t g;
    bool in_device;
    const uint8_t* dI[2];            // device: image k of pair j at dI[k] + j*in_stride
    size_t in_stride;
    const uint8_t* const* hI[2];     // host: per-pair pointers
    int32_t pitch;
    bool out_device;
    float* dD[2];                    // device: map k of pair j at dD[k] + j*out_stride (floats)
    size_t out_stride;
    float* const* hD[2];             // host: per-pair pointers
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
    if (!count) return SVH_OK;
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

// one group of pairs through one lane; status[j] per pair
// A group runs in three steps that may be issued separately (the batch workers overlap the
// host step of one group with the device steps of the next, see batch_impl):
//   RG_A       enqueue descriptor + support matching + the candidate download
//   RG_HOST_B  wait for A, lattice filters + Delaunay on the host, enqueue everything else
//   RG_FINISH  wait for the stream, collect errors and kernel timings
enum { RG_A = 1, RG_HOST_B = 2, RG_FINISH = 4, RG_ALL = 7 };

// SVH_HOST_PROF=1: per-stage thread-CPU time of the batch workers, printed at exit
enum { HP_ENQ_A, HP_WAIT, HP_FILTER, HP_DELAUNAY, HP_PACK, HP_ENQ_B, HP_N };
static std::atomic<uint64_t> g_hp_ns[HP_N];
static std::atomic<uint64_t> g_hp_pairs{0};
static uint64_t cpu_ns() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + ts.tv_nsec;
}
struct HostProfDump {
    ~HostProfDump() {
        if (!g_hostprof || !g_hp_pairs) return;
        static const char* nm[HP_N] = {"enqueue A", "waits", "filters", "delaunay", "pack", "enqueue B"};
        double tot = 0;
        for (int i = 0; i < HP_N; i++) tot += g_hp_ns[i];
        fprintf(stderr, "[svh host prof] %llu pairs, %.1f us cpu/pair:", (unsigned long long)g_hp_pairs.load(),
                tot / 1e3 / g_hp_pairs);
        for (int i = 0; i < HP_N; i++) fprintf(stderr, "  %s %.1f", nm[i], g_hp_ns[i] / 1e3 / g_hp_pairs);
        fprintf(stderr, "\n");
    }
};
static HostProfDump g_hp_dump;
#define HP_MARK(slot) do { if (g_hostprof) { uint64_t n_ = cpu_ns(); g_hp_ns[slot] += n_ - hp_t; hp_t = n_; } } while (0)

// where E5-E7 run: -1 = auto (batches on the device, single calls on the host: two host threads
// are the lower latency for one pair), 0 = host, 1 = device.  SVH_STAGE=host|device, svh_elas_set_stage()
static int stage_mode_from_env() {   // (called by apply_config)
    const char* e = svh::env("SVH_STAGE");
    if (!e || !*e || !strcmp(e, "auto") || !strcmp(e, "-1")) return -1;
    if (!strcmp(e, "host") || !strcmp(e, "0")) return 0;
    if (!strcmp(e, "device") || !strcmp(e, "1")) return 1;
    fprintf(stderr, "svhip: SVH_STAGE=%s is not one of auto|-1, host|0, device|1: using auto\n", e);
    return -1;
}

static std::atomic<int64_t> g_stage_dev_groups{0}, g_stage_redo_groups{0};

// sleep-poll (batch workers) or block on an event
static hipError_t event_wait(Lane& L, hipEvent_t ev) {
    if (!L.poll_wait || g_wait_us <= 0) return hipEventSynchronize(ev);
    for (;;) {
        hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        std::this_thread::sleep_for(std::chrono::microseconds(g_wait_us.load(std::memory_order_relaxed)));
    }
}

static int run_group_body(Lane& L, const svh_elas_params& p, const int32_t* dims, const GroupIO& io,
                          int32_t* status, Taps* taps, svh_elas* timing, int mode, int prefer_device);

// A HIP failure anywhere in a group: nothing of that group may still be running when the caller hears of it (its
// output buffers may go away next), and the lane must be fit for the next group.  The lane's streams are drained
// (best effort: a lost device fails these calls too), the sticky error is cleared, events recorded for the timer are
// dropped and the flags a half-issued group leaves behind are reset.  The maps of the group are untouched when the
// failure came before its first output was written (allocation, upload, any launch check or wait of phase A and of
// the host stage); a failure of the final wait is reported with the maps already written.
static int run_group(Lane& L, const svh_elas_params& p, const int32_t* dims, const GroupIO& io,
                     int32_t* status, Taps* taps, svh_elas* timing, int mode = RG_ALL, int prefer_device = 0) {
    const int rc = run_group_body(L, p, dims, io, status, taps, timing, mode, prefer_device);
    if (rc == SVH_ERR_HIP) {
        const std::string keep = t_error;
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        if (L.copy_stream) (void)hipStreamSynchronize(L.copy_stream);
        (void)hipGetLastError();
        L.prof.used = 0;
        L.resident = false;
        L.force_host = false;
        L.P_key[0] = -1;      // the prior table may not have reached the device
        t_error = keep;
    }
    return rc;
}

static int run_group_body(Lane& L, const svh_elas_params& p, const int32_t* dims, const GroupIO& io,
                          int32_t* status, Taps* taps, svh_elas* timing, int mode, int prefer_device) {
    const int32_t W = dims[0], H = dims[1], g = io.g;
    int rc = check_params(p, W, H);
    if (rc) return rc;
    if (g < 1 || g > kMaxGroup) return fail(SVH_ERR_BAD_ARG, "bad group size");
    HIP_TRY(hipSetDevice(L.device));
    // lanes are sized for the group at hand: a single Elas::process call holds buffers for one pair
    // (41 MB at KITTI size), not for the batch default; batch workers size theirs up front
    rc = L.ensure(p, W, H, g);
    if (rc) return rc;
    if (g > 1) taps = nullptr;
    const Dims& d = L.d;
    hipStream_t s = L.stream;
    L.prof.stream = s;
    const LaunchCtx cx = {s, g_prof_on.load() ? &L.prof : nullptr, L.lone_group};
    const size_t N = (size_t)W * H, DN = (size_t)d.DW * d.DH;
    const size_t nc = (size_t)d.Wc * d.Hc;
    const bool tapping = taps && taps->enabled;
    double t0 = now_ms();
    uint64_t hp_t = g_hostprof ? cpu_ns() : 0;

    GroupHdr* hdr = reinterpret_cast<GroupHdr*>(L.h_prior);   // host view of the header
    DevMaps out;
    if (io.out_device) {
        // the callers' maps live on the device: the post-processing chain runs in place on them
        out.D[0] = io.dD[0]; out.D[1] = io.dD[1]; out.stride[0] = out.stride[1] = io.out_stride;
    } else {
        out.D[0] = L.D; out.D[1] = L.D + DN; out.stride[0] = out.stride[1] = 2 * DN;
    }
    // With postprocess_only_left the right map is final once the L/R check is done: its copy to the
    // caller's host buffer runs on a second stream while the left map is still being post-processed.
    bool early_d2 = false;
    int32_t plane_radius = 2;

    // ---- everything after the triangulations (E8 .. E16); totals < 0: counts are on the device
    auto enqueue_phase_b = [&](size_t o_P, size_t o_sup, size_t o_tri, int32_t total_sup, int32_t total_tri,
                               int32_t tri_bound) -> int {
        GroupDev G;
        G.hdr = reinterpret_cast<const GroupHdr*>(L.prior_dev);
        G.P = reinterpret_cast<const int32_t*>(L.prior_dev + o_P);
        G.support = reinterpret_cast<const int32_t*>(L.prior_dev + o_sup);
        G.tri = reinterpret_cast<const int32_t*>(L.prior_dev + o_tri);
        G.raster = L.raster;
        G.planes = L.planes;
        G.seed = L.seed;
        G.mask = L.mask;
        G.lists = L.lists;
        G.desc = L.desc;
        G.desc_fly = L.desc_fly ? 1 : 0;
        G.owner = L.owner;
        // triangle ownership is stored as owner_base + 1 + index; the base moves above everything
        // written so far, so the 2 x N x g map is never cleared (one memset when int32 would overflow)
        if (L.owner_hi + 1 + tri_bound >= INT32_MAX) {
            HIP_TRY(hipMemsetAsync(L.owner, 0, (size_t)2 * L.gcap * N * sizeof(int32_t), s));
            L.owner_hi = 0;
        }
        G.owner_base = (int32_t)L.owner_hi;
        L.owner_hi += 1 + tri_bound;
        G.Draw = L.Draw;
        G.plane_radius = plane_radius;
        G.prior_absmax = 0;
        for (int32_t dd = 0; dd <= plane_radius && dd < (int32_t)L.P.size(); dd++)
            G.prior_absmax = std::max(G.prior_absmax, (int32_t)std::min<int64_t>(std::llabs((long long)L.P[dd]), INT32_MAX));
        launch_prior(cx, p, d, g, total_sup, total_tri, G);
        if (tapping) {
            const int32_t n1 = hdr->tri_end[0], n2 = hdr->tri_end[1] - hdr->tri_end[0];
            rc = tap_dev(L, taps, SVH_ELAS_PLANES1, L.planes, (size_t)6 * n1); if (rc) return rc;
            rc = tap_dev(L, taps, SVH_ELAS_PLANES2, L.planes + (size_t)6 * n1, (size_t)6 * n2); if (rc) return rc;
            const size_t words = (size_t)d.gw * d.gh * d.gwords;
            std::vector<uint32_t> m(2 * words);
            HIP_TRY(hipMemcpyAsync(m.data(), L.mask, m.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(lane_wait(L));
            for (int k = 0; k < 2; k++) {
                std::vector<int32_t> gr;
                expand_grid(p, d, m.data() + k * words, gr);
                tap_host(taps, SVH_ELAS_GRID1 + k, gr.data(), gr.size());
            }
        }
        launch_owner(cx, p, d, g, total_tri, G);
        const bool tiles = !tapping && post_tiles_ok(p);   // gap + mean (+ speckle mask) tile kernels
        // the row kernel also applies the L/R check (its inputs are the row it just matched)
        const char* match_err = nullptr;
        const bool lr_done = launch_match(cx, p, d, g, G, &out, tapping, &match_err);
        if (match_err) return fail(SVH_ERR_HIP, std::string("launch_match: ") + match_err);
        early_d2 = !io.out_device && lr_done && !tapping && p.postprocess_only_left;
        if (early_d2) {
            if (!L.copy_stream) {
                HIP_TRY(hipStreamCreateWithFlags(&L.copy_stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&L.match_ev, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&L.copy_ev, hipEventDisableTiming));
            }
            HIP_TRY(hipEventRecord(L.match_ev, s));
        }
        if (tapping) {
            rc = tap_dev(L, taps, SVH_ELAS_D1_RAW, L.Draw, DN); if (rc) return rc;
            rc = tap_dev(L, taps, SVH_ELAS_D2_RAW, L.Draw + DN, DN); if (rc) return rc;
        }
        const PostScratch ps = {L.tmp, L.labels, L.counts, L.seg_nroots};
        const int nside = p.postprocess_only_left ? 1 : 2;
        if (!lr_done) launch_lr(cx, p, d, g, G, out);
        if (tapping) {
            rc = tap_dev(L, taps, SVH_ELAS_D1_LR, out.D[0], DN); if (rc) return rc;
            rc = tap_dev(L, taps, SVH_ELAS_D2_LR, out.D[1], DN); if (rc) return rc;
        }
        launch_segments(cx, p, d, g, nside, G, out, ps, /*mask=*/!tiles);
        if (tapping) {
            rc = tap_dev(L, taps, SVH_ELAS_D1_SEG, out.D[0], DN); if (rc) return rc;
            rc = tap_dev(L, taps, SVH_ELAS_D2_SEG, out.D[1], DN); if (rc) return rc;
        }
        if (tiles) {
            launch_gap_mean_tiles(cx, p, d, g, nside, G, out, ps);   // gap + adaptive mean, two tile kernels
        } else {
            launch_gap(cx, p, d, g, nside, G, out, ps);
            if (tapping) {
                rc = tap_dev(L, taps, SVH_ELAS_D1_GAP, out.D[0], DN); if (rc) return rc;
                rc = tap_dev(L, taps, SVH_ELAS_D2_GAP, out.D[1], DN); if (rc) return rc;
            }
            if (p.filter_adaptive_mean) launch_adaptive_mean(cx, p, d, g, nside, G, out, ps);
            if (p.filter_median) launch_median(cx, d, g, nside, G, out, ps);
        }
        return SVH_OK;
    };

    // ---- finished maps of the active pairs to the callers' host buffers
    // map k of the active pairs [j0, j1) -> the callers' buffers.  Runs of pairs whose host buffers follow
    // one another (one big array, the usual case) go down as ONE strided copy (device maps of a pair are
    // interleaved D1, D2: source pitch 2 DN, destination pitch DN) instead of one copy per pair.
    static const bool strided = !(svh::env("SVH_D2H_STRIDED") && atoi(svh::env("SVH_D2H_STRIDED")) == 0);
    auto copy_map = [&](int k, hipStream_t cs, const int32_t* active) -> int {
        const size_t bytes = DN * sizeof(float);
        for (int32_t j = 0; j < g;) {
            if (!active[j]) { j++; continue; }
            int32_t e = j + 1;
            while (strided && e < g && active[e] &&
                   reinterpret_cast<const char*>(io.hD[k][e]) == reinterpret_cast<const char*>(io.hD[k][e - 1]) + bytes)
                e++;
            const float* src = L.D + ((size_t)2 * j + k) * DN;
            if (e - j > 1)
                HIP_TRY(hipMemcpy2DAsync(io.hD[k][j], bytes, src, 2 * bytes, bytes, (size_t)(e - j),
                                         hipMemcpyDeviceToHost, cs));
            else
                HIP_TRY(hipMemcpyAsync(io.hD[k][j], src, bytes, hipMemcpyDeviceToHost, cs));
            j = e;
        }
        return SVH_OK;
    };
    auto copy_out = [&](const int32_t* active) -> int {
        if (io.out_device) return SVH_OK;
        if (early_d2) {   // (issued after the post-processing launches: a pageable copy blocks this thread)
            HIP_TRY(hipStreamWaitEvent(L.copy_stream, L.match_ev, 0));
            rc = copy_map(1, L.copy_stream, active);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(L.copy_ev, L.copy_stream));
        }
        for (int k = 0; k < (early_d2 ? 1 : 2); k++) {
            rc = copy_map(k, s, active);
            if (rc) return rc;
        }
        if (early_d2) HIP_TRY(hipStreamWaitEvent(s, L.copy_ev, 0));   // the lane's stream ends after both
        return SVH_OK;
    };
    // per-pair result of the device stage, from the counts it sent back
    auto active_of = [&](int32_t j) {
        return L.h_counts->nsup[j] >= 3 && !(L.h_counts->flags[j] & (STG_DUP | STG_OVERFLOW));
    };

    auto finish = [&]() -> int {
        HIP_TRY(lane_wait(L));
        HIP_TRY(hipGetLastError());
        L.prof.collect();
        HP_MARK(HP_WAIT);
        if (!L.resident) return SVH_OK;
        g_stage_dev_groups++;
        bool redo = false;
        for (int32_t j = 0; j < g; j++) redo = redo || (L.h_counts->flags[j] & (STG_DUP | STG_OVERFLOW)) != 0;
        if (redo) {
            // coincident support points (or more points than the scratch holds): which duplicate
            // survives is decided by Triangle's pivot stream -- the host path reproduces that.  The whole
            // group goes through it again (rare: not with the reference's presets); its statuses and its
            // "need at least 3 support points" messages -- one per pair, as in the reference -- are the
            // ones the caller gets, so nothing is printed for this group here.
            g_stage_redo_groups++;
            L.resident = false;
            L.force_host = true;
            rc = run_group(L, p, dims, io, status, taps, timing, RG_ALL, 0);
            L.force_host = false;
            return rc;
        }
        for (int32_t j = 0; j < g; j++) {
            if (L.h_counts->nsup[j] < 3) {
                // elas.cpp:69-75: message on stdout, outputs untouched
                printf("ERROR: Need at least 3 support points!\n");
                fflush(stdout);
                status[j] = SVH_ERR_FEW_SUPPORT;
            } else {
                status[j] = SVH_OK;
            }
        }
        return SVH_OK;
    };
    if (mode == RG_FINISH) return finish();

    // ---- phase A ---------------------------------------------------------
    if (mode & RG_A) {
        DevImages img;
        if (io.in_device) {
            img.I[0] = io.dI[0]; img.I[1] = io.dI[1];
            img.stride = io.in_stride; img.pitch = io.pitch;
        } else {
            // Images in pinned (page-locked) host memory go up straight from the caller's buffers,
            // one DMA per image.  Pageable rows (any stride) are packed into the lane's pinned
            // staging on the host first, then the whole group goes up in one linear DMA.
            bool pinned = true;
            for (int32_t j = 0; j < g && pinned; j++)
                for (int k = 0; k < 2 && pinned; k++) {
                    hipPointerAttribute_t at;
                    pinned = hipPointerGetAttributes(&at, io.hI[k][j]) == hipSuccess && at.type == hipMemoryTypeHost;
                }
            (void)hipGetLastError();   // a pageable pointer makes the query fail: not an error here
            if (pinned) {
                // packed images whose buffers follow one another go up as one strided copy per camera
                // (device layout: I1, I2 of a pair interleaved)
                static const bool strided_up = !(svh::env("SVH_H2D_STRIDED") && atoi(svh::env("SVH_H2D_STRIDED")) == 0);
                for (int k = 0; k < 2; k++)
                    for (int32_t j = 0; j < g;) {
                        uint8_t* dst = L.img + ((size_t)2 * j + k) * N;
                        if (io.pitch != W) {
                            HIP_TRY(hipMemcpy2DAsync(dst, W, io.hI[k][j], io.pitch, W, H, hipMemcpyHostToDevice, s));
                            j++;
                            continue;
                        }
                        int32_t e = j + 1;
                        while (strided_up && e < g && io.hI[k][e] == io.hI[k][e - 1] + N) e++;
                        if (e - j > 1)
                            HIP_TRY(hipMemcpy2DAsync(dst, 2 * N, io.hI[k][j], N, N, (size_t)(e - j), hipMemcpyHostToDevice, s));
                        else
                            HIP_TRY(hipMemcpyAsync(dst, io.hI[k][j], N, hipMemcpyHostToDevice, s));
                        j = e;
                    }
            } else {
                for (int32_t j = 0; j < g; j++)
                    for (int k = 0; k < 2; k++) {
                        uint8_t* dst = L.h_img + ((size_t)2 * j + k) * N;
                        const uint8_t* src = io.hI[k][j];
                        if (io.pitch == W) memcpy(dst, src, N);
                        else
                            for (int32_t v = 0; v < H; v++) memcpy(dst + (size_t)v * W, src + (size_t)v * io.pitch, W);
                    }
                HIP_TRY(hipMemcpyAsync(L.img, L.h_img, (size_t)2 * g * N, hipMemcpyHostToDevice, s));
            }
            img.I[0] = L.img; img.I[1] = L.img + N;
            img.stride = 2 * N; img.pitch = W;
        }
        {
            // E1 writes the Sobel planes only when both matchers will assemble their descriptor rows themselves
            // (the descriptor taps of the parity tests need the full maps)
            std::vector<int32_t> Pt;
            int32_t pr = 2;
            prior_table(p, Pt, &pr);
            int32_t absmax = 0;
            for (int32_t dd = 0; dd <= pr && dd < (int32_t)Pt.size(); dd++)
                absmax = std::max(absmax, (int32_t)std::min<int64_t>(std::llabs((long long)Pt[dd]), INT32_MAX));
            static const int fly_env = svh::env("SVH_DESC_FLY") ? atoi(svh::env("SVH_DESC_FLY")) : 1;
            L.desc_fly = (!tapping || fly_env == 2) && descriptors_on_the_fly(p, d, absmax, pr, L.lists != nullptr);
        }
        if (tapping && L.desc_fly) {
            // SVH_DESC_FLY=2 (tests): the descriptor taps come from the full kernel, then the planes replace them
            launch_descriptor(cx, img, g, W, H, p.subsampling, L.desc, false);
            int rc2 = tap_dev(L, taps, SVH_ELAS_DESC1, L.desc, N * 16); if (rc2) return rc2;
            rc2 = tap_dev(L, taps, SVH_ELAS_DESC2, L.desc + N * 16, N * 16); if (rc2) return rc2;
            HIP_TRY(hipStreamSynchronize(s));
        }
        launch_descriptor(cx, img, g, W, H, p.subsampling, L.desc, L.desc_fly);
        launch_support(cx, p, d, g, L.desc, L.dcan, L.desc_fly);
        const int sm = g_stage_mode.load();
        L.resident = !L.force_host && L.stage_ok &&
                     (sm == 1 || (sm < 0 && prefer_device && stage_device_preferred(p, d, prefer_device > 1)));
        // a batch worker whose group runs the device stage has nothing to do on the host until the group
        // is done: it sleeps between polls also in a one-round (latency-bound) batch, instead of spinning
        // on a core that another rank may need (8 pairs of 1920x1080 per step: 4-7 cores -> < 1)
        if (L.resident && prefer_device) L.poll_wait = true;
        if (!L.resident || tapping)
            HIP_TRY(hipMemcpyAsync(L.h_dcan, L.dcan, g * nc * sizeof(int16_t), hipMemcpyDeviceToHost, s));
        if (L.resident) {
            // ---- E5-E7 on the device, then everything else right behind: no host round trip
            prior_table(p, L.P, &plane_radius);
            const float key[5] = {p.beta, p.gamma, p.sigma, p.sradius, (float)p.disp_max};
            if (memcmp(key, L.P_key, sizeof(key)) != 0) {
                // (the staging copy is reused right away: wait for it, once per parameter set)
                memcpy(L.h_prior + L.o_P, L.P.data(), L.P.size() * sizeof(int32_t));
                HIP_TRY(hipMemcpyAsync(L.prior_dev + L.o_P, L.h_prior + L.o_P, L.P.size() * sizeof(int32_t),
                                       hipMemcpyHostToDevice, s));
                HIP_TRY(hipStreamSynchronize(s));
                memcpy(L.P_key, key, sizeof(key));
            }
            if (tapping) {
                HIP_TRY(hipStreamSynchronize(s));
                if (!L.desc_fly) {
                    rc = tap_dev(L, taps, SVH_ELAS_DESC1, L.desc, N * 16); if (rc) return rc;
                    rc = tap_dev(L, taps, SVH_ELAS_DESC2, L.desc + N * 16, N * 16); if (rc) return rc;
                }
                tap_host(taps, SVH_ELAS_DCAN_RAW, L.h_dcan, nc);
            }
            launch_stage_device(cx, p, d, g, L.stg, reinterpret_cast<GroupHdr*>(L.prior_dev),
                                reinterpret_cast<int32_t*>(L.prior_dev + L.o_sup),
                                reinterpret_cast<int32_t*>(L.prior_dev + L.o_tri));
            if (!L.stage_ev) HIP_TRY(hipEventCreateWithFlags(&L.stage_ev, hipEventDisableTiming));
            HIP_TRY(hipMemcpyAsync(L.h_counts, L.stg.counts, sizeof(StageCounts), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(L.stage_ev, s));
            if (tapping) {
                // host copies of the header and the lists the device built (g == 1)
                HIP_TRY(hipMemcpyAsync(hdr, L.prior_dev, sizeof(GroupHdr), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                if (hdr->active[0]) {
                    rc = tap_dev(L, taps, SVH_ELAS_SUPPORT, reinterpret_cast<int32_t*>(L.prior_dev + L.o_sup),
                                 (size_t)3 * hdr->total_sup); if (rc) return rc;
                    rc = tap_dev(L, taps, SVH_ELAS_TRI1, reinterpret_cast<int32_t*>(L.prior_dev + L.o_tri),
                                 (size_t)3 * hdr->tri_end[0]); if (rc) return rc;
                    rc = tap_dev(L, taps, SVH_ELAS_TRI2,
                                 reinterpret_cast<int32_t*>(L.prior_dev + L.o_tri) + (size_t)3 * hdr->tri_end[0],
                                 (size_t)3 * (hdr->tri_end[1] - hdr->tri_end[0])); if (rc) return rc;
                }
            }
            if (!tapping || hdr->active[0]) {
                // bound of the triangle count for the ownership base: 2 n - 5 per side
                const int32_t tri_bound = 2 * g * 2 * L.stg.sup_cap;
                rc = enqueue_phase_b(L.o_P, L.o_sup, L.o_tri, -1, -1, tri_bound);
                if (rc) return rc;
            }
        }
        HP_MARK(HP_ENQ_A);
    }
    if (!(mode & RG_HOST_B)) return SVH_OK;
    double t1 = t0, t2 = t0;
    if (L.resident) {
        if (!io.out_device) {
            // host outputs: only maps of pairs that went through may be copied out
            HIP_TRY(event_wait(L, L.stage_ev));
            HP_MARK(HP_WAIT);
            int32_t active[kMaxGroup];
            for (int32_t j = 0; j < g; j++) active[j] = active_of(j);
            rc = copy_out(active);
            if (rc) return rc;
        }
        if (g_hostprof) g_hp_pairs += g;
        HP_MARK(HP_ENQ_B);
        t1 = t2 = now_ms();
    } else {
    // the host stage of a latency call triangulates on helper threads: they wake while the device still works
    const bool warm = L.parallel_host && g == 1;
    if (warm) helpers_warm(3, 1500);
    struct WarmOff {
        bool on;
        ~WarmOff() { if (on) helpers_warm(0, 0); }
    } warm_off{warm};
    HIP_TRY(lane_wait(L));
    HIP_TRY(hipGetLastError());
    L.prof.collect();
    HP_MARK(HP_WAIT);
    if (g_hostprof) g_hp_pairs += g;
    t1 = now_ms();
    if (tapping) {
        if (!L.desc_fly) {
            rc = tap_dev(L, taps, SVH_ELAS_DESC1, L.desc, N * 16); if (rc) return rc;
            rc = tap_dev(L, taps, SVH_ELAS_DESC2, L.desc + N * 16, N * 16); if (rc) return rc;
        }
        tap_host(taps, SVH_ELAS_DCAN_RAW, L.h_dcan, nc);
    }

    // ---- host: lattice filters + Delaunay, then one packed upload ------------
    memset(hdr, 0, sizeof(GroupHdr));
    hdr->npairs = g;
    int32_t total_sup = 0, total_tri = 0, nactive = 0;
    for (int32_t j = 0; j < g; j++) {
        HostPrior& hp = L.hp[j];
        support_from_candidates(p, d, L.h_dcan + j * nc, hp.support, /*write_back=*/false);
        HP_MARK(HP_FILTER);
        hdr->sup_off[j] = total_sup;
        status[j] = SVH_OK;
        if (hp.support.size() / 3 < 3) {
            // elas.cpp:69-75: message on stdout, outputs untouched
            printf("ERROR: Need at least 3 support points!\n");
            fflush(stdout);
            status[j] = SVH_ERR_FEW_SUPPORT;
            hp.support.clear();
            hp.tri[0].clear();
            hp.tri[1].clear();
        } else if (!triangulate_support(hp, /*parallel=*/L.parallel_host)) {
            return fail(SVH_ERR_UNSUPPORTED, "triangulation failed");
        } else {
            hdr->active[j] = 1;
            nactive++;
        }
        HP_MARK(HP_DELAUNAY);
        total_sup += (int32_t)(hp.support.size() / 3);
        for (int k = 0; k < 2; k++) {
            total_tri += (int32_t)(hp.tri[k].size() / 3);
            hdr->tri_end[2 * j + k] = total_tri;
        }
    }
    hdr->sup_off[g] = total_sup;
    hdr->total_sup = total_sup;
    hdr->total_tri = total_tri;
    if (tapping) {
        tap_host(taps, SVH_ELAS_SUPPORT, L.hp[0].support.data(), L.hp[0].support.size());
        for (int k = 0; k < 2; k++)
            tap_host(taps, SVH_ELAS_TRI1 + k, L.hp[0].tri[k].data(), L.hp[0].tri[k].size());
    }
    if (nactive == 0) return SVH_OK;   // nothing to match; every status is already set
    prior_table(p, L.P, &plane_radius);
    L.P_key[0] = -1;   // this upload overwrites the resident prior table
    size_t off = (sizeof(GroupHdr) + 63) & ~(size_t)63;
    auto put = [&](const void* src, size_t bytes) {
        size_t at = off;
        if (bytes) memcpy(L.h_prior + at, src, bytes);
        off = (off + bytes + 63) & ~(size_t)63;
        return at;
    };
    const size_t o_P = put(L.P.data(), L.P.size() * sizeof(int32_t));
    const size_t o_sup = off;
    for (int32_t j = 0; j < g; j++) {
        const std::vector<int32_t>& v = L.hp[j].support;
        if (!v.empty()) memcpy(L.h_prior + off, v.data(), v.size() * sizeof(int32_t));
        off += v.size() * sizeof(int32_t);
    }
    off = (off + 63) & ~(size_t)63;
    const size_t o_tri = off;
    for (int32_t j = 0; j < g; j++)
        for (int k = 0; k < 2; k++) {
            const std::vector<int32_t>& v = L.hp[j].tri[k];
            if (!v.empty()) memcpy(L.h_prior + off, v.data(), v.size() * sizeof(int32_t));
            off += v.size() * sizeof(int32_t);
        }
    if (off > L.prior_cap) return fail(SVH_ERR_BAD_ARG, "prior exceeds staging capacity");
    if (warm_off.on) {
        helpers_warm(0, 0);
        warm_off.on = false;
    }
    t2 = now_ms();
    HP_MARK(HP_PACK);

    // ---- phase B ---------------------------------------------------------
    HIP_TRY(hipMemcpyAsync(L.prior_dev, L.h_prior, off, hipMemcpyHostToDevice, s));
    rc = enqueue_phase_b(o_P, o_sup, o_tri, total_sup, total_tri, total_tri);
    if (rc) return rc;
    rc = copy_out(hdr->active);
    if (rc) return rc;
    HP_MARK(HP_ENQ_B);
    }
    if (!(mode & RG_FINISH)) return SVH_OK;
    rc = finish();
    if (rc) return rc;
    double t3 = now_ms();
    if (timing) {
        timing->tnames = {"Descriptor+Support Matches (device)", "Filters+Delaunay (host)",
                          "Planes+Grid+Matching+L/R+Segments+Gap+Mean (device)"};
        timing->tms = {(float)(t1 - t0), (float)(t2 - t1), (float)(t3 - t2)};
        if (L.resident) {
            timing->tnames = {"enqueue (host)", "whole pipeline incl. Filters+Delaunay (device)"};
            timing->tms = {(float)(t1 - t0), (float)(t3 - t1)};
        }
    }
    return SVH_OK;
}

}  // namespace svh

// ---------------------------------------------------------------------------
// streaming submission (svh_elas_stream_*): a bounded queue of groups in front of workers that
// pipeline them over two lanes each, exactly as a batch worker does -- but the workers and their
// lanes live as long as the stream, so nothing ramps up or drains between calls
// ---------------------------------------------------------------------------
struct StreamGroup {
    uint64_t first = 0;              // ticket of its first pair
    int32_t n = 0;
    bool device = false, closed = false, done = false;
    int rc = SVH_OK;                 // < 0: the whole group failed (err has the text)
    std::string err;
    int32_t status[svh::kMaxGroup];
    const uint8_t* hI[2][svh::kMaxGroup];
    float* hD[2][svh::kMaxGroup];
    const uint8_t* dI[2] = {nullptr, nullptr};
    float* dD[2] = {nullptr, nullptr};
    ptrdiff_t in_stride = 0, out_stride = 0;   // bytes / floats between consecutive pairs (device form)
};

struct svh_elas_stream {
    svh_elas_params p;
    int device = 0;
    int32_t dims[3] = {0, 0, 0};
    int32_t G = 1, depth = 0;
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_space, cv_push;
    std::deque<std::shared_ptr<StreamGroup>> order;   // every group with pairs that were not popped yet
    std::deque<std::shared_ptr<StreamGroup>> ready;   // closed, waiting for a worker
    std::shared_ptr<StreamGroup> open;                // the group being filled (also order.back())
    uint64_t next_ticket = 0;
    int32_t popped_in_head = 0, inflight = 0;
    bool stop = false;
    bool closing = false;        // svh_elas_stream_close has begun: pushes are refused, blocked ones return
    int32_t users = 0;           // producers inside stream_push (close waits for them before the stream goes away)
    std::vector<std::thread> workers;
};

// Parked worker threads for the batch entries: grown on demand, never torn down (they sleep on
// a condition variable between batches).
namespace {
class Crew {
public:
    void post(std::function<void()> job) {
        std::unique_lock<std::mutex> lk(mu_);
        jobs_.push_back(std::move(job));
        if (idle_ < (int)jobs_.size()) {
            std::thread(&Crew::run, this).detach();
        }
        lk.unlock();
        cv_.notify_one();
    }
    // the caller's jobs count `left` down; it sleeps in short naps (the jobs run for milliseconds)
    void wait(std::atomic<int>& left) {
        while (left.load(std::memory_order_acquire) > 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }

private:
    void run() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            idle_++;
            cv_.wait(lk, [&] { return !jobs_.empty(); });
            idle_--;
            std::function<void()> job = std::move(jobs_.front());
            jobs_.pop_front();
            lk.unlock();
            job();
            lk.lock();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> jobs_;
    int idle_ = 0;
};
Crew& crew() {
    static Crew* c = new Crew();   // leaked on purpose: its threads outlive static destruction
    return *c;
}
}  // namespace

// ===========================================================================
// C-ABI
// ===========================================================================
using namespace svh;

extern "C" {

#ifndef SVH_SRC_SHA
#define SVH_SRC_SHA "unstamped"
#endif
const char* svh_version(void) { return "svhip 0.2 (gfx950) src " SVH_SRC_SHA; }
const char* svh_last_error(void) { return t_error.c_str(); }

int32_t svh_device_count(void) {
    svh::ensure_init();
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

/* restrict the timer to one kernel name (NULL or "" = all kernels) */
void svh_profile_only(const char* kernel) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    strncpy(g_prof_only, kernel ? kernel : "", sizeof(g_prof_only) - 1);
    g_prof_only[sizeof(g_prof_only) - 1] = 0;
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

int64_t svh_elas_trim(void) {
    // lanes that are not borrowed right now give back their device and pinned memory, streams and
    // events; lanes in use are left alone and the pool regrows on demand
    int64_t freed = 0;
    std::vector<Pool*> pools;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& kv : g_pools) pools.push_back(kv.second);
    }
    for (Pool* p : pools) {
        std::vector<Lane*> idle;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            idle.swap(p->free_list);
            for (Lane* l : idle) p->lanes.erase(std::find(p->lanes.begin(), p->lanes.end(), l));
        }
        for (Lane* l : idle) {
            l->destroy();
            delete l;
            freed++;
        }
        p->cv.notify_all();
    }
    return freed;
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
    svh::ensure_init();
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
    GroupIO io{};
    io.g = 1;
    io.in_device = false;
    io.hI[0] = &I1; io.hI[1] = &I2; io.pitch = dims[2];
    io.out_device = false;
    io.hD[0] = &D1; io.hD[1] = &D2;
    int32_t st = SVH_OK;
    rc = run_group(*L, e->p, dims, io, &st, &e->taps, e);
    release_lane(L);
    if (rc == SVH_ERR_HIP) report_hip_failure("svh_elas_process");
    return rc ? rc : st;
}

// n pairs -> groups of up to g_group consecutive pairs, spread over the lanes
static int32_t batch_impl(svh_elas* e, int32_t n, const int32_t* dims, int32_t* status,
                          const std::function<GroupIO(int32_t, int32_t)>& io_of) {
    int32_t rc = require_device(e->device);
    if (rc) return rc;
    rc = check_params(e->p, dims[0], dims[1]);
    if (rc) return rc;
    const int32_t G = group_for((size_t)dims[0] * dims[1]);
    const int32_t ngroups = (n + G - 1) / G;
    const int lanes = std::min<int>(g_lanes.load(), ngroups);
    // automatic stage choice: 1 = a batch, 2 = a deep batch (32 pairs or more: what counts is throughput,
    // not the ~2 ms a group of large lattices spends in k_lattice + k_delaunay -- the device stage there too)
    const int deep = n >= 32 ? 2 : 1;
    std::vector<int32_t> st(n, SVH_OK);
    std::vector<int32_t> grc(ngroups, SVH_OK);
    std::vector<std::string> errs(std::max(lanes, 1));
    // Each worker owns TWO lanes (stream + buffers) and software-pipelines its groups: while it
    // filters and triangulates group i on the host, the device already runs descriptor + support
    // matching of group i+1 on the other lane, and the tail kernels of group i-1 behind that.
    auto worker = [&](int w) {
        Lane* slot[2] = {nullptr, nullptr};
        acquire_lanes(e->device, ngroups > 1 ? 2 : 1, slot);
        // size the lanes' buffers before taking work: allocation synchronises the
        // device and must not land in the middle of other lanes' launches later
        for (Lane* L : slot)
            if (L) {
                // a batch with no more groups than workers is one round deep: latency-bound, so
                // the two triangulations of a pair may run on two threads and the waits block in
                // the driver; otherwise every core already has a worker and the waits sleep-poll
                L->parallel_host = ngroups <= lanes;
                L->lone_group = ngroups == 1;
                L->poll_wait = !L->parallel_host;
                // (a failure here -- out of device memory -- is reported by the first run_group on this lane,
                // which sizes it again and returns the error for its group)
                if (hipSetDevice(L->device) == hipSuccess) (void)L->ensure(e->p, dims[0], dims[1], G);
            }
        struct Job { int32_t gi = -1; GroupIO io{}; int32_t first = 0; };
        // static round-robin shares (groups cost the same): a worker that prefetches its next
        // group never takes one away from a worker that has nothing to do yet
        int32_t my_next = w;
        auto take = [&]() {
            Job j;
            const int32_t gi = my_next;
            my_next += lanes;
            if (gi < ngroups) {
                j.gi = gi;
                j.first = gi * G;
                j.io = io_of(j.first, std::min(G, n - j.first));
            }
            return j;
        };
        auto note = [&](const Job& j, int rcj) {
            if (rcj != SVH_OK && grc[j.gi] == SVH_OK) {
                grc[j.gi] = rcj;
                if (rcj < 0) errs[w] = t_error;
            }
        };
        Job pending[2];   // group whose tail is still on slot k's stream
        Job cur = take();
        int k = 0;
        if (cur.gi >= 0) note(cur, run_group(*slot[0], e->p, dims, cur.io, &st[cur.first], nullptr, nullptr, RG_A, deep));
        while (cur.gi >= 0) {
            Job nxt = slot[1] ? take() : Job();
            if (nxt.gi >= 0) {
                const int o = 1 - k;
                if (pending[o].gi >= 0) {
                    note(pending[o], run_group(*slot[o], e->p, dims, pending[o].io, &st[pending[o].first], nullptr,
                                               nullptr, RG_FINISH));
                    pending[o].gi = -1;
                }
                note(nxt, run_group(*slot[o], e->p, dims, nxt.io, &st[nxt.first], nullptr, nullptr, RG_A, deep));
            }
            if (grc[cur.gi] == SVH_OK)
                note(cur, run_group(*slot[k], e->p, dims, cur.io, &st[cur.first], nullptr, nullptr, RG_HOST_B));
            pending[k] = cur;
            if (!slot[1]) {   // single lane: finish right away and take the next group on it
                note(cur, run_group(*slot[0], e->p, dims, cur.io, &st[cur.first], nullptr, nullptr, RG_FINISH));
                pending[0].gi = -1;
                nxt = take();
                if (nxt.gi >= 0)
                    note(nxt, run_group(*slot[0], e->p, dims, nxt.io, &st[nxt.first], nullptr, nullptr, RG_A, deep));
            } else {
                k = 1 - k;
            }
            cur = nxt;
        }
        for (int q = 0; q < 2; q++)
            if (slot[q]) {
                if (pending[q].gi >= 0)
                    note(pending[q], run_group(*slot[q], e->p, dims, pending[q].io, &st[pending[q].first], nullptr,
                                               nullptr, RG_FINISH));
                slot[q]->poll_wait = false;
                slot[q]->parallel_host = true;
                slot[q]->lone_group = true;
                release_lane(slot[q]);
            }
    };
    if (lanes <= 1) {
        worker(0);
    } else {
        // the workers are parked threads of a process-wide crew (a batch call used to create and
        // join `lanes` threads); worker 0 runs on the calling thread.  The thread-local error text
        // and device binding of a crew thread are set by the job itself.
        std::atomic<int> left{lanes - 1};
        const int dev = e->device;
        for (int w = 1; w < lanes; w++)
            crew().post([&, w, dev]() {
                t_device = dev;
                worker(w);
                left.fetch_sub(1, std::memory_order_release);
            });
        worker(0);
        crew().wait(left);
    }
    int32_t first_bad = SVH_OK;
    for (int32_t gi = 0; gi < ngroups; gi++)
        if (grc[gi] != SVH_OK) {
            for (int32_t i = gi * G; i < std::min(n, (gi + 1) * G); i++) st[i] = grc[gi];
        }
    for (int32_t i = 0; i < n; i++) {
        if (status) status[i] = st[i];
        if (first_bad == SVH_OK && st[i] != SVH_OK) first_bad = st[i];
    }
    for (auto& m : errs)
        if (!m.empty()) t_error = m;
    if (first_bad == SVH_ERR_HIP) report_hip_failure("svh_elas_process_batch");
    return first_bad;
}

int32_t svh_elas_process_batch(svh_elas* e, int32_t n, const uint8_t* const* I1,
                               const uint8_t* const* I2, float* const* D1, float* const* D2,
                               const int32_t* dims, int32_t* status) {
    if (!e || n < 0 || !I1 || !I2 || !D1 || !D2 || !dims) return fail(SVH_ERR_BAD_ARG, "null argument");
    if (n == 0) return SVH_OK;
    return batch_impl(e, n, dims, status, [&](int32_t first, int32_t cnt) {
        GroupIO io{};
        io.g = cnt;
        io.in_device = false;
        io.hI[0] = I1 + first; io.hI[1] = I2 + first; io.pitch = dims[2];
        io.out_device = false;
        io.hD[0] = D1 + first; io.hD[1] = D2 + first;
        return io;
    });
}

int32_t svh_elas_process_batch_device(svh_elas* e, int32_t n, const uint8_t* dI1, const uint8_t* dI2,
                                      size_t in_stride, float* dD1, float* dD2, size_t out_stride,
                                      const int32_t* dims, int32_t* status) {
    if (!e || n < 0 || !dI1 || !dI2 || !dD1 || !dD2 || !dims) return fail(SVH_ERR_BAD_ARG, "null argument");
    if (out_stride % sizeof(float)) return fail(SVH_ERR_BAD_ARG, "out_stride must be a multiple of 4");
    if (n == 0) return SVH_OK;
    return batch_impl(e, n, dims, status, [&](int32_t first, int32_t cnt) {
        GroupIO io{};
        io.g = cnt;
        io.in_device = true;
        io.dI[0] = dI1 + (size_t)first * in_stride; io.dI[1] = dI2 + (size_t)first * in_stride;
        io.in_stride = in_stride; io.pitch = dims[2];
        io.out_device = true;
        io.dD[0] = dD1 + (size_t)first * (out_stride / sizeof(float));
        io.dD[1] = dD2 + (size_t)first * (out_stride / sizeof(float));
        io.out_stride = out_stride / sizeof(float);
        return io;
    });
}

// ---- streaming submission ----------------------------------------------------------------
static GroupIO stream_io(const StreamGroup& g, int32_t pitch) {
    GroupIO io{};
    io.g = g.n;
    io.pitch = pitch;
    io.in_device = io.out_device = g.device;
    if (g.device) {
        io.dI[0] = g.dI[0]; io.dI[1] = g.dI[1];
        io.in_stride = (size_t)g.in_stride;
        io.dD[0] = g.dD[0]; io.dD[1] = g.dD[1];
        io.out_stride = (size_t)g.out_stride;
    } else {
        io.hI[0] = g.hI[0]; io.hI[1] = g.hI[1];
        io.hD[0] = g.hD[0]; io.hD[1] = g.hD[1];
    }
    return io;
}

// (mu held) the open group goes to the workers
static void stream_close_open(svh_elas_stream* s) {
    if (!s->open) return;
    s->open->closed = true;
    s->ready.push_back(s->open);
    s->open.reset();
    s->cv_work.notify_one();
}

// A stream worker holds its two lanes only WHILE IT HAS GROUPS IN FLIGHT (round 5; until then for the stream's
// lifetime, so that a second stream on the device found the pool empty, its workers blocked in acquire_lanes()
// for ever and svh_elas_stream_close() of it hung in join()).  Taking and returning lanes is a mutex and a
// vector pop; the lanes keep their buffers, so a busy stream loses nothing, and any number of streams, batch
// calls and single calls share the pool: a worker asks for both lanes at once (no hold-and-wait) and gives them
// back as soon as its queue runs dry.
static void stream_worker(svh_elas_stream* s) {
    t_device = s->device;
    (void)hipSetDevice(s->device);
    Lane* slot[2] = {nullptr, nullptr};
    auto take = [&](bool block) -> std::shared_ptr<StreamGroup> {
        std::unique_lock<std::mutex> lk(s->mu);
        if (block) s->cv_work.wait(lk, [&] { return s->stop || !s->ready.empty(); });
        if (s->ready.empty()) return nullptr;
        std::shared_ptr<StreamGroup> g = s->ready.front();
        s->ready.pop_front();
        return g;
    };
    auto note = [&](StreamGroup& g, int rc) {
        if (rc != SVH_OK && g.rc == SVH_OK) {
            g.rc = rc;
            if (rc < 0) g.err = t_error;
        }
    };
    std::shared_ptr<StreamGroup> pend[2];
    auto finish = [&](int q) {
        if (!pend[q]) return;
        StreamGroup& g = *pend[q];
        note(g, run_group(*slot[q], s->p, s->dims, stream_io(g, s->dims[2]), g.status, nullptr, nullptr, RG_FINISH));
        {
            std::lock_guard<std::mutex> lk(s->mu);
            g.done = true;
        }
        s->cv_done.notify_all();
        pend[q].reset();
    };
    for (;;) {
        std::shared_ptr<StreamGroup> cur = take(true);
        if (!cur) break;
        acquire_lanes(s->device, 2, slot);
        for (Lane* L : slot) {
            L->parallel_host = false;
            L->lone_group = false;
            L->poll_wait = true;      // the workers sleep between polls: a stream must not cost a core per lane
            (void)L->ensure(s->p, s->dims[0], s->dims[1], s->G);   // (a failure is reported by the group's run_group)
        }
        int k = 0;
        note(*cur, run_group(*slot[k], s->p, s->dims, stream_io(*cur, s->dims[2]), cur->status, nullptr, nullptr, RG_A, 2));
        while (cur) {
            std::shared_ptr<StreamGroup> nxt = take(false);
            if (nxt) {
                finish(1 - k);
                note(*nxt, run_group(*slot[1 - k], s->p, s->dims, stream_io(*nxt, s->dims[2]), nxt->status, nullptr,
                                     nullptr, RG_A, 2));
            }
            if (cur->rc == SVH_OK)
                note(*cur, run_group(*slot[k], s->p, s->dims, stream_io(*cur, s->dims[2]), cur->status, nullptr, nullptr,
                                     RG_HOST_B));
            pend[k] = cur;
            if (nxt) {
                k = 1 - k;
            } else {   // nothing queued right now: complete what is on the lanes, then wait for work
                finish(k);
                finish(1 - k);
            }
            cur = nxt;
        }
        for (Lane*& L : slot) {
            L->poll_wait = false;
            L->parallel_host = true;
            L->lone_group = true;
            release_lane(L);
            L = nullptr;
        }
    }
}

svh_elas_stream* svh_elas_stream_open(svh_elas* e, const int32_t* dims, int32_t depth) {
    if (!e || !dims) {
        fail(SVH_ERR_BAD_ARG, "null argument");
        return nullptr;
    }
    if (require_device(e->device) || check_params(e->p, dims[0], dims[1])) return nullptr;
    svh_elas_stream* s = new svh_elas_stream();
    s->p = e->p;
    s->device = e->device;
    for (int k = 0; k < 3; k++) s->dims[k] = dims[k];
    s->G = group_for((size_t)dims[0] * dims[1]);
    const int lanes = std::max(1, g_lanes.load());
    s->depth = depth > 0 ? depth : lanes * 2 * s->G;
    // a worker keeps two groups in flight: no more workers than the depth can feed
    // (and no more than the pool has lane pairs for: 2 * lanes + 2 lanes per device)
    const int nw = std::max(1, std::min(lanes, (s->depth + 2 * s->G - 1) / (2 * s->G)));
    for (int w = 0; w < nw; w++) s->workers.emplace_back(stream_worker, s);
    return s;
}

static int32_t stream_push(svh_elas_stream* s, bool device, const uint8_t* I1, const uint8_t* I2, float* D1,
                           float* D2, uint64_t* ticket) {
    if (!s || !I1 || !I2 || !D1 || !D2) return fail(SVH_ERR_BAD_ARG, "null argument");
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->stop || s->closing) return fail(SVH_ERR_BAD_ARG, "stream is closing");
    // (a producer blocked here while the stream is closed: close() wakes it, it leaves with an error, and
    // close() waits for `users` to reach zero before the stream is deleted)
    s->users++;
    s->cv_space.wait(lk, [&] { return s->stop || s->closing || s->inflight < s->depth; });
    if (s->stop || s->closing) {
        s->users--;
        s->cv_done.notify_all();
        return fail(SVH_ERR_BAD_ARG, "stream is closing");
    }
    StreamGroup* g = s->open.get();
    if (g && g->device != device) {
        stream_close_open(s);
        g = nullptr;
    }
    if (g && device) {
        // pairs of one launch lie at base + j * stride: the new pair must continue the progression
        const ptrdiff_t di1 = I1 - (g->dI[0] + (g->n - 1) * g->in_stride), di2 = I2 - (g->dI[1] + (g->n - 1) * g->in_stride);
        const ptrdiff_t dd1 = D1 - (g->dD[0] + (g->n - 1) * g->out_stride), dd2 = D2 - (g->dD[1] + (g->n - 1) * g->out_stride);
        bool fits;
        if (g->n == 1) {
            fits = di1 > 0 && di1 == di2 && dd1 > 0 && dd1 == dd2;
            if (fits) {
                g->in_stride = di1;
                g->out_stride = dd1;
            }
        } else {
            fits = di1 == g->in_stride && di2 == g->in_stride && dd1 == g->out_stride && dd2 == g->out_stride;
        }
        if (!fits) {
            stream_close_open(s);
            g = nullptr;
        }
    }
    if (!g) {
        s->open = std::make_shared<StreamGroup>();
        g = s->open.get();
        g->first = s->next_ticket;
        g->device = device;
        s->order.push_back(s->open);
    }
    const int32_t j = g->n++;
    g->status[j] = SVH_OK;
    if (device) {
        if (j == 0) {
            g->dI[0] = I1; g->dI[1] = I2;
            g->dD[0] = D1; g->dD[1] = D2;
        }
    } else {
        g->hI[0][j] = I1; g->hI[1][j] = I2;
        g->hD[0][j] = D1; g->hD[1][j] = D2;
    }
    if (ticket) *ticket = s->next_ticket;
    s->next_ticket++;
    s->inflight++;
    if (g->n >= s->G) stream_close_open(s);
    s->users--;
    s->cv_push.notify_all();
    s->cv_done.notify_all();
    return SVH_OK;
}

int32_t svh_elas_stream_push(svh_elas_stream* s, const uint8_t* I1, const uint8_t* I2, float* D1, float* D2,
                             uint64_t* ticket) {
    return stream_push(s, false, I1, I2, D1, D2, ticket);
}

int32_t svh_elas_stream_push_device(svh_elas_stream* s, const uint8_t* dI1, const uint8_t* dI2, float* dD1,
                                    float* dD2, uint64_t* ticket) {
    return stream_push(s, true, dI1, dI2, dD1, dD2, ticket);
}

int32_t svh_elas_stream_flush(svh_elas_stream* s) {
    if (!s) return fail(SVH_ERR_BAD_ARG, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    stream_close_open(s);
    return SVH_OK;
}

int32_t svh_elas_stream_pop(svh_elas_stream* s, uint64_t* ticket, int32_t* status, int32_t timeout_ms) {
    if (!s) return fail(SVH_ERR_BAD_ARG, "null argument");
    std::unique_lock<std::mutex> lk(s->mu);
    if (s->order.empty()) return SVH_ERR_EMPTY;
    std::shared_ptr<StreamGroup> g = s->order.front();
    if (!g->closed) stream_close_open(s);   // nothing older to wait for: the partial group starts now
    auto ready = [&] { return g->done; };
    if (timeout_ms < 0) {
        s->cv_done.wait(lk, ready);
    } else if (!s->cv_done.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready)) {
        return SVH_ERR_TIMEOUT;
    }
    const int32_t j = s->popped_in_head++;
    if (ticket) *ticket = g->first + (uint64_t)j;
    if (status) *status = g->rc != SVH_OK ? g->rc : g->status[j];
    if (g->rc < 0) {
        t_error = g->err;
        if (g->rc == SVH_ERR_HIP && j == 0) report_hip_failure("svh_elas_stream (group)");
    }
    if (s->popped_in_head >= g->n) {
        s->order.pop_front();
        s->popped_in_head = 0;
    }
    s->inflight--;
    lk.unlock();
    s->cv_space.notify_all();
    return SVH_OK;
}

int32_t svh_elas_stream_push_device_n(svh_elas_stream* s, int32_t n, const uint8_t* dI1, const uint8_t* dI2,
                                      size_t in_stride, float* dD1, float* dD2, size_t out_stride,
                                      uint64_t* first_ticket) {
    if (n < 0 || out_stride % sizeof(float)) return fail(SVH_ERR_BAD_ARG, "bad count / out_stride must be a multiple of 4");
    for (int32_t i = 0; i < n; i++) {
        uint64_t t = 0;
        const int32_t rc = stream_push(s, true, dI1 + (size_t)i * in_stride, dI2 + (size_t)i * in_stride,
                                       dD1 + (size_t)i * (out_stride / sizeof(float)),
                                       dD2 + (size_t)i * (out_stride / sizeof(float)), &t);
        if (rc) return rc;
        if (i == 0 && first_ticket) *first_ticket = t;
    }
    return SVH_OK;
}

int32_t svh_elas_stream_push_n(svh_elas_stream* s, int32_t n, const uint8_t* const* I1, const uint8_t* const* I2,
                               float* const* D1, float* const* D2, uint64_t* first_ticket) {
    if (n < 0 || (n > 0 && (!I1 || !I2 || !D1 || !D2))) return fail(SVH_ERR_BAD_ARG, "bad count / null pointer array");
    for (int32_t i = 0; i < n; i++) {
        uint64_t t = 0;
        const int32_t rc = stream_push(s, false, I1[i], I2[i], D1[i], D2[i], &t);
        if (rc) return rc;
        if (i == 0 && first_ticket) *first_ticket = t;
    }
    return SVH_OK;
}

int32_t svh_elas_stream_pop_n(svh_elas_stream* s, int32_t n, int32_t* status, int32_t* popped) {
    int32_t first_bad = SVH_OK, k = 0;
    for (; k < n; k++) {
        int32_t st = SVH_OK;
        int32_t rc = svh_elas_stream_pop(s, nullptr, &st, -1);
        while (rc == SVH_ERR_EMPTY) {
            // the consumer got ahead of the producer: wait for the next push (a closing stream ends the wait)
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv_push.wait(lk, [&] { return s->stop || !s->order.empty(); });
            if (s->stop && s->order.empty()) break;
            lk.unlock();
            rc = svh_elas_stream_pop(s, nullptr, &st, -1);
        }
        if (rc) {
            if (popped) *popped = k;
            return rc;
        }
        if (status) status[k] = st;
        if (first_bad == SVH_OK && st != SVH_OK) first_bad = st;
    }
    if (popped) *popped = k;
    return first_bad;
}

int32_t svh_elas_stream_close(svh_elas_stream* s) {
    if (!s) return fail(SVH_ERR_BAD_ARG, "null argument");
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->closing = true;              // no new pairs; producers blocked on a full stream return an error
        s->cv_space.notify_all();
        stream_close_open(s);
        // every pushed pair completes (the callers' buffers are written or left untouched as promised) and
        // every producer has left stream_push
        s->cv_done.wait(lk, [&] {
            if (s->users > 0) return false;
            for (auto& g : s->order)
                if (!g->done) return false;
            return true;
        });
        s->stop = true;
    }
    s->cv_work.notify_all();
    s->cv_space.notify_all();
    s->cv_push.notify_all();
    for (std::thread& t : s->workers) t.join();
    delete s;
    return SVH_OK;
}

int32_t svh_elas_support_from_candidates(const svh_elas_params* p, int32_t width, int32_t height,
                                         int16_t* dcan, int32_t* support, int32_t cap) {
    if (!p || !dcan || !support) return fail(SVH_ERR_BAD_ARG, "null argument");
    const Dims d = make_dims(*p, width, height);
    std::vector<int32_t> s;
    support_from_candidates(*p, d, dcan, s);
    const int32_t n = (int32_t)(s.size() / 3);
    for (int32_t i = 0; i < n && i < cap; i++)
        for (int k = 0; k < 3; k++) support[3 * i + k] = s[3 * i + k];
    return n;
}


/* tests: arm / disarm the fault injection ("" or NULL disarms); see svh_internal.h */
int32_t svh_test_fail_at(const char* spec) { return fi_parse(spec) ? SVH_OK : fail(SVH_ERR_BAD_ARG, "bad fault specification"); }

int32_t svh_elas_set_stage(int32_t where) {
    g_stage_mode.store(where < 0 ? -1 : (where ? 1 : 0));
    return g_stage_mode.load();
}

/* debug: phase time stamps of the last group's first slot (not in svh.h) */
void svh_debug_stage_stamps(int64_t* out32) {
    Pool* p = pool_for(t_device);
    std::lock_guard<std::mutex> lk(p->mu);
    for (Lane* l : p->lanes)
        if (l->stage_ok) {
            (void)hipMemcpy(out32, l->stg.counts->dbg, 32 * sizeof(int64_t), hipMemcpyDeviceToHost);
            return;
        }
}

void svh_elas_stage_stats(int64_t* device_groups, int64_t* handed_back) {
    if (device_groups) *device_groups = g_stage_dev_groups.load();
    if (handed_back) *handed_back = g_stage_redo_groups.load();
}

void svh_elas_get_settings(int32_t out[4]) {
    if (!out) return;
    out[0] = g_lanes.load();
    out[1] = g_group_set.load() ? g_group.load() : 0;
    out[2] = g_stage_mode.load();
    out[3] = g_wait_us;
}

int32_t svh_elas_set_group(int32_t pairs) {
    if (pairs < 1) pairs = 1;
    if (pairs > kMaxGroup) pairs = kMaxGroup;
    g_group.store(pairs);
    g_group_set.store(true);
    return pairs;
}

}  // extern "C"
