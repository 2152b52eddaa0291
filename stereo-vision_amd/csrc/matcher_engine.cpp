// Engine behind the svh_matcher_* C-ABI (include/svh.h): libviso2's Matcher.
//
//   pushBack       device: half-resolution image, 5x5 Sobel (matching and full
//                  resolution), blob + checkerboard responses, two non-maximum
//                  suppressions with ordered compaction, 32-byte descriptors.
//                  Feature tables stay on the device; only the four counts come
//                  back.  (reference: matcher.cpp:102-205, 780-878)
//   matchFeatures  device: bin indices, circular matching (one thread per query
//                  feature, four dependent hops), ordered compaction, 5x5
//                  relocation on the full-resolution Sobel images.
//                  host: Delaunay outlier vote and per-bin prior statistics on
//                  a few hundred / thousand matches (matcher.cpp:209-293).
//
// One object per visual-odometry instance, used by one thread at a time (like
// the reference's, viso.cpp:33); every object owns its stream and buffers, so
// objects on different threads do not interact.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "batch_rec.h"
#include "matcher_internal.h"
#include "vo_internal.h"

namespace svh {

int fail(int code, const std::string& msg);   // elas_engine.cpp: sets svh_last_error()
bool fi_armed();                                    // elas_engine.cpp: fault injection (svh_internal.h)
bool fi_hit(const char* expr_text);
void report_hip_failure(const char* entry);
static int mfail(int code, const std::string& msg) { return fail(code, msg); }

// Threads that are inside a compute entry of the Matcher / visual odometry right now (svh_matcher_push_back,
// svh_matcher_match_features, svh_vo_process, svh_vo_estimate_motion).  One or two = a single sequence (the
// latency path: spin on the stream, triangulate the outlier vote on the helper pool); more = several sequences
// share this GPU and the host cores: waits sleep between polls and the helper pool is left alone.  What counts
// is concurrent ACTIVITY, not how many objects exist (a process may hold many Matchers and drive one).
// SVH_MATCHER_WAIT=0 (spin) / 1 (sleep-poll) overrides the choice.
static std::atomic<int> g_active_callers{0};
static thread_local int t_entry_depth = 0;   // svh_vo_process calls the Matcher's entries: a thread counts once
ActiveCaller::ActiveCaller() {
    if (t_entry_depth++ == 0) g_active_callers.fetch_add(1, std::memory_order_relaxed);
}
ActiveCaller::~ActiveCaller() {
    if (--t_entry_depth == 0) g_active_callers.fetch_sub(1, std::memory_order_relaxed);
}
int wait_stream(void* stream) {
    hipStream_t s = (hipStream_t)stream;
    static const int forced = svh::env("SVH_MATCHER_WAIT") ? atoi(svh::env("SVH_MATCHER_WAIT")) : -1;   // 0 spin, 1 sleep-poll
    const bool poll = forced >= 0 ? forced == 1 : g_active_callers.load(std::memory_order_relaxed) > 2;
    if (!poll) return (int)hipStreamSynchronize(s);
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return (int)e;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// ---------------------------------------------------------------------------
// batch recorder (batch_rec.h) and the host-side helpers of the batched entries
// ---------------------------------------------------------------------------
thread_local BatchRec* t_rec = nullptr;

hipError_t BatchRec::flush(hipStream_t s) {
    cursor = 0;
    if (slots.empty()) return hipSuccess;
    auto al = [](size_t n) { return (n + 255) & ~(size_t)255; };
    size_t need = 0;
    for (const Slot& sl : slots) need += al(sl.jobs.size());
    if (used + need > cap) {
        // the arena may still be read by launches in flight: wait, then grow
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        if (h_arena) (void)hipHostFree(h_arena);
        if (d_arena) (void)hipFree(d_arena);
        h_arena = d_arena = nullptr;
        const size_t want = std::max<size_t>(2 * (used + need), 256 * 1024);
        cap = 0;      // (a failed allocation leaves an arena of size 0: the next flush allocates again)
        used = 0;
        e = hipHostMalloc((void**)&h_arena, want);
        if (e != hipSuccess) return e;
        e = hipMalloc((void**)&d_arena, want);
        if (e != hipSuccess) {
            (void)hipHostFree(h_arena);
            h_arena = nullptr;
            return e;
        }
        cap = want;
    }
    size_t off = used;
    std::vector<size_t> at;
    for (const Slot& sl : slots) {
        memcpy(h_arena + off, sl.jobs.data(), sl.jobs.size());
        at.push_back(off);
        off += al(sl.jobs.size());
    }
    hipError_t e = hipMemcpyAsync(d_arena + used, h_arena + used, need, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    for (size_t i = 0; i < slots.size(); i++) {
        const Slot& sl = slots[i];
        if (sl.njobs > 0) sl.fn(d_arena + at[i], sl.njobs, sl.gx, sl.gy, sl.lds, s);
    }
    used += need;
    slots.clear();
    if (track) {
        last_stream = s;
        flush_pending = true;
    }
    return hipGetLastError();
}

hipError_t BatchRec::ensure_side() {
    for (int i = 0; i < kSide; i++) {
        if (side[i]) continue;
        hipError_t e = hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        e = hipEventCreateWithFlags(&side_done[i], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t BatchRec::join_side(hipStream_t s) {
    for (int i = 0; i < kSide; i++) {
        if (!side[i]) continue;
        hipError_t e = hipEventRecord(side_done[i], side[i]);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, side_done[i], 0);
        if (e != hipSuccess) {
            // (seen once in a while with several threads in the runtime: "event last recorded in a capturing
            // stream"; the host waits for the side stream instead)
            (void)hipGetLastError();
            e = hipStreamSynchronize(side[i]);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

void BatchRec::release() {
    if (h_arena) (void)hipHostFree(h_arena);
    if (d_arena) (void)hipFree(d_arena);
    h_arena = d_arena = nullptr;
    cap = 0;
    for (int i = 0; i < kSide; i++) {
        if (side[i]) (void)hipStreamDestroy(side[i]);
        if (side_done[i]) (void)hipEventDestroy(side_done[i]);
        side[i] = nullptr;
        side_done[i] = nullptr;
    }
    flush_pending = false;
}

// The calling thread's recorder FOR A DEVICE (arena, side streams and events live on the device that was current
// when they were created; they are kept for the thread's lifetime).  Round 5: one recorder per (thread, device) --
// a thread that drove a lockstep batch on GPU 0 and then one on GPU 1 used to launch the second batch's kernels with
// a job table in GPU 0's memory.
namespace {
BatchRec& recorder_of(std::map<int, std::unique_ptr<BatchRec>>& recs, int device) {
    std::unique_ptr<BatchRec>& r = recs[device];
    if (!r) r.reset(new BatchRec());
    return *r;
}
}   // namespace
BatchRec& batch_recorder(int device) {
    static thread_local std::map<int, std::unique_ptr<BatchRec>> recs;
    return recorder_of(recs, device);
}
// a second one for svh_matcher_prefetch_batch: its launches are still in flight when the thread records the next
// phases of the frame before
BatchRec& prefetch_recorder(int device) {
    static thread_local std::map<int, std::unique_ptr<BatchRec>> recs;
    BatchRec& rec = recorder_of(recs, device);
    rec.track = true;
    return rec;
}

hipError_t BatchRec::reuse() {
    // (a stream wait, not an event: events recorded on this thread and waited for on streams that another thread
    // synchronises at the same moment came back as "event last recorded in a capturing stream" now and then)
    if (flush_pending) {
        const hipError_t e = hipStreamSynchronize(last_stream);
        if (e != hipSuccess) return e;
        flush_pending = false;
    }
    used = 0;
    return hipSuccess;
}

// Parked helper threads for the per-object HOST work of a batch call (outlier votes, prior statistics, row packing):
// parallel_for(n, fn) runs fn(0..n-1) on the helpers and the caller, returns when all are done.
// Several calls may be in flight at once (the prefetch thread packing frame t+1 while the caller votes on frame t,
// two calling threads with their own objects): every call is a job on the pool's list, the helpers take tasks from
// the jobs in turn, a caller works on its OWN job only (it returns as soon as that job is done).  Until round 5's
// last session the calls took turns on a mutex: the packing of the next frame and the votes of this one -- both on
// the critical path of a pipelined lockstep call -- waited for each other with helpers idle in the tail of either.
// SVH_POOL_SERIAL=1 keeps the take-turns form (A/B).
namespace {
class BatchPool {
    struct Job {
        const std::function<void(int)>* fn;
        int n;
        int next = 0, done = 0;
    };

public:
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 1) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        static const bool serial = svh::env("SVH_POOL_SERIAL") && atoi(svh::env("SVH_POOL_SERIAL")) != 0;
        std::unique_lock<std::mutex> one_call(call_mu_, std::defer_lock);
        if (serial) one_call.lock();
        Job job{&fn, n};
        {
            std::lock_guard<std::mutex> lk(mu_);
            const int want = std::min(n - 1, max_threads());
            while ((int)threads_.size() < want) threads_.emplace_back(&BatchPool::run, this);
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(mu_);
        while (job.next < job.n) {            // the caller's share of its own job
            const int i = take(&job);
            lk.unlock();
            fn(i);
            lk.lock();
            job.done++;
        }
        cv_done_.wait(lk, [&] { return job.done == job.n; });   // (the job left the list with its last task)
    }
    static BatchPool& get() {
        static BatchPool* p = new BatchPool();   // leaked on purpose: its threads outlive static destruction
        return *p;
    }

private:
    static int max_threads() {
        static const int n = std::max(1, std::min(15, (int)std::thread::hardware_concurrency() - 1));
        return n;
    }
    // mu_ held: next task of the job; a job whose tasks are all handed out leaves the list
    int take(Job* j) {
        const int i = j->next++;
        if (j->next == j->n) jobs_.erase(std::find(jobs_.begin(), jobs_.end(), j));
        return i;
    }
    void run() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return !jobs_.empty(); });
            Job* j = jobs_[turn_++ % jobs_.size()];   // the jobs in turn: neither call starves the other
            const int i = take(j);
            const std::function<void(int)>* fn = j->fn;
            lk.unlock();
            (*fn)(i);
            lk.lock();
            // (the job lives on its caller's stack until done == n, and this is the helper's last touch of it)
            if (++j->done == j->n) cv_done_.notify_all();
        }
    }
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, cv_done_;
    std::vector<std::thread> threads_;
    std::vector<Job*> jobs_;      // jobs with tasks left to hand out
    size_t turn_ = 0;
};
// One parked thread that runs the host side of svh_matcher_prefetch_batch (row packing on the helper threads,
// uploads, the recorded feature extraction) while the caller goes on with the frame before.
class PrefetchWorker {
public:
    std::shared_future<int32_t> post(std::function<int32_t()> fn) {
        std::packaged_task<int32_t()> task(std::move(fn));
        std::shared_future<int32_t> f = task.get_future().share();
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!started_) {
                std::thread(&PrefetchWorker::run, this).detach();
                started_ = true;
            }
            q_.push_back(std::move(task));
        }
        cv_.notify_one();
        return f;
    }
    static PrefetchWorker& get() {
        static PrefetchWorker* w = new PrefetchWorker();   // leaked on purpose, like the helper pool
        return *w;
    }

private:
    void run() {
        for (;;) {
            std::packaged_task<int32_t()> task;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !q_.empty(); });
                task = std::move(q_.front());
                q_.pop_front();
            }
            task();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::packaged_task<int32_t()>> q_;
    bool started_ = false;
};
thread_local bool t_in_batch = false;   // inside a batch call: the outlier vote does not fork (the pool is the parallelism)
}  // namespace

// a failed HIP call: svh_last_error() names it, one line on stderr at the point of failure (the entries above it
// only pass the code on), SVH_ERR_HIP
static int hip_failed(const char* expr, bool injected, hipError_t e) {
    const int rc = mfail(SVH_ERR_HIP, std::string(expr) + ": " + (injected ? "injected failure (SVH_TEST_FAIL_AT)" : hipGetErrorString(e)));
    report_hip_failure("Matcher");
    return rc;
}
#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        const bool inj_ = fi_armed() && fi_hit(#expr);   /* svh_internal.h: fault injection */ \
        hipError_t e_ = inj_ ? hipErrorUnknown : (expr);                                     \
        if (e_ != hipSuccess) return hip_failed(#expr, inj_, e_);                            \
    } while (0)

template <typename T>
static hipError_t dalloc(T** p, size_t count) {
    return hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
}
// Growing a buffer: the old one is freed and its pointer CLEARED before the new allocation is tried, and the callers
// set the capacity they track to 0 first and to the new size only after every allocation of the set succeeded -- a
// failure in the middle (found by the sanitizer run with injected failures, tools/sanitize_viso.cpp) then leaves
// "nothing allocated", which the next call repairs, instead of a stale pointer behind a capacity that says "fits".
template <typename T>
static hipError_t drealloc(T** p, size_t count) {
    (void)hipFree(*p);
    *p = nullptr;
    return dalloc(p, count);
}
template <typename T>
static hipError_t hrealloc(T** p, size_t bytes) {
    (void)hipHostFree(*p);
    *p = nullptr;
    return hipHostMalloc((void**)p, bytes);
}

// device-resident data of one camera image of one frame
struct DevView {
    bool valid = false;
    int32_t w = 0, h = 0, bpl = 0;       // full resolution
    int32_t mw = 0, mh = 0, mbpl = 0;    // matching resolution
    int32_t half = -1;
    uint8_t *I = nullptr, *Ih = nullptr, *du = nullptr, *dv = nullptr, *du_full = nullptr,
            *dv_full = nullptr;
    int16_t *f1 = nullptr, *f2 = nullptr;
    int32_t* tab[2] = {nullptr, nullptr};   // sparse, dense
    int32_t cap[2] = {0, 0};
    int32_t* cnt = nullptr;                 // device: n_sparse, n_dense
    int32_t n[2] = {0, 0};                  // host copy
    int32_t *off[2] = {nullptr, nullptr}, *ids[2] = {nullptr, nullptr};
    int32_t nbins = 0;
    int32_t off_cap = 0;      // bins the off[] arrays were allocated for
    uint8_t* stage = nullptr;               // pinned: the packed image (upload source, and getGain's host copy),
                                            // h rows of bpl bytes + one zero row

    void release() {
        (void)hipFree(I); (void)hipFree(Ih); (void)hipFree(du); (void)hipFree(dv); (void)hipFree(du_full);
        (void)hipFree(dv_full); (void)hipFree(f1); (void)hipFree(f2); (void)hipFree(cnt);
        (void)hipHostFree(stage);
        stage = nullptr;
        for (int k = 0; k < 2; k++) {
            (void)hipFree(tab[k]); (void)hipFree(off[k]); (void)hipFree(ids[k]);
            tab[k] = off[k] = ids[k] = nullptr;
            cap[k] = 0;
        }
        I = Ih = du = dv = du_full = dv_full = nullptr;
        f1 = f2 = nullptr;
        cnt = nullptr;
        w = h = 0;
        half = -1;
        nbins = 0;
        off_cap = 0;
        valid = false;
    }
};

}  // namespace svh

using namespace svh;

struct svh_matcher {
    svh_matcher_params p;
    int32_t margin;
    int device;
    hipStream_t stream;
    DevView prev[2], cur[2];
    // a frame handed over early (svh_matcher_prefetch_batch): packed, uploaded and its features computed into a
    // third view set while the frame before it is still being matched; the next pushBack without images takes it
    DevView next[2];
    int32_t dims_n[3] = {0, 0, 0};
    bool has_next = false;
    int next_cams = 0;
    hipStream_t next_stream = nullptr;        // where the prefetch was issued (waited for when the frame is taken)
    std::shared_future<int32_t> next_job;     // its host side, on the prefetch thread
    std::shared_ptr<std::string> next_why;    // ... and its error text, if it failed
    int32_t dims_p[3], dims_c[3];
    // scratch
    int4* slots[4] = {nullptr, nullptr, nullptr, nullptr};      // NMS scratch, one set per camera (the cameras' feature
    int32_t* flags[4] = {nullptr, nullptr, nullptr, nullptr};   // extraction runs on two streams) and, [2 + camera], a second
    int32_t* order[4] = {nullptr, nullptr, nullptr, nullptr};   // one for the sparse table (both tables by the same launches)
    int32_t slot_cap = 0;
    hipStream_t stream2 = nullptr;            // camera 1 during pushBack
    int32_t* cursor = nullptr;
    int32_t cursor_cap = 0;
    svh_p_match *pm_slots = nullptr, *pm_out = nullptr;
    int32_t *pm_flags = nullptr, *pm_count = nullptr;
    int32_t pm_cap = 0;
    int32_t* pixel_owner = nullptr;
    size_t owner_cap = 0;
    float* ranges_dev = nullptr;
    int32_t ranges_cap = 0;
    float* h_ranges = nullptr;                  // pinned copy of `ranges` (batched calls upload it from a kernel)
    size_t h_ranges_cap = 0;
    svh_p_match* h_pm = nullptr;                // pinned download staging for match lists
    int32_t* h_cnt = nullptr;                   // pinned: match count
    int32_t* h_n = nullptr;                     // pinned: feature counts [camera][sparse, dense]
    int32_t h_pm_cap = 0;
    bool taps = false;                          // keep every intermediate stage (parity tests)
    int32_t last_dense = 0;                     // matches of the last dense pass before the vote (predicts the next one)
    bool warm_on_wait = false;                  // wake the vote's helper threads once the dense pass is enqueued
    // results
    std::vector<svh_p_match> m1, m2;
    std::vector<svh_p_match> m2_kept;   // the match list of the last good matchFeatures while a new one is being built
    std::vector<float> ranges;    // [bins][16]
    std::vector<svh_p_match> stage[SVH_M_STAGE_COUNT];
    // SVH_MATCHER_TIMING=1: host wall-clock per step, printed by svh_matcher_destroy
    double tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double tfine[6] = {0, 0, 0, 0, 0, 0};
    int64_t tcalls[2] = {0, 0};
    // svh_matcher_set_timing (round 6): device time of the three device phases (pushBack's feature kernels, sparse
    // matching, dense matching + refinement) from HIP events on the object's streams: first launch -> last copy
    hipEvent_t tev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double tdev[3] = {0, 0, 0};
};

static double mnow_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// SVH_MATCHER_TIMING, evaluated at the first use (never while the library is loaded)
static std::atomic<int> g_mtiming_api{0};   // svh_matcher_set_timing
static bool mtiming_on() {
    static const bool on = svh::env("SVH_MATCHER_TIMING") != nullptr;
    return on || g_mtiming_api.load(std::memory_order_relaxed) != 0;
}
// event pair k of an object's device phase (0 pushBack stream 1, 1 pushBack stream 2, 2 sparse, 3 dense): created on
// first use while timing is on; failures only lose the measurement
static void tev_record(svh_matcher* m, int idx, hipStream_t s) {
    if (!m->tev[idx] && hipEventCreate(&m->tev[idx]) != hipSuccess) { m->tev[idx] = nullptr; (void)hipGetLastError(); return; }
    (void)hipEventRecord(m->tev[idx], s);
}
static double tev_ms(svh_matcher* m, int a, int b) {
    float ms = 0;
    if (!m->tev[a] || !m->tev[b] || hipEventElapsedTime(&ms, m->tev[a], m->tev[b]) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return ms;
}
#define g_mtiming mtiming_on()
enum { T_PACK = 0, T_PUSH_GPU, T_SPARSE, T_OUT1, T_PRIOR, T_DENSE, T_OUT2 };
// SVH_MATCHER_TIMING=1: wall-clock of the phases of the lockstep entries, printed at exit
struct BatchTiming {
    double t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int64_t calls[2] = {0, 0};
    ~BatchTiming() {
        if (!calls[0] || !calls[1]) return;
        const double a = 1.0 / (double)calls[0], b = 1.0 / (double)calls[1];
        fprintf(stderr, "[svh lockstep timing] pushBack: prepare %.3f, pack %.3f, record %.3f, flush %.3f, gpu wait %.3f ms | "
                        "matchFeatures: sparse record+flush %.3f, wait %.3f, votes+prior %.3f, dense record+flush %.3f, "
                        "wait %.3f, votes %.3f ms\n",
                t[0] * a, t[1] * a, t[2] * a, t[3] * a, t[4] * a, t[5] * b, t[6] * b, t[7] * b, t[8] * b, t[9] * b, t[10] * b);
    }
};
static BatchTiming g_btime;

namespace svh {

static int size_view(svh_matcher* m, DevView& V, int32_t w, int32_t h, int32_t bpl);
static int ensure_view(svh_matcher* m, DevView& V, int32_t w, int32_t h, int32_t bpl) {
    if (V.w == w && V.h == h && V.bpl == bpl && V.half == m->p.half_resolution) return SVH_OK;
    V.release();
    const int rc = size_view(m, V, w, h, bpl);
    if (rc) V.release();     // (a half-sized view must not look like one of the right geometry to the next call)
    return rc;
}
static int size_view(svh_matcher* m, DevView& V, int32_t w, int32_t h, int32_t bpl) {
    const svh_matcher_params& p = m->p;
    V.w = w; V.h = h; V.bpl = bpl;
    V.half = p.half_resolution;
    if (p.half_resolution) {
        V.mw = w / 2;
        V.mh = h / 2;
        V.mbpl = V.mw + 15 - (V.mw - 1) % 16;   // matcher.cpp:751-756
    } else {
        V.mw = w; V.mh = h; V.mbpl = bpl;
    }
    const size_t fn = (size_t)bpl * h, mn = (size_t)V.mbpl * V.mh;
    HIP_TRY(dalloc(&V.I, fn));
    HIP_TRY(dalloc(&V.du, mn));
    HIP_TRY(dalloc(&V.dv, mn));
    HIP_TRY(dalloc(&V.f1, mn));
    HIP_TRY(dalloc(&V.f2, mn));
    if (p.half_resolution) {
        HIP_TRY(dalloc(&V.Ih, mn));
        HIP_TRY(dalloc(&V.du_full, fn));
        HIP_TRY(dalloc(&V.dv_full, fn));
    }
    // one zero row past the image: getGain clamps its window to [0,H] INCLUSIVE like the reference
    // (matcher.cpp:362-371), whose read of row H is out of bounds; here it reads zeros
    HIP_TRY(hipHostMalloc((void**)&V.stage, fn + bpl));
    memset(V.stage + fn, 0, bpl);
    HIP_TRY(dalloc(&V.cnt, 2));
    HIP_TRY(hipMemsetAsync(V.cnt, 0, 2 * sizeof(int32_t), m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));   // (re)allocation path only; the cameras use two streams
    int32_t ns = p.nms_n * 3;
    if (ns > 10) ns = std::max(p.nms_n, 10);           // matcher.cpp:824-828
    const int32_t nn[2] = {ns, p.nms_n};
    for (int k = 0; k < 2; k++) {
        V.cap[k] = 4 * mnms_blocks(V.mw, nn[k], m->margin) * mnms_blocks(V.mh, nn[k], m->margin);
        HIP_TRY(dalloc(&V.tab[k], (size_t)12 * V.cap[k]));
        HIP_TRY(dalloc(&V.ids[k], (size_t)V.cap[k]));
    }
    return SVH_OK;
}

// scratch of the feature extraction (may run on the prefetch thread) and of the matching (the caller's thread):
// two functions, so that neither side reads the other's bookkeeping
static int ensure_feature_scratch(svh_matcher* m, int32_t slot_need) {
    if (slot_need > m->slot_cap) {
        m->slot_cap = 0;
        for (int c = 0; c < 4; c++) {
            HIP_TRY(drealloc(&m->slots[c], (size_t)slot_need));
            HIP_TRY(drealloc(&m->flags[c], (size_t)slot_need + 4));
            HIP_TRY(drealloc(&m->order[c], (size_t)slot_need));
        }
        m->slot_cap = slot_need;
    }
    return SVH_OK;
}

static int ensure_match_scratch(svh_matcher* m, int32_t pm_need, size_t owner_need) {
    if (pm_need > m->pm_cap) {
        m->pm_cap = 0;
        HIP_TRY(drealloc(&m->pm_slots, (size_t)pm_need));
        HIP_TRY(drealloc(&m->pm_out, (size_t)pm_need));
        HIP_TRY(drealloc(&m->pm_flags, (size_t)pm_need));
        m->pm_cap = pm_need;
    }
    if (!m->pm_count) HIP_TRY(dalloc(&m->pm_count, 2));
    if (owner_need > m->owner_cap) {
        m->owner_cap = 0;
        HIP_TRY(drealloc(&m->pixel_owner, owner_need));
        m->owner_cap = owner_need;
    }
    return SVH_OK;
}

// M1..M5  Matcher::computeFeatures   matcher.cpp:780-878, in two steps: the host packs the rows, then the
// device work is enqueued (or, in a batched call, recorded)
static int features_pack(svh_matcher* m, DevView& V, int cam, const uint8_t* src, int32_t pitch) {
    (void)m; (void)cam;
    // rows are packed at the aligned pitch into the view's pinned buffer (the ring buffer double-buffers it:
    // the previous frame's copy stays valid for getGain), then one linear DMA
    for (int32_t v = 0; v < V.h; v++) {
        uint8_t* row = V.stage + (size_t)v * V.bpl;
        memcpy(row, src + (size_t)v * pitch, V.w);
        memset(row + V.w, 0, V.bpl - V.w);
    }
    return SVH_OK;
}

static int features_enqueue(svh_matcher* m, DevView& V, int cam, double* tf, bool uploaded = false,
                            int32_t* counts = nullptr, hipStream_t on = nullptr) {
    const svh_matcher_params& p = m->p;
    hipStream_t s = on ? on : (cam == 1 ? m->stream2 : m->stream);
    const size_t fn = (size_t)V.bpl * V.h;
    const uint8_t* stage = V.stage;
    auto ftick = [&](int i) { if (g_mtiming && tf) tf[i] = mnow_ms(); };
    ftick(1);
    if (uploaded)
        ;   // (a lockstep call: issued by the thread that packed the image)
    else if (fn % 16 == 0)
        mlaunch_upload(s, stage, V.I, fn);
    else
        mlaunch_copy(s, V.I, stage, fn, hipMemcpyHostToDevice);   // (bpl is a multiple of 16: not taken)
    ftick(2);
    const uint8_t* Im = V.I;
    if (p.half_resolution) {
        mlaunch_half_filters(s, V.I, V.w, V.h, V.bpl, V.Ih, V.mw, V.mh, V.mbpl, V.du_full, V.dv_full);
        Im = V.Ih;
    }
    mlaunch_filters(s, Im, V.mw, V.mh, V.mbpl, V.du, V.dv, V.f1, V.f2);
    ftick(3);
    const int32_t scale = p.half_resolution ? 2 : 1;
    int32_t ns = p.nms_n * 3;
    if (ns > 10) ns = std::max(p.nms_n, 10);
    int rc = ensure_feature_scratch(m, std::max(V.cap[0], V.cap[1]));
    if (rc) return rc;
    if (p.multi_stage) {
        mlaunch_features2(s, V.f1, V.f2, V.du, V.dv, V.mw, V.mh, V.mbpl, p.nms_tau, m->margin, scale,
                          ns, m->slots[2 + cam], m->flags[2 + cam], m->order[2 + cam], V.tab[0], V.cnt + 0,
                          p.nms_n, m->slots[cam], m->flags[cam], m->order[cam], V.tab[1], V.cnt + 1,
                          (counts ? counts : m->h_n) + 2 * cam);
    } else {
        mlaunch_fill(s, V.cnt, 0, sizeof(int32_t));
        mlaunch_features(s, V.f1, V.f2, V.du, V.dv, V.mw, V.mh, V.mbpl, p.nms_n, p.nms_tau, m->margin, scale,
                         m->slots[cam], m->flags[cam], m->order[cam], V.tab[1], V.cnt + 1);
    }
    ftick(4);
    // feature counts come back through pinned memory after BOTH cameras are enqueued
    // (a copy into pageable memory would block here until this camera's kernels finish; with both tables the
    // compaction launch writes them there itself)
    if (!p.multi_stage)
        mlaunch_copy(s, (counts ? counts : m->h_n) + 2 * cam, V.cnt, 2 * sizeof(int32_t), hipMemcpyDeviceToHost);
    ftick(5);
    V.nbins = 0;   // bin indices are (re)built by matchFeatures for the current bin grid
    V.valid = true;
    return SVH_OK;   // the caller synchronises once after both cameras
}

static int compute_features(svh_matcher* m, DevView& V, int cam, const uint8_t* src, int32_t pitch) {
    double tf[6] = {0, 0, 0, 0, 0, 0};
    if (g_mtiming) tf[0] = mnow_ms();
    int rc = features_pack(m, V, cam, src, pitch);
    if (rc) return rc;
    rc = features_enqueue(m, V, cam, tf);
    if (rc) return rc;
    if (g_mtiming)
        for (int i = 0; i < 5; i++) m->tfine[i] += tf[i + 1] - tf[i];
    return SVH_OK;
}

// (re)build the bin indices of every view whose tables changed since the last call: one launch
static int ensure_bins(svh_matcher* m, DevView* const* views, int nviews, int32_t ub, int32_t vb) {
    const int32_t nb = 4 * ub * vb;
    BinJobs J;
    int nj = 0, nmax = 0;
    for (int v = 0; v < nviews; v++) {
        DevView& V = *views[v];
        if (!V.valid || V.nbins == nb) continue;
        if (nb > V.off_cap) {   // (re)allocate only when the bin grid grows: hipFree synchronises the device
            V.off_cap = 0;
            for (int k = 0; k < 2; k++) HIP_TRY(drealloc(&V.off[k], (size_t)nb + 1));
            V.off_cap = nb;
        }
        for (int k = 0; k < 2; k++) {
            J.table[nj] = V.tab[k];
            J.count[nj] = V.cnt + k;
            J.off[nj] = V.off[k];
            J.ids[nj] = V.ids[k];
            nmax = std::max(nmax, V.n[k]);
            nj++;
        }
        V.nbins = nb;
    }
    if (!nj) return SVH_OK;
    if (nb > m->cursor_cap) {
        m->cursor_cap = 0;
        HIP_TRY(drealloc(&m->cursor, (size_t)nb));
        m->cursor_cap = nb;
    }
    mlaunch_bin_index(m->stream, J, nj, nmax, ub, vb, m->p.match_binsize, m->cursor);
    return SVH_OK;
}

// the triangulation of a large vote runs on 2^depth threads when this is the only sequence on the GPU's host side
// (several callers at once = several sequences: their own threads already fill the cores, the helper pool would only
// be fought over; inside a batch call the pool of the batch is the parallelism)
static int vote_par_depth() {
    static const int par = svh::env("SVH_DELAUNAY_PAR") ? atoi(svh::env("SVH_DELAUNAY_PAR")) : 3;
    const bool alone = g_active_callers.load(std::memory_order_relaxed) <= 2 && !t_in_batch;
    return alone ? std::max(0, std::min(par, 3)) : 0;
}
static const int32_t kVoteParMin = 1500;

// M9  Matcher::removeOutliers   matcher.cpp:1383-1570 (host: Delaunay + edge votes)
static int remove_outliers(const svh_matcher_params& p, std::vector<svh_p_match>& pm, int32_t method) {
    if (pm.size() <= 3) return SVH_OK;
    const int32_t n = (int32_t)pm.size();
    std::vector<float> pts((size_t)2 * n);
    for (int32_t i = 0; i < n; i++) {
        pts[2 * i] = pm[i].u1c;
        pts[2 * i + 1] = pm[i].v1c;
    }
    std::vector<int32_t> tri((size_t)3 * (2 * n + 16));
    // the Matcher is a single-stream, latency-bound path: large votes triangulate on helper threads
    const int par = n >= kVoteParMin ? vote_par_depth() : 0;
    const int32_t nt = delaunay(pts.data(), n, tri.data(), 2 * n + 16, par, /*expect_dups=*/true);
    if (nt < 0) return mfail(SVH_ERR_UNSUPPORTED, "outlier triangulation failed");
    const float ft = (float)p.outlier_flow_tolerance, dt = (float)p.outlier_disp_tolerance;
    // support of a vertex = its triangle edges whose endpoints agree (an edge counts once per triangle it bounds)
    auto count = [&](int32_t t0, int32_t t1, int32_t* votes) {
        for (int32_t t = t0; t < t1; t++) {
            const int32_t* c = &tri[3 * t];
            static const int e[3][2] = {{0, 1}, {1, 2}, {0, 2}};
            for (int k = 0; k < 3; k++) {
                const svh_p_match& a = pm[c[e[k][0]]];
                const svh_p_match& b = pm[c[e[k][1]]];
                bool ok = true;
                if (method == 1) {
                    ok = fabsf((a.u1c - a.u2c) - (b.u1c - b.u2c)) < dt;
                } else {
                    if (method == 2) ok = fabsf((a.u1p - a.u2p) - (b.u1p - b.u2p)) < dt;
                    const float fu = (a.u1c - a.u1p) - (b.u1c - b.u1p), fv = (a.v1c - a.v1p) - (b.v1c - b.v1p);
                    ok = ok && fabsf(fu) + fabsf(fv) < ft;
                }
                if (ok) {
                    votes[c[e[k][0]]]++;
                    votes[c[e[k][1]]]++;
                }
            }
        }
    };
    const int parts = par > 0 ? 4 : 1;
    std::vector<int32_t> votes((size_t)n * parts, 0);
    if (parts == 1) {
        count(0, nt, votes.data());
    } else {   // (the helpers are still awake from the triangulation)
        run_many(parts, [&](int i) { count((int32_t)((int64_t)nt * i / parts), (int32_t)((int64_t)nt * (i + 1) / parts), &votes[(size_t)n * i]); });
        for (int k = 1; k < parts; k++)
            for (int32_t i = 0; i < n; i++) votes[i] += votes[(size_t)n * k + i];
    }
    size_t w = 0;
    for (int32_t i = 0; i < n; i++)
        if (votes[i] >= 4) pm[w++] = pm[i];
    pm.resize(w);
    return SVH_OK;
}

// M10  Matcher::computePriorStatistics   matcher.cpp:882-1032
static void prior_statistics(svh_matcher* m, const std::vector<svh_p_match>& pm, int32_t method,
                             int32_t ub, int32_t vb) {
    const svh_matcher_params& p = m->p;
    const int32_t stages = method == 2 ? 4 : 2, nb = ub * vb;
    const float big = 1000000.f;
    std::vector<float> lo((size_t)nb * 8, big), hi((size_t)nb * 8, -big);
    std::vector<uint8_t> seen(nb, 0);
    const float bs = (float)p.match_binsize;
    for (const svh_p_match& it : pm) {
        float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (method == 0) {
            d[0] = it.u1p - it.u1c; d[1] = it.v1p - it.v1c; d[2] = it.u1c - it.u1p; d[3] = it.v1c - it.v1p;
        } else if (method == 1) {
            d[0] = it.u2c - it.u1c; d[2] = it.u1c - it.u2c;
        } else {
            d[0] = it.u2p - it.u1p;
            d[2] = it.u2c - it.u2p; d[3] = it.v2c - it.v2p;
            d[4] = it.u1c - it.u2c;
            d[6] = it.u1p - it.u1c; d[7] = it.v1p - it.v1c;
        }
        const float ru = method < 2 ? it.u1c : it.u1p, rv = method < 2 ? it.v1c : it.v1p;
        const int32_t cu = (int32_t)floorf(ru / bs), cv = (int32_t)floorf(rv / bs);
        auto cl = [](int32_t x, int32_t n) { return std::min(std::max(x, 0), n - 1); };
        for (int32_t vbn = cl(cv - 1, vb); vbn <= cl(cv + 1, vb); vbn++)
            for (int32_t ubn = cl(cu - 1, ub); ubn <= cl(cu + 1, ub); ubn++) {
                const size_t b = (size_t)vbn * ub + ubn;
                seen[b] = 1;
                for (int k = 0; k < stages * 2; k++) {
                    if (d[k] < lo[b * 8 + k]) lo[b * 8 + k] = d[k];
                    if (d[k] > hi[b * 8 + k]) hi[b * 8 + k] = d[k];
                }
            }
    }
    m->ranges.assign((size_t)nb * 16, 0.f);
    for (int32_t b = 0; b < nb; b++)
        for (int i = 0; i < stages; i++)
            for (int ax = 0; ax < 2; ax++) {
                float mn = seen[b] ? lo[(size_t)b * 8 + 2 * i + ax] : (float)-p.match_radius;
                float mx = seen[b] ? hi[(size_t)b * 8 + 2 * i + ax] : (float)+p.match_radius;
                const float span = mx - mn;
                if (span < 20) {   // search window of at least 20 px
                    mn -= ceilf((20 - span) / 2);
                    mx += ceilf((20 - span) / 2);
                }
                // layout: u_min[4], u_max[4], v_min[4], v_max[4]
                m->ranges[(size_t)b * 16 + (ax ? 8 : 0) + i] = mn;
                m->ranges[(size_t)b * 16 + (ax ? 12 : 4) + i] = mx;
            }
}

static FeatView view_of(const DevView& V, int dense) {
    FeatView f;
    f.rec = V.tab[dense];
    f.count = V.cnt + dense;
    f.off = V.off[dense];
    f.ids = V.ids[dense];
    return f;
}

static SobelView sobel_of(const DevView& V, bool half) {
    SobelView s;
    s.du = half ? V.du_full : V.du;
    s.dv = half ? V.dv_full : V.dv;
    s.w = V.w; s.h = V.h; s.bpl = V.bpl;
    return s;
}

// one matching() pass on the device, in steps a batched call can interleave over its objects:
// match_enqueue -> [refine_enqueue] -> download_enqueue -> (stream wait) -> match_collect
struct MatchPass {
    bool count_on_host = false;   // the compaction wrote the count to h_cnt itself
    int32_t nq = 0;
    const svh_p_match* result = nullptr;
    const int32_t* result_count = nullptr;
};

static int match_enqueue(svh_matcher* m, int dense, int32_t method, bool use_prior, const double* Tr, MatchPass& mp) {
    const svh_matcher_params& p = m->p;
    MatchParams P;
    memset(&P, 0, sizeof(P));
    P.method = method;
    P.width = m->dims_c[0];
    P.height = m->dims_c[1];
    P.ub = (int32_t)ceilf((float)m->dims_c[0] / (float)p.match_binsize);
    P.vb = (int32_t)ceilf((float)m->dims_c[1] / (float)p.match_binsize);
    P.binsize = p.match_binsize;
    P.match_radius = p.match_radius;
    P.match_disp_tolerance = p.match_disp_tolerance;
    P.has_tr = Tr ? 1 : 0;
    P.f = p.f; P.cu = p.cu; P.cv = p.cv; P.base = p.base;
    if (Tr) memcpy(P.tr, Tr, sizeof(P.tr));
    const DevView& q = method == 2 ? m->prev[0] : m->cur[0];
    const int32_t nq = q.n[dense];
    int rc = ensure_match_scratch(m, std::max(nq, 1), method < 2 ? (size_t)P.width * P.height : 0);
    if (rc) return rc;
    if (nq > m->h_pm_cap || !m->h_cnt) {
        m->h_pm_cap = 0;
        HIP_TRY(hrealloc(&m->h_pm, (size_t)std::max(nq, 1) * sizeof(svh_p_match)));
        if (!m->h_cnt) HIP_TRY(hipHostMalloc((void**)&m->h_cnt, sizeof(int32_t)));
        m->h_pm_cap = std::max(nq, 1);
    }
    mlaunch_match(m->stream, P, view_of(m->prev[0], dense), view_of(m->prev[1], dense), view_of(m->cur[0], dense),
                  view_of(m->cur[1], dense), nq, m->ranges_dev, use_prior ? 1 : 0, m->pm_slots, m->pm_flags,
                  m->pixel_owner, m->pm_out, m->pm_count, m->h_cnt);
    mp.count_on_host = t_rec == nullptr;   // (a recorded batch compacts through the job table: its count is copied)
    mp.nq = nq;
    mp.result = m->pm_out;
    mp.result_count = m->pm_count;
    return SVH_OK;
}

static void refine_enqueue(svh_matcher* m, int32_t method, MatchPass& mp) {
    const svh_matcher_params& p = m->p;
    const bool half = p.half_resolution != 0;
    const int parabolic = p.refinement == 2;
    mlaunch_refine(m->stream, m->pm_out, m->pm_count, mp.nq, method, m->margin, sobel_of(m->prev[0], half),
                   sobel_of(m->prev[1], half), sobel_of(m->cur[0], half), sobel_of(m->cur[1], half), parabolic,
                   m->pm_flags, m->pm_slots, m->pm_count + 1);
    if (parabolic) {
        mp.result = m->pm_slots;
        mp.result_count = m->pm_count + 1;
        mp.count_on_host = false;
    }
}

static void download_enqueue(svh_matcher* m, const MatchPass& mp) {
    if (!mp.count_on_host) mlaunch_copy(m->stream, m->h_cnt, mp.result_count, sizeof(int32_t), hipMemcpyDeviceToHost);
    if (mp.nq > 0)
        mlaunch_copy(m->stream, m->h_pm, mp.result, (size_t)mp.nq * sizeof(svh_p_match), hipMemcpyDeviceToHost);
}

static void match_collect(svh_matcher* m, const MatchPass& mp, std::vector<svh_p_match>& dst) {
    const int32_t count = std::max(0, std::min(*m->h_cnt, mp.nq));
    dst.assign(m->h_pm, m->h_pm + count);
}

static int run_matching(svh_matcher* m, int dense, int32_t method, bool use_prior, const double* Tr,
                        std::vector<svh_p_match>& out, bool refine, std::vector<svh_p_match>* raw_tap) {
    MatchPass mp;
    const bool timed = g_mtiming;
    if (timed) tev_record(m, dense ? 6 : 4, m->stream);
    int rc = match_enqueue(m, dense, method, use_prior, Tr, mp);
    if (rc) return rc;
    auto download = [&](std::vector<svh_p_match>& dst) -> int {
        download_enqueue(m, mp);
        if (timed) tev_record(m, dense ? 7 : 5, m->stream);
        if (m->warm_on_wait) {   // (costs this thread a few futex wakes: paid while the device works)
            m->warm_on_wait = false;
            helpers_warm(7, 1500);
        }
        HIP_TRY((hipError_t)wait_stream(m->stream));
        HIP_TRY(hipGetLastError());
        match_collect(m, mp, dst);
        return SVH_OK;
    };
    if (refine && raw_tap && m->taps) {
        rc = download(*raw_tap);
        if (rc) return rc;
    }
    if (refine) refine_enqueue(m, method, mp);
    rc = download(out);
    if (rc) return rc;
    if (timed) m->tdev[dense ? 2 : 1] += tev_ms(m, dense ? 6 : 4, dense ? 7 : 5);
    if (!refine && raw_tap && m->taps) *raw_tap = out;
    return SVH_OK;
}

// Matcher::bucketFeatures   matcher.cpp:297-343; draws from `rs` (libc rand() or the caller's private stream)
int32_t bucket_features(svh_matcher* m, int32_t max_features, float bw, float bh, RandStream& rs) {
    if (!m) return 0;
    float u_max = 0, v_max = 0;
    for (const svh_p_match& q : m->m2) {
        if (q.u1c > u_max) u_max = q.u1c;
        if (q.v1c > v_max) v_max = q.v1c;
    }
    const int32_t cols = (int32_t)floorf(u_max / bw) + 1, rows = (int32_t)floorf(v_max / bh) + 1;
    // counting sort into the buckets (stable: the order inside a bucket is the order of the match list)
    const size_t nb = (size_t)cols * rows, n = m->m2.size();
    std::vector<int32_t> start(nb + 1, 0), cell(n);
    for (size_t i = 0; i < n; i++) {
        const svh_p_match& q = m->m2[i];
        cell[i] = (int32_t)floorf(q.v1c / bh) * cols + (int32_t)floorf(q.u1c / bw);
        start[cell[i] + 1]++;
    }
    for (size_t b = 0; b < nb; b++) start[b + 1] += start[b];
    std::vector<int32_t> order(n), fill(start.begin(), start.end() - 1);
    for (size_t i = 0; i < n; i++) order[fill[cell[i]]++] = (int32_t)i;
    std::vector<svh_p_match> kept;
    for (size_t b = 0; b < nb; b++) {
        int32_t* o = order.data() + start[b];
        const size_t len = (size_t)(start[b + 1] - start[b]);
        // same shuffle as std::random_shuffle(first,last) of libstdc++: rand() % (i+1)
        for (size_t i = 1; i < len; i++) std::swap(o[i], o[(size_t)rs.next() % (i + 1)]);
        for (size_t i = 0; i < len && (int32_t)i < max_features; i++) kept.push_back(m->m2[o[i]]);
    }
    m->m2.swap(kept);
    return (int32_t)m->m2.size();
}

void batch_parallel_for(int n, const std::function<void(int)>& fn) {
    const bool was = t_in_batch;
    t_in_batch = true;
    BatchPool::get().parallel_for(n, [&](int i) {
        const bool w = t_in_batch;
        t_in_batch = true;
        fn(i);
        t_in_batch = w;
    });
    t_in_batch = was;
}

}  // namespace svh

extern "C" {

void svh_matcher_params_default(svh_matcher_params* p) {
    // Matcher::parameters::parameters()   libviso2/src/matcher.h:56-68
    p->nms_n = 3;
    p->nms_tau = 50;
    p->match_binsize = 50;
    p->match_radius = 200;
    p->match_disp_tolerance = 2;
    p->outlier_disp_tolerance = 5;
    p->outlier_flow_tolerance = 5;
    p->multi_stage = 1;
    p->half_resolution = 1;
    p->refinement = 1;
    p->f = p->cu = p->cv = p->base = 0;
}

svh_matcher* svh_matcher_create(const svh_matcher_params* p) {
    if (!p) return nullptr;
    svh::ensure_init();
    svh_matcher* m = new svh_matcher();
    m->p = *p;
    m->margin = 8 + 1;                                      // matcher.cpp:56
    if (p->half_resolution) m->p.match_radius /= 2;         // matcher.cpp:59-62
    m->device = 0;
    (void)hipGetDevice(&m->device);
    m->stream = nullptr;
    memset(m->dims_p, 0, sizeof(m->dims_p));
    memset(m->dims_c, 0, sizeof(m->dims_c));
    return m;
}

void svh_matcher_set_timing(int32_t on) { g_mtiming_api.store(on ? 1 : 0); }

int32_t svh_matcher_get_timing(svh_matcher* m, const char** names, double* ms, int32_t cap, int32_t reset) {
    static const char* kNames[10] = {
        "pushBack: pack + enqueue (host)", "pushBack: wait for the device (host)", "matchFeatures: sparse matching (host wall, waits for the device)",
        "matchFeatures: sparse outlier vote (host)", "matchFeatures: prior statistics (host)",
        "matchFeatures: dense matching + refinement (host wall, waits for the device)", "matchFeatures: dense outlier vote (host)",
        "device: pushBack kernels", "device: sparse matching", "device: dense matching + refinement"};
    if (!m) return 0;
    const double a = m->tcalls[0] ? 1.0 / (double)m->tcalls[0] : 0, b = m->tcalls[1] ? 1.0 / (double)m->tcalls[1] : 0;
    const double v[10] = {m->tacc[T_PACK] * a, m->tacc[T_PUSH_GPU] * a, m->tacc[T_SPARSE] * b, m->tacc[T_OUT1] * b,
                          m->tacc[T_PRIOR] * b, m->tacc[T_DENSE] * b, m->tacc[T_OUT2] * b,
                          m->tdev[0] * a, m->tdev[1] * b, m->tdev[2] * b};
    int32_t n = 0;
    for (; n < 10 && n < cap; n++) {
        if (names) names[n] = kNames[n];
        if (ms) ms[n] = v[n];
    }
    if (reset) {
        for (double& x : m->tacc) x = 0;
        for (double& x : m->tdev) x = 0;
        m->tcalls[0] = m->tcalls[1] = 0;
    }
    return n;
}

void svh_matcher_destroy(svh_matcher* m) {
    if (!m) return;
    if (g_mtiming && m->tcalls[0] && m->tcalls[1]) {
        const double a = 1.0 / (double)m->tcalls[0], b = 1.0 / (double)m->tcalls[1];
        fprintf(stderr, "[svh matcher timing] pushBack: pack+enqueue %.3f ms, gpu wait %.3f ms | matchFeatures: "
                        "sparse %.3f, outliers %.3f, prior %.3f, dense+refine %.3f, outliers %.3f ms\n",
                m->tacc[T_PACK] * a, m->tacc[T_PUSH_GPU] * a, m->tacc[T_SPARSE] * b, m->tacc[T_OUT1] * b,
                m->tacc[T_PRIOR] * b, m->tacc[T_DENSE] * b, m->tacc[T_OUT2] * b);
        fprintf(stderr, "[svh matcher timing] pushBack per frame: pack %.3f, h2d enqueue %.3f, filters enqueue %.3f, "
                        "features enqueue %.3f, d2h enqueue %.3f ms\n",
                m->tfine[0] * a, m->tfine[1] * a, m->tfine[2] * a, m->tfine[3] * a, m->tfine[4] * a);
    }
    for (auto& e : m->tev)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (m->next_job.valid()) (void)m->next_job.get();   // a hand-over still running on the prefetch thread
    if (m->stream) {
        (void)hipSetDevice(m->device);
        (void)hipStreamSynchronize(m->stream);
        if (m->stream2) (void)hipStreamSynchronize(m->stream2);
        for (int k = 0; k < 2; k++) {
            m->prev[k].release();
            m->cur[k].release();
            m->next[k].release();
        }
        for (int c = 0; c < 4; c++) {
            (void)hipFree(m->slots[c]); (void)hipFree(m->flags[c]); (void)hipFree(m->order[c]);
        }
        (void)hipFree(m->cursor);
        (void)hipFree(m->pm_slots);
        (void)hipFree(m->pm_out); (void)hipFree(m->pm_flags); (void)hipFree(m->pm_count);
        (void)hipFree(m->pixel_owner); (void)hipFree(m->ranges_dev);
        (void)hipHostFree(m->h_pm); (void)hipHostFree(m->h_cnt);
        (void)hipHostFree(m->h_n);
        (void)hipHostFree(m->h_ranges);
        (void)hipStreamDestroy(m->stream);
        if (m->stream2) (void)hipStreamDestroy(m->stream2);
    }
    delete m;
}

void svh_matcher_set_intrinsics(svh_matcher* m, double f, double cu, double cv, double base) {
    if (!m) return;
    m->p.f = f; m->p.cu = cu; m->p.cv = cv; m->p.base = base;
}

// pushBack up to the device work: argument checks, ring-buffer rotation, buffers.  `src` = the two rows of images
// takes the prefetched frame: waits for its device work, rotates it into `cur`
static int32_t push_take_prefetched(svh_matcher* m, int32_t replace, bool stream_waited = false) {
    if (m->next_job.valid()) {
        const int32_t rc = m->next_job.get();
        m->next_job = std::shared_future<int32_t>();
        if (rc) {
            m->has_next = false;
            return mfail(rc, "the hand-over of the prefetched frame failed on the prefetch thread: " + (m->next_why ? *m->next_why : std::string()));
        }
    }
    if (!stream_waited) {
        HIP_TRY(hipSetDevice(m->device));
        HIP_TRY((hipError_t)wait_stream(m->next_stream));
        HIP_TRY(hipGetLastError());
    }
    for (int k = 0; k < 2; k++) {
        if (!replace) {
            // ring buffer: current -> previous, prefetched -> current; the old previous buffers are recycled
            std::swap(m->prev[k], m->cur[k]);
            std::swap(m->cur[k], m->next[k]);
        } else {
            std::swap(m->cur[k], m->next[k]);
        }
        m->next[k].valid = false;
        if (k >= m->next_cams) m->cur[k].valid = false;
    }
    if (!replace) memcpy(m->dims_p, m->dims_c, sizeof(m->dims_p));
    memcpy(m->dims_c, m->dims_n, sizeof(m->dims_c));
    for (int k = 0; k < m->next_cams; k++) {
        m->cur[k].n[0] = m->h_n[4 + 2 * k];
        m->cur[k].n[1] = m->h_n[4 + 2 * k + 1];
    }
    m->has_next = false;
    return SVH_OK;
}

static int32_t push_prepare(svh_matcher* m, const uint8_t* I1, const uint8_t* I2, const int32_t* dims, int32_t replace) {
    if (!m || !dims) return mfail(SVH_ERR_BAD_ARG, "null argument");
    if (m->has_next) return mfail(SVH_ERR_BAD_ARG, "a prefetched frame is pending: pass no images to take it");
    const int32_t w = dims[0], h = dims[1], pitch = dims[2];
    if (w <= 0 || h <= 0 || pitch < w || I1 == 0) {
        // matcher.cpp:110-114
        fprintf(stderr, "ERROR: Image dimension mismatch!\n");
        return mfail(SVH_ERR_BAD_DIMS, "image dimension mismatch");   // (its own code: callers that mimic the reference ignore THIS one only)
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return mfail(SVH_ERR_NO_DEVICE, "no HIP device visible: libsvhip has no CPU fallback");
    HIP_TRY(hipSetDevice(m->device));
    if (!m->stream) HIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    if (!m->stream2) HIP_TRY(hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking));
    if (!replace) {
        // ring buffer: current -> previous; the old previous buffers are recycled
        for (int k = 0; k < 2; k++) {
            std::swap(m->prev[k], m->cur[k]);
            m->cur[k].valid = false;
        }
        memcpy(m->dims_p, m->dims_c, sizeof(m->dims_p));
    } else {
        m->cur[0].valid = m->cur[1].valid = false;
    }
    m->dims_c[0] = w;
    m->dims_c[1] = h;
    m->dims_c[2] = w + 16 - w % 16;   // +16 even when w % 16 == 0 (matcher.cpp:173)
    if (!m->h_n) HIP_TRY(hipHostMalloc((void**)&m->h_n, 8 * sizeof(int32_t)));   // [4..7]: a prefetched frame's
    const uint8_t* src[2] = {I1, I2};
    for (int k = 0; k < 2; k++) {
        if (!src[k]) continue;
        int rc = ensure_view(m, m->cur[k], w, h, m->dims_c[2]);
        if (rc) return rc;
    }
    return SVH_OK;
}

static void push_finish(svh_matcher* m, const uint8_t* I1, const uint8_t* I2) {
    const uint8_t* src[2] = {I1, I2};
    for (int k = 0; k < 2; k++)
        if (src[k]) {
            m->cur[k].n[0] = m->h_n[2 * k];
            m->cur[k].n[1] = m->h_n[2 * k + 1];
        }
}

int32_t svh_matcher_push_back(svh_matcher* m, const uint8_t* I1, const uint8_t* I2, const int32_t* dims,
                              int32_t replace) {
    svh::ActiveCaller active_;
    if (m && !I1 && !I2 && m->has_next) return push_take_prefetched(m, replace);
    int32_t rc = push_prepare(m, I1, I2, dims, replace);
    if (rc) return rc;
    const uint8_t* src[2] = {I1, I2};
    const bool timed = g_mtiming;
    const double t0 = timed ? mnow_ms() : 0;
    for (int k = 0; k < 2; k++) {
        if (!src[k]) continue;
        if (timed) tev_record(m, 2 * k, k == 1 ? m->stream2 : m->stream);
        rc = compute_features(m, m->cur[k], k, src[k], dims[2]);
        if (rc) return rc;
        if (timed) tev_record(m, 2 * k + 1, k == 1 ? m->stream2 : m->stream);
    }
    const double t1 = timed ? mnow_ms() : 0;
    HIP_TRY((hipError_t)wait_stream(m->stream));
    HIP_TRY((hipError_t)wait_stream(m->stream2));
    HIP_TRY(hipGetLastError());
    push_finish(m, I1, I2);
    if (timed) {
        m->tacc[T_PACK] += t1 - t0;
        m->tacc[T_PUSH_GPU] += mnow_ms() - t1;
        m->tcalls[0]++;
        // (the two cameras run on two streams side by side: the longer one is the phase's device time)
        const double a = src[0] ? tev_ms(m, 0, 1) : 0, b = src[1] ? tev_ms(m, 2, 3) : 0;
        m->tdev[0] += a > b ? a : b;
    }
    return SVH_OK;
}

}  // extern "C"

namespace svh {
// host side of svh_matcher_prefetch_batch (prefetch thread): buffers, packing + uploads, feature extraction issued
static int32_t prefetch_body(const std::vector<svh_matcher*>& ms, const std::vector<const uint8_t*>& I1,
                             const std::vector<const uint8_t*>& I2, int32_t w, int32_t h, int32_t pitch, int ncam,
                             bool lockstep) {
    const int K = (int)ms.size();
    for (int i = 0; i < K; i++) {
        svh_matcher* m = ms[i];
        HIP_TRY(hipSetDevice(m->device));
        if (!m->stream) HIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
        if (!m->stream2) HIP_TRY(hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking));
        if (!m->h_n) HIP_TRY(hipHostMalloc((void**)&m->h_n, 8 * sizeof(int32_t)));
        m->dims_n[0] = w;
        m->dims_n[1] = h;
        m->dims_n[2] = w + 16 - w % 16;
        for (int k = 0; k < ncam; k++) {
            const int rc = ensure_view(m, m->next[k], w, h, m->dims_n[2]);
            if (rc) return rc;
        }
    }
    // one stream for the whole prefetch when the objects run in lockstep, else each object's second stream
    // The lockstep hand-over runs on a stream of the prefetch thread's own recorder: it outlives every Matcher, so the
    // recorder may wait for it before it reuses its arena whatever happened to the objects of the last hand-over.
    BatchRec& pr = prefetch_recorder(ms[0]->device);
    HIP_TRY(pr.ensure_side());
    HIP_TRY(pr.reuse());
    hipStream_t const pf_own = pr.side[0];
    std::vector<int> rcs((size_t)K * ncam, 0);
    // Lockstep hand-over: the K * ncam image uploads are recorded with the features (one k_upload_b launch per camera
    // over the K objects instead of one k_upload per image -- 32 launches a call at K = 16); the packing threads then
    // launch nothing.  SVH_UPLOAD_BATCH=0: every packing thread launches its image's upload as soon as it is packed.
    static const bool upload_batch = !(svh::env("SVH_UPLOAD_BATCH") && atoi(svh::env("SVH_UPLOAD_BATCH")) == 0);
    const bool upload_recorded = lockstep && upload_batch;
    batch_parallel_for(K * ncam, [&](int j) {
        (void)hipSetDevice(ms[0]->device);
        svh_matcher* m = ms[j / ncam];
        const int cam = j % ncam;
        DevView& V = m->next[cam];
        rcs[j] = features_pack(m, V, cam, cam ? I2[j / ncam] : I1[j / ncam], pitch);
        if (!rcs[j] && !upload_recorded) {
            // (all on the stream the features follow on: this work is hidden behind the frame before, so the
            // uploads need not overlap each other -- and the hand-over needs no events)
            mlaunch_upload(lockstep ? pf_own : m->stream2, V.stage, V.I, (size_t)V.bpl * V.h);
        }
    });
    for (int rc : rcs)
        if (rc) return rc;
    int rc = SVH_OK;
    if (lockstep) {
        hipStream_t pf = pf_own;
        pr.reset();
        t_rec = &pr;
        for (int i = 0; i < K && !rc; i++) {
            pr.begin_object();
            for (int cam = 0; cam < ncam && !rc; cam++)
                rc = features_enqueue(ms[i], ms[i]->next[cam], cam, nullptr, !upload_recorded, ms[i]->h_n + 4);
        }
        t_rec = nullptr;
        if (rc) return rc;
        if (pr.broken) {
            pr.reset();
            lockstep = false;   // (not reachable with equal parameters and sizes) issue them one by one below
            if (upload_recorded)
                for (int i = 0; i < K; i++)
                    for (int cam = 0; cam < ncam; cam++) {
                        DevView& V = ms[i]->next[cam];
                        mlaunch_upload(ms[i]->stream2, V.stage, V.I, (size_t)V.bpl * V.h);
                    }
            HIP_TRY(hipStreamSynchronize(pf));
        } else {
            HIP_TRY(pr.flush(pf));
            for (int i = 0; i < K; i++) ms[i]->next_stream = pf;
        }
    }
    if (!lockstep) {
        for (int i = 0; i < K; i++) {
            for (int cam = 0; cam < ncam; cam++) {
                rc = features_enqueue(ms[i], ms[i]->next[cam], cam, nullptr, true, ms[i]->h_n + 4, ms[i]->stream2);
                if (rc) return rc;
            }
            ms[i]->next_stream = ms[i]->stream2;
        }
    }
    return SVH_OK;
}
}  // namespace svh

extern "C" {

// The NEXT frame of K Matchers handed over early: packed, uploaded and its features computed into a third view
// set on the objects' second streams, without waiting -- so that this device and host work overlaps the
// matchFeatures / motion estimate of the frame before it.  The following pushBack of these objects (single or
// batch entry) is called WITHOUT images and takes the prefetched frame.  One prefetched frame per object at a time.
int32_t svh_matcher_prefetch_batch(svh_matcher* const* ms, int32_t K, const uint8_t* const* I1,
                                   const uint8_t* const* I2, const int32_t* dims) {
    if (!ms || K <= 0 || !I1 || !dims) return mfail(SVH_ERR_BAD_ARG, "null argument");
    const int32_t w = dims[0], h = dims[1], pitch = dims[2];
    if (w <= 0 || h <= 0 || pitch < w) return mfail(SVH_ERR_BAD_DIMS, "image dimension mismatch");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return mfail(SVH_ERR_NO_DEVICE, "no HIP device visible: libsvhip has no CPU fallback");
    const int ncam = (I2 && I2[0]) ? 2 : 1;
    bool lockstep = K > 1;
    for (int i = 0; i < K; i++) {
        if (!ms[i] || !I1[i] || (ncam == 2 && !I2[i])) return mfail(SVH_ERR_BAD_ARG, "null matcher or image in the batch");
        if (ms[i]->has_next) return mfail(SVH_ERR_BAD_ARG, "a prefetched frame is already pending");
        for (int j = 0; j < i; j++)
            if (ms[j] == ms[i]) return mfail(SVH_ERR_BAD_ARG, "the same matcher twice in one batch");
        lockstep = lockstep && memcmp(&ms[i]->p, &ms[0]->p, sizeof(ms[0]->p)) == 0 && ms[i]->device == ms[0]->device;
    }
    // the rest runs on the prefetch thread (the caller's pointer arrays are copied, the images are read there:
    // they must stay unchanged until the frame is taken)
    std::vector<svh_matcher*> mv(ms, ms + K);
    std::vector<const uint8_t*> a(I1, I1 + K), b2;
    if (ncam == 2) b2.assign(I2, I2 + K);
    std::shared_ptr<std::string> why = std::make_shared<std::string>();
    std::shared_future<int32_t> job = PrefetchWorker::get().post([mv, a, b2, w, h, pitch, ncam, lockstep, why]() {
        const int32_t rc = prefetch_body(mv, a, b2, w, h, pitch, ncam, lockstep);
        if (rc) *why = svh_last_error();   // (this thread's message: handed to the thread that takes the frame)
        return rc;
    });
    for (int i = 0; i < K; i++) {
        ms[i]->has_next = true;
        ms[i]->next_cams = ncam;
        ms[i]->next_job = job;
        ms[i]->next_why = why;
    }
    return SVH_OK;
}

// K Matchers in lockstep (one frame of K sequences): the host packs the K x 2 images on the helper threads,
// the device work of all K is recorded and issued as ONE launch per kernel (batch_rec.h) on ms[0]'s stream.
// Results are those of K svh_matcher_push_back calls.  Objects must share parameters and image size;
// otherwise (or with taps on) the call runs them one after the other.
int32_t svh_matcher_push_back_batch(svh_matcher* const* ms, int32_t K, const uint8_t* const* I1,
                                    const uint8_t* const* I2, const int32_t* dims, int32_t replace) {
    svh::ActiveCaller active_;
    if (!ms || K < 0 || !dims) return mfail(SVH_ERR_BAD_ARG, "null argument");
    if (K == 0) return SVH_OK;
    if (!I1) {
        // no images: every object takes the frame handed over by svh_matcher_prefetch_batch
        for (int i = 0; i < K; i++)
            if (!ms[i] || !ms[i]->has_next) return mfail(SVH_ERR_BAD_ARG, "no images and no prefetched frame");
        // (the objects of a lockstep hand-over share one stream: wait for each distinct stream once)
        std::vector<hipStream_t> waited;
        for (int i = 0; i < K; i++) {
            svh_matcher* m = ms[i];
            if (m->next_job.valid()) {
                const int32_t rc = m->next_job.get();
                m->next_job = std::shared_future<int32_t>();
                if (rc) {
                    m->has_next = false;
                    return mfail(rc, "the hand-over of the prefetched frame failed on the prefetch thread: " + (m->next_why ? *m->next_why : std::string()));
                }
            }
            if (std::find(waited.begin(), waited.end(), m->next_stream) == waited.end()) {
                HIP_TRY(hipSetDevice(m->device));
                HIP_TRY((hipError_t)wait_stream(m->next_stream));
                waited.push_back(m->next_stream);
            }
        }
        HIP_TRY(hipGetLastError());
        for (int i = 0; i < K; i++) {
            const int32_t rc = push_take_prefetched(ms[i], replace, true);
            if (rc) return rc;
        }
        return SVH_OK;
    }
    bool lockstep = K > 1;
    for (int i = 0; i < K && lockstep; i++) {
        if (!ms[i]) return mfail(SVH_ERR_BAD_ARG, "null matcher in the batch");
        for (int j = 0; j < i; j++)
            if (ms[j] == ms[i]) return mfail(SVH_ERR_BAD_ARG, "the same matcher twice in one batch");
        lockstep = memcmp(&ms[i]->p, &ms[0]->p, sizeof(ms[0]->p)) == 0 && !ms[i]->taps &&
                   ms[i]->device == ms[0]->device && (!I2 || !I2[i]) == (!I2 || !I2[0]) && I1[i];
    }
    auto serial = [&]() -> int32_t {
        for (int i = 0; i < K; i++) {
            const int32_t rc = svh_matcher_push_back(ms[i], I1[i], I2 ? I2[i] : nullptr, dims, replace);
            if (rc) return rc;
        }
        return SVH_OK;
    };
    if (!lockstep) return serial();
    double tb[6] = {0, 0, 0, 0, 0, 0};
    auto btick = [&](int i) { if (g_mtiming) tb[i] = mnow_ms(); };
    btick(0);
    for (int i = 0; i < K; i++) {
        const int32_t rc = push_prepare(ms[i], I1[i], I2 ? I2[i] : nullptr, dims, replace);
        if (rc) return rc;
    }
    btick(1);
    const int ncam = (I2 && I2[0]) ? 2 : 1;
    std::vector<int> rcs((size_t)K * ncam, 0);
    BatchRec& up = batch_recorder(ms[0]->device);
    HIP_TRY(up.ensure_side());
    struct InBatch {     // (cleared on every exit, the error returns included)
        InBatch() { t_in_batch = true; }
        ~InBatch() { t_in_batch = false; }
    };
    InBatch in_batch_;
    // The uploads are recorded with the features (one k_upload_b launch per camera over the K objects; they start when
    // every image is packed instead of overlapping the packing, which costs less than 2 K launches from the packing
    // threads did: 2 x 16 objects 7.8-8.6 -> 8.7-9.0 k frames/s, 1 x 32 6.1-6.2 -> 6.3-6.6 k, 1 x 16 unchanged).
    // SVH_UPLOAD_BATCH=0: every packing thread launches its image's upload, rotating over the side streams.
    static const bool upload_recorded = !(svh::env("SVH_UPLOAD_BATCH") && atoi(svh::env("SVH_UPLOAD_BATCH")) == 0);
    BatchPool::get().parallel_for(K * ncam, [&](int j) {
        (void)hipSetDevice(ms[0]->device);
        svh_matcher* m = ms[j / ncam];
        const int cam = j % ncam;
        DevView& V = m->cur[cam];
        rcs[j] = features_pack(m, V, cam, cam ? I2[j / ncam] : I1[j / ncam], dims[2]);
        // the image goes to the device while the other threads still pack theirs; the uploads rotate over the
        // call's stream and the side streams (several copy kernels in flight fill the PCIe link better)
        if (!rcs[j] && !upload_recorded) {
            const int q = j % (BatchRec::kSide + 1);
            mlaunch_upload(q == 0 ? ms[0]->stream : up.side[q - 1], V.stage, V.I, (size_t)V.bpl * V.h);
        }
    });
    HIP_TRY(up.join_side(ms[0]->stream));
    for (int rc : rcs)
        if (rc) return rc;
    btick(2);
    BatchRec& rec = batch_recorder(ms[0]->device);
    rec.reset();
    t_rec = &rec;
    int rc = SVH_OK;
    for (int i = 0; i < K && !rc; i++) {
        rec.begin_object();
        for (int cam = 0; cam < ncam && !rc; cam++)
            rc = features_enqueue(ms[i], ms[i]->cur[cam], cam, nullptr, !upload_recorded);
    }
    t_rec = nullptr;
    if (rc) return rc;
    btick(3);
    hipStream_t s = ms[0]->stream;
    if (rec.broken) {
        // (not reachable with equal parameters and sizes; kept so that a future launcher change cannot corrupt a batch)
        rec.reset();
        HIP_TRY((hipError_t)wait_stream(s));   // (the uploads issued above)
        for (int i = 0; i < K; i++) {
            for (int cam = 0; cam < ncam; cam++) {
                rc = features_enqueue(ms[i], ms[i]->cur[cam], cam, nullptr);
                if (rc) return rc;
            }
            HIP_TRY((hipError_t)wait_stream(ms[i]->stream));
            HIP_TRY((hipError_t)wait_stream(ms[i]->stream2));
            push_finish(ms[i], I1[i], I2 ? I2[i] : nullptr);
        }
        return SVH_OK;
    }
    HIP_TRY(rec.flush(s));
    btick(4);
    HIP_TRY((hipError_t)wait_stream(s));
    HIP_TRY(hipGetLastError());
    rec.synced();
    for (int i = 0; i < K; i++) push_finish(ms[i], I1[i], I2 ? I2[i] : nullptr);
    btick(5);
    if (g_mtiming) {
        for (int i = 0; i < 5; i++) g_btime.t[i] += tb[i + 1] - tb[i];
        g_btime.calls[0]++;
    }
    return SVH_OK;
}

// matchFeatures' sanity checks (matcher.cpp:216-259): false = return silently, previous matches stay
static bool match_inputs_present(const svh_matcher* m, int32_t method) {
    const svh_matcher_params& p = m->p;
    auto missing = [&](const DevView& V, int dense) { return !V.valid || V.n[dense] == 0; };
    const bool need_1p = method == 0 || method >= 2, need_2p = method >= 2;
    const bool need_2c = method >= 1;
    for (int dense = 1; dense >= (p.multi_stage ? 0 : 1); dense--) {
        if (need_1p && missing(m->prev[0], dense)) return false;
        if (need_2p && missing(m->prev[1], dense)) return false;
        if (missing(m->cur[0], dense)) return false;
        if (need_2c && missing(m->cur[1], dense)) return false;
    }
    return true;
}

// A matchFeatures that fails (a HIP error half-way) leaves the object's match list as the last good call made it
// (getMatches / getGain / bucketFeatures of the caller keep working on a consistent list): the list is set aside for
// the duration of the call and put back unless the call commits.  Two vectors swapped: no copy, capacities kept.
struct KeepMatches {
    std::vector<svh_matcher*> ms;
    bool committed = false;
    explicit KeepMatches(const std::vector<svh_matcher*>& list) : ms(list) {
        for (svh_matcher* m : ms) m->m2.swap(m->m2_kept);
    }
    void commit() { committed = true; }
    ~KeepMatches() {
        if (!committed)
            for (svh_matcher* m : ms) m->m2.swap(m->m2_kept);
    }
};

// result vectors cleared, bin indices of changed tables rebuilt, prior-range buffer sized
static int32_t match_prepare(svh_matcher* m, int32_t ub, int32_t vb) {
    for (int s = 0; s < SVH_M_STAGE_COUNT; s++) m->stage[s].clear();
    m->m1.clear();
    m->m2.clear();
    DevView* views[4] = {&m->prev[0], &m->prev[1], &m->cur[0], &m->cur[1]};
    const int rc = ensure_bins(m, views, 4, ub, vb);
    if (rc) return rc;
    if (ub * vb > m->ranges_cap) {
        m->ranges_cap = 0;
        HIP_TRY(drealloc(&m->ranges_dev, (size_t)16 * ub * vb));
        m->ranges_cap = ub * vb;
    }
    return SVH_OK;
}

int32_t svh_matcher_match_features(svh_matcher* m, int32_t method, const double* Tr) {
    svh::ActiveCaller active_;
    if (!m) return mfail(SVH_ERR_BAD_ARG, "null argument");
    const svh_matcher_params& p = m->p;
    if (!match_inputs_present(m, method)) return SVH_OK;
    if (method > 2) method = 2;
    KeepMatches keep_({m});
    HIP_TRY(hipSetDevice(m->device));
    const int32_t ub = (int32_t)ceilf((float)m->dims_c[0] / (float)p.match_binsize);
    const int32_t vb = (int32_t)ceilf((float)m->dims_c[1] / (float)p.match_binsize);
    int rc = match_prepare(m, ub, vb);
    if (rc) return rc;
    double tm[6] = {0, 0, 0, 0, 0, 0};
    auto tick = [&](int i) { if (g_mtiming) tm[i] = mnow_ms(); };
    tick(0);
    // the dense vote of a frame like the last one will triangulate on helper threads: have them awake by then
    const bool warm = m->last_dense >= kVoteParMin && vote_par_depth() > 0;
    if (p.multi_stage) {
        rc = run_matching(m, 0, method, false, Tr, m->m1, false, &m->stage[SVH_M_SPARSE_RAW]);
        if (rc) return rc;
        tick(1);
        rc = remove_outliers(p, m->m1, method);
        if (rc) return rc;
        tick(2);
        if (m->taps) m->stage[SVH_M_SPARSE] = m->m1;
        prior_statistics(m, m->m1, method, ub, vb);
        HIP_TRY(hipMemcpyAsync(m->ranges_dev, m->ranges.data(), m->ranges.size() * sizeof(float),
                               hipMemcpyHostToDevice, m->stream));
        tick(3);
        m->warm_on_wait = warm;
        rc = run_matching(m, 1, method, true, Tr, m->m2, p.refinement > 0, &m->stage[SVH_M_DENSE_RAW]);
    } else {
        tick(1); tick(2); tick(3);
        m->warm_on_wait = warm;
        rc = run_matching(m, 1, method, false, Tr, m->m2, p.refinement > 0, &m->stage[SVH_M_DENSE_RAW]);
    }
    m->warm_on_wait = false;
    if (rc) {
        if (warm) helpers_warm(0, 0);
        return rc;
    }
    tick(4);
    if (m->taps) m->stage[SVH_M_DENSE_REFINED] = m->m2;
    m->last_dense = (int32_t)m->m2.size();
    rc = remove_outliers(p, m->m2, method);
    if (warm) helpers_warm(0, 0);
    if (rc) return rc;
    tick(5);
    if (g_mtiming) {
        for (int i = 0; i < 5; i++) m->tacc[T_SPARSE + i] += tm[i + 1] - tm[i];
        m->tcalls[1]++;
    }
    if (m->taps) m->stage[SVH_M_DENSE] = m->m2;
    keep_.commit();
    return SVH_OK;
}

// matchFeatures for K Matchers in lockstep: every device step is one launch over the objects that pass the
// sanity checks, the host steps between them (outlier votes, prior statistics) run on the helper threads.
// Tr = K pointers (or null), one Tr_delta per object.  Results are those of K svh_matcher_match_features calls.
int32_t svh_matcher_match_features_batch(svh_matcher* const* ms, int32_t K, int32_t method,
                                         const double* const* Tr) {
    svh::ActiveCaller active_;
    if (!ms || K < 0) return mfail(SVH_ERR_BAD_ARG, "null argument");
    if (K == 0) return SVH_OK;
    bool lockstep = K > 1;
    for (int i = 0; i < K && lockstep; i++) {
        if (!ms[i]) return mfail(SVH_ERR_BAD_ARG, "null matcher in the batch");
        for (int j = 0; j < i; j++)
            if (ms[j] == ms[i]) return mfail(SVH_ERR_BAD_ARG, "the same matcher twice in one batch");
        lockstep = memcmp(&ms[i]->p, &ms[0]->p, sizeof(ms[0]->p)) == 0 && !ms[i]->taps &&
                   ms[i]->device == ms[0]->device && memcmp(ms[i]->dims_c, ms[0]->dims_c, sizeof(ms[0]->dims_c)) == 0;
    }
    auto serial = [&](const std::vector<svh_matcher*>& list, const std::vector<const double*>& trs) -> int32_t {
        for (size_t i = 0; i < list.size(); i++) {
            const int32_t rc = svh_matcher_match_features(list[i], method, trs[i]);
            if (rc) return rc;
        }
        return SVH_OK;
    };
    std::vector<svh_matcher*> part;
    std::vector<const double*> trs;
    for (int i = 0; i < K; i++) {
        if (!ms[i]) return mfail(SVH_ERR_BAD_ARG, "null matcher in the batch");
        if (lockstep && !match_inputs_present(ms[i], method)) continue;
        part.push_back(ms[i]);
        trs.push_back(Tr ? Tr[i] : nullptr);
    }
    if (!lockstep || part.size() < 2) return serial(part, trs);
    if (method > 2) method = 2;
    KeepMatches keep_(part);
    const int n = (int)part.size();
    const svh_matcher_params& p = part[0]->p;
    HIP_TRY(hipSetDevice(part[0]->device));
    const int32_t ub = (int32_t)ceilf((float)part[0]->dims_c[0] / (float)p.match_binsize);
    const int32_t vb = (int32_t)ceilf((float)part[0]->dims_c[1] / (float)p.match_binsize);
    hipStream_t s = part[0]->stream;
    BatchRec& rec = batch_recorder(part[0]->device);
    std::vector<MatchPass> mp(n);
    std::vector<int> rcs(n, 0);
    double t_wait = 0, tm[5] = {0, 0, 0, 0, 0};
    auto mtick = [&](int i) { if (g_mtiming) tm[i] = mnow_ms(); };
    mtick(0); mtick(1); mtick(2);
    // record one device phase over all objects; on a sequence mismatch the objects run one by one instead
    // (an error exit from a recording pass: ensure_bins has marked the bin indices of the recorded objects as built
    // although no index kernel ran -- they are marked unbuilt again, the next matchFeatures rebuilds them)
    auto unbuild = [&]() {
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 2; k++) part[i]->prev[k].nbins = part[i]->cur[k].nbins = 0;
    };
    auto device_phase_inner = [&](const std::function<int(int)>& body) -> int {
        rec.reset();
        t_rec = &rec;
        int rc = SVH_OK;
        for (int i = 0; i < n && !rc; i++) {
            rec.begin_object();
            rc = body(i);
        }
        t_rec = nullptr;
        if (rc) return rc;
        if (rec.broken) {
            rec.reset();
            for (int i = 0; i < n; i++) {
                // (the recording pass marked the bin indices as built: build them for real)
                for (int k = 0; k < 2; k++) part[i]->prev[k].nbins = part[i]->cur[k].nbins = 0;
                rc = body(i);
                if (rc) return rc;
                HIP_TRY((hipError_t)wait_stream(part[i]->stream));
            }
            HIP_TRY(hipGetLastError());
            return SVH_OK;
        }
        HIP_TRY(rec.flush(s));
        const double tw = g_mtiming ? mnow_ms() : 0;
        HIP_TRY((hipError_t)wait_stream(s));
        HIP_TRY(hipGetLastError());
        rec.synced();
        if (g_mtiming) t_wait += mnow_ms() - tw;
        return SVH_OK;
    };
    auto device_phase = [&](const std::function<int(int)>& body) -> int {
        const int rc = device_phase_inner(body);
        if (rc) {
            t_rec = nullptr;
            rec.reset();
            unbuild();
        }
        return rc;
    };
    auto host_phase = [&](const std::function<int(int)>& body) -> int {
        t_in_batch = true;
        BatchPool::get().parallel_for(n, [&](int i) {
            t_in_batch = true;
            rcs[i] = body(i);
            t_in_batch = false;
        });
        t_in_batch = false;
        for (int rc : rcs)
            if (rc) return rc;
        return SVH_OK;
    };
    int rc;
    if (p.multi_stage) {
        rc = device_phase([&](int i) {
            int r = match_prepare(part[i], ub, vb);
            if (!r) r = match_enqueue(part[i], 0, method, false, trs[i], mp[i]);
            if (!r) download_enqueue(part[i], mp[i]);
            return r;
        });
        if (rc) return rc;
        mtick(1);
        if (g_mtiming) {
            g_btime.t[5] += tm[1] - tm[0] - t_wait;
            g_btime.t[6] += t_wait;
            t_wait = 0;
        }
        const size_t nr = (size_t)16 * ub * vb;
        for (int i = 0; i < n; i++) {
            svh_matcher* m = part[i];
            match_collect(m, mp[i], m->m1);
            if (nr > m->h_ranges_cap) {
                m->h_ranges_cap = 0;
                HIP_TRY(hrealloc(&m->h_ranges, nr * sizeof(float)));
                m->h_ranges_cap = nr;
            }
        }
        rc = host_phase([&](int i) {
            svh_matcher* m = part[i];
            const int r = remove_outliers(m->p, m->m1, method);
            if (r) return r;
            prior_statistics(m, m->m1, method, ub, vb);
            memcpy(m->h_ranges, m->ranges.data(), m->ranges.size() * sizeof(float));
            return (int)SVH_OK;
        });
        if (rc) return rc;
        mtick(2);
        if (g_mtiming) g_btime.t[7] += tm[2] - tm[1];
    }
    rc = device_phase([&](int i) {
        svh_matcher* m = part[i];
        int r = SVH_OK;
        if (p.multi_stage)
            mlaunch_copy(m->stream, m->ranges_dev, m->h_ranges, m->ranges.size() * sizeof(float), hipMemcpyHostToDevice);
        else
            r = match_prepare(m, ub, vb);
        if (!r) r = match_enqueue(m, 1, method, p.multi_stage != 0, trs[i], mp[i]);
        if (r) return r;
        if (p.refinement > 0) refine_enqueue(m, method, mp[i]);
        download_enqueue(m, mp[i]);
        return (int)SVH_OK;
    });
    if (rc) return rc;
    for (int i = 0; i < n; i++) match_collect(part[i], mp[i], part[i]->m2);
    mtick(3);
    rc = host_phase([&](int i) { return remove_outliers(part[i]->p, part[i]->m2, method); });
    mtick(4);
    if (g_mtiming) {
        g_btime.t[8] += tm[3] - tm[2] - t_wait;
        g_btime.t[9] += t_wait;
        g_btime.t[10] += tm[4] - tm[3];
        g_btime.calls[1]++;
    }
    if (rc == SVH_OK) keep_.commit();
    return rc;
}

// Matcher::bucketFeatures   matcher.cpp:297-343 (host; std::random_shuffle like the reference)
int32_t svh_matcher_bucket_features(svh_matcher* m, int32_t max_features, float bw, float bh) {
    RandStream libc;   // not private: libc rand()
    return bucket_features(m, max_features, bw, bh, libc);
}

int32_t svh_matcher_get_matches(svh_matcher* m, svh_p_match* out, int32_t cap) {
    if (!m) return 0;
    for (int32_t i = 0; i < (int32_t)m->m2.size() && i < cap && out; i++) out[i] = m->m2[i];
    return (int32_t)m->m2.size();
}

// Matcher::getGain + mean   matcher.cpp:347-389, 1825-1837
float svh_matcher_get_gain(svh_matcher* m, const int32_t* inliers, int32_t n) {
    if (!m || !m->prev[0].valid || !m->cur[0].valid || m->m2.empty() || n == 0) return 1;
    auto meanf = [](const DevView& V, int32_t u0, int32_t u1, int32_t v0, int32_t v1) {
        float s = 0;
        for (int32_t v = v0; v <= v1; v++)
            for (int32_t u = u0; u <= u1; u++) s += (float)V.stage[(size_t)v * V.bpl + u];
        return s /= (float)((u1 - u0 + 1) * (v1 - v0 + 1));
    };
    auto cl = [](int32_t x, int32_t hi) { return std::min(std::max(x, 0), hi); };
    float gain = 0;
    int32_t num = 0;
    for (int32_t q = 0; q < n; q++) {
        if (inliers[q] >= (int32_t)m->m2.size()) continue;
        const svh_p_match& it = m->m2[inliers[q]];
        // (the reference clamps both windows with dims_p; the views here are indexed by their
        // own geometry, so the bound of a view is never exceeded when the two frames differ)
        const int32_t W = std::min(m->dims_p[0], m->prev[0].bpl - 1), H = std::min(m->dims_p[1], m->prev[0].h);
        const int32_t Wc = std::min(m->dims_p[0], m->cur[0].bpl - 1), Hc = std::min(m->dims_p[1], m->cur[0].h);
        const float mp = meanf(m->prev[0], cl((int32_t)it.u1p - 3, W), cl((int32_t)it.u1p + 3, W),
                               cl((int32_t)it.v1p - 3, H), cl((int32_t)it.v1p + 3, H));
        const float mc = meanf(m->cur[0], cl((int32_t)it.u1c - 3, Wc), cl((int32_t)it.u1c + 3, Wc),
                               cl((int32_t)it.v1c - 3, Hc), cl((int32_t)it.v1c + 3, Hc));
        if (mp > 10) {
            gain += mc / mp;
            num++;
        }
    }
    return num > 0 ? gain / (float)num : 1;
}

int32_t svh_matcher_set_taps(svh_matcher* m, int32_t enable) {
    if (!m) return SVH_ERR_BAD_ARG;
    m->taps = enable != 0;
    return SVH_OK;
}

int32_t svh_matcher_get_features(svh_matcher* m, int32_t table, int32_t* out, int32_t cap) {
    if (!m || table < 0 || table > 7) return SVH_ERR_BAD_ARG;
    DevView& V = (table < 4 ? m->prev : m->cur)[(table >> 1) & 1];
    const int dense = table & 1;
    if (!V.valid) return 0;
    const int32_t n = V.n[dense];
    if (out && n > 0) {
        if (hipSetDevice(m->device) != hipSuccess) return SVH_ERR_HIP;
        if (hipMemcpy(out, V.tab[dense], sizeof(int32_t) * 12 * std::min(n, cap), hipMemcpyDeviceToHost) !=
            hipSuccess)
            return mfail(SVH_ERR_HIP, "feature table copy failed");
    }
    return n;
}

int32_t svh_matcher_get_stage(svh_matcher* m, int32_t stage, void* buf, size_t cap, size_t* size) {
    if (!m || stage < 0 || stage >= SVH_M_STAGE_COUNT) return SVH_ERR_BAD_ARG;
    const void* src;
    size_t n;
    if (stage == SVH_M_RANGES) {
        src = m->ranges.data();
        n = m->ranges.size() * sizeof(float);
    } else {
        src = m->stage[stage].data();
        n = m->stage[stage].size() * sizeof(svh_p_match);
    }
    if (size) *size = n;
    if (!buf) return SVH_OK;
    if (cap < n) return mfail(SVH_ERR_BAD_ARG, "stage buffer too small");
    if (n) memcpy(buf, src, n);
    return SVH_OK;
}

int32_t svh_matcher_get_filter(svh_matcher* m, int32_t which, void* buf, size_t cap, size_t* size,
                               int32_t* dims3) {
    if (!m || which < 0 || which > 5) return SVH_ERR_BAD_ARG;
    const DevView& V = m->cur[0];
    if (!V.valid) return SVH_ERR_BAD_ARG;
    const bool full = which == 2 || which == 3;
    if (full && !V.du_full) return SVH_ERR_BAD_ARG;
    const int32_t d[3] = {full ? V.w : V.mw, full ? V.h : V.mh, full ? V.bpl : V.mbpl};
    const void* src[6] = {V.du, V.dv, V.du_full, V.dv_full, V.f1, V.f2};
    const size_t n = (size_t)d[2] * d[1] * (which >= 4 ? 2 : 1);
    if (size) *size = n;
    if (dims3) memcpy(dims3, d, sizeof(d));
    if (!buf) return SVH_OK;
    if (cap < n) return mfail(SVH_ERR_BAD_ARG, "filter buffer too small");
    if (hipSetDevice(m->device) != hipSuccess ||
        hipMemcpy(buf, src[which], n, hipMemcpyDeviceToHost) != hipSuccess)
        return mfail(SVH_ERR_HIP, "filter image copy failed");
    return SVH_OK;
}

}  // extern "C"
