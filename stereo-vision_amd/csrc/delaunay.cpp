// Host-side Delaunay triangulation for the ELAS disparity prior and the
// libviso2 outlier filter.
//
// The reference calls Shewchuk's Triangle in library mode with switches "zQB"
// (libelas/src/elas.cpp:581-582, libviso2/src/matcher.cpp:1432-1433).  Support
// points sit on a 5-px lattice, so several percent of the interior edges are
// exactly co-circular and the result depends on Triangle's tie-breaking; the
// triangle ORDER decides which plane the few multiply-covered pixels get, and
// libviso2 match indices depend on which duplicate vertex survives.  This file
// is an independent implementation of the same published algorithm --
// Guibas & Stolfi divide-and-conquer on a triangle-based structure with Dwyer's
// alternating cuts -- arranged so that it reproduces Triangle's observable
// output exactly:
//
//   * vertex order: randomised quicksort / quickselect driven by the LCG
//     seed = (seed*1366 + 150889) % 714025, seed 1 per call
//     (triangle.cpp:4045-4049, 5446-5568), duplicates dropped keeping the first
//     in sorted order (triangle.cpp:6179-6196), alternating-axis reorder
//     (triangle.cpp:5582-5604);
//   * recursion: split at n/2, left first, base cases of 2 and 3 vertices that
//     create 2 resp. 4 triangle records (triangle.cpp:5953-6103);
//   * merge: strict ">0" orientation / in-circle tests, "<=0" finish tests, the
//     left candidate kept on co-circular ties, edge removal by in-place flips,
//     exactly two new records per merge (triangle.cpp:5638-5934);
//   * output: surviving records in creation order, corners (org,dest,apex) of
//     orientation 0 (triangle.cpp:7800-7860).
//
// Differences by design: records live in flat index arrays (no pointers, no
// pools), predicates are exact integer determinants (__int128) instead of
// adaptive floating-point expansions -- same sign, coordinates are integers or
// dyadic fractions -- and all state is per call, so the routine is re-entrant
// (the reference keeps its LCG seed and predicate constants in globals,
// triangle.cpp:541-550).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svh.h"

#ifndef TICK
#define TICK(i)
#endif

namespace svh {

// ---------------------------------------------------------------------------
// Helper threads for the latency paths (parallel divide-and-conquer, the two triangulations of
// one stereo pair, the Matcher's outlier vote).  The tasks are 20-100 us, so what a hand-over costs
// decides whether helping pays:
//   * every helper has its own mailbox (one cache line): a task goes to ONE chosen helper, nobody
//     else sees it, no lock is fought over;
//   * a helper polls its mailbox while helpers_warm() says a parallel section is near (and for a
//     moment after a task: the nested split of its half follows at once) and sleeps on its own
//     condition variable otherwise -- a sleeping helper costs a futex wake, 30-50 us late;
//   * a helper moves to the cores that share the submitter's L3 (sysfs): the halves of a
//     triangulation exchange their records through that cache instead of across the fabric
//     (EPYC 9575F, 2.9 k points, 8 threads: 190-205 us with parked helpers anywhere, 130 us so).
// Started lazily, never more than kHelpers.
// ---------------------------------------------------------------------------
namespace {
inline int64_t mono_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct L3Map {
    std::vector<int> group;        // cpu -> L3 domain (-1: unknown)
    std::vector<cpu_set_t> mask;   // L3 domain -> its cpus this process may use
    L3Map() {
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
        std::vector<std::string> seen;
        int missing = 0;
        for (int c = 0; c < CPU_SETSIZE && missing < 64; c++) {
            char path[128];
            snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
            FILE* f = fopen(path, "r");
            if (!f) { missing++; continue; }
            missing = 0;
            char buf[512];
            if (!fgets(buf, sizeof(buf), f)) buf[0] = 0;
            fclose(f);
            const std::string key(buf);
            int g = -1;
            for (size_t i = 0; i < seen.size(); i++)
                if (seen[i] == key) g = (int)i;
            if (g < 0) {
                g = (int)seen.size();
                seen.push_back(key);
                cpu_set_t m;
                CPU_ZERO(&m);
                const char* q = buf;   // "0-7,128-135"
                while (*q >= '0' && *q <= '9') {
                    char* e;
                    long lo = strtol(q, &e, 10), hi = lo;
                    if (*e == '-') hi = strtol(e + 1, &e, 10);
                    for (long k = lo; k <= hi && k < CPU_SETSIZE; k++)
                        if (CPU_ISSET(k, &allowed)) CPU_SET(k, &m);
                    q = *e == ',' ? e + 1 : e;
                }
                mask.push_back(m);
            }
            if ((int)group.size() <= c) group.resize(c + 1, -1);
            group[c] = g;
        }
    }
    int of(int cpu) const { return cpu >= 0 && cpu < (int)group.size() ? group[cpu] : -1; }
};
const L3Map& l3map() {
    static const L3Map* m = new L3Map();
    return *m;
}

struct HelperPool {
    static constexpr int kHelpers = 7;
    static constexpr int64_t kAfterTaskNs = 20000;
    enum : int { ABSENT = 0, PARKED, HOT, CLAIMED, TASK };
    struct alignas(128) Box {
        std::atomic<int> state{ABSENT};
        std::function<void()> fn;
        std::atomic<int>* done = nullptr;
        int l3 = -1;                 // the submitter's L3 domain
        std::mutex mu;
        std::condition_variable cv;
    };
    Box box[kHelpers];
    // A helper that polls is a busy core: no more helpers than the process has CPUs to spare (its affinity mask, and
    // the CPU quota of its cgroup if it has one) beyond the calling thread's -- none at all below three CPUs, where a
    // polling helper would run INSTEAD of the caller.  Callers see a pool that is always busy and do the work themselves.
    const int usable = helper_budget();
    static int helper_budget() {
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        int cpus = sched_getaffinity(0, sizeof(allowed), &allowed) == 0 ? CPU_COUNT(&allowed) : 1;
        for (const char* path : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"}) {
            FILE* f = fopen(path, "r");
            if (!f) continue;
            char buf[64] = {0};
            long quota = -1, period = 100000;
            if (fgets(buf, sizeof(buf), f)) {
                if (sscanf(buf, "%ld %ld", &quota, &period) < 1) quota = -1;   // ("max 100000": no quota)
            }
            fclose(f);
            if (quota > 0 && period > 0) cpus = std::min<long>(cpus, (quota + period - 1) / period);
            break;
        }
        return cpus < 3 ? 0 : std::min(kHelpers, cpus - 1);
    }
    std::atomic<int64_t> n_tasks{0}, n_moves{0}, n_hot{0}, n_woken{0};   // tasks run, of them after a move to another L3; taken by a polling / a sleeping helper
    std::atomic<int64_t> warm_until{0};   // helpers poll instead of sleeping until then
    std::mutex spawn_mu;

    void worker(Box* b) {
        int64_t stay = 0;
        int mine = -2;
        for (;;) {
            int st = b->state.load(std::memory_order_acquire);
            if (st == TASK) {
                const int g = b->l3;
                if (g >= 0 && g != mine && CPU_COUNT(&l3map().mask[g]) >= 2 &&
                    sched_setaffinity(0, sizeof(cpu_set_t), &l3map().mask[g]) == 0) {
                    mine = g;
                    n_moves.fetch_add(1, std::memory_order_relaxed);
                }
                n_tasks.fetch_add(1, std::memory_order_relaxed);
                b->fn();
                b->fn = nullptr;
                std::atomic<int>* d = b->done;
                b->state.store(HOT, std::memory_order_release);   // free again BEFORE the submitter is released
                d->store(1, std::memory_order_release);
                stay = mono_ns() + kAfterTaskNs;
            } else if (st == HOT) {
                bool changed = false;
                for (int k = 0; k < 32 && !changed; k++) {
                    __builtin_ia32_pause();
                    changed = b->state.load(std::memory_order_acquire) != HOT;
                }
                if (changed || mono_ns() < std::max(stay, warm_until.load(std::memory_order_relaxed))) continue;
                int want = HOT;
                if (!b->state.compare_exchange_strong(want, PARKED, std::memory_order_acq_rel)) continue;   // claimed meanwhile
                std::unique_lock<std::mutex> lk(b->mu);
                b->cv.wait(lk, [&] { return b->state.load(std::memory_order_acquire) != PARKED; });
            } else if (st == PARKED) {
                std::unique_lock<std::mutex> lk(b->mu);
                b->cv.wait(lk, [&] { return b->state.load(std::memory_order_acquire) != PARKED; });
            } else {   // CLAIMED: the submitter is writing the task
                __builtin_ia32_pause();
            }
        }
    }
    bool hand(Box* b, int from, const std::function<void()>& fn, std::atomic<int>* done) {
        int want = from;
        if (!b->state.compare_exchange_strong(want, CLAIMED, std::memory_order_acq_rel)) return false;
        b->fn = fn;
        b->done = done;
        b->l3 = l3map().of(sched_getcpu());
        if (from == PARKED) {
            {
                std::lock_guard<std::mutex> lk(b->mu);
                b->state.store(TASK, std::memory_order_release);
            }
            b->cv.notify_one();
            n_woken.fetch_add(1, std::memory_order_relaxed);
        } else {
            b->state.store(TASK, std::memory_order_release);
            n_hot.fetch_add(1, std::memory_order_relaxed);
        }
        return true;
    }
    // false: no helper free right now -- the caller runs the task itself
    bool submit(const std::function<void()>& fn, std::atomic<int>* done) {
        for (int i = 0; i < usable; i++)
            if (box[i].state.load(std::memory_order_relaxed) == HOT && hand(&box[i], HOT, fn, done)) return true;
        for (int i = 0; i < usable; i++)
            if (box[i].state.load(std::memory_order_relaxed) == PARKED && hand(&box[i], PARKED, fn, done)) return true;
        if (usable == 0) return false;
        std::lock_guard<std::mutex> lk(spawn_mu);
        for (int i = 0; i < usable; i++) {
            Box& b = box[i];
            if (b.state.load(std::memory_order_relaxed) == ABSENT) {
                b.fn = fn;
                b.done = done;
                b.l3 = l3map().of(sched_getcpu());
                b.state.store(TASK, std::memory_order_release);
                std::thread(&HelperPool::worker, this, &b).detach();
                return true;
            }
        }
        return false;
    }
    void warm(int want, int64_t ns) {
        if (ns <= 0) {
            warm_until.store(0, std::memory_order_relaxed);
            return;
        }
        warm_until.store(mono_ns() + ns, std::memory_order_relaxed);
        int n = 0;
        for (Box& b : box) {
            if (n++ >= std::min(want, usable)) break;
            const int st = b.state.load(std::memory_order_relaxed);
            if (st == ABSENT) {
                std::lock_guard<std::mutex> lk(spawn_mu);
                if (b.state.load(std::memory_order_relaxed) == ABSENT) {
                    b.state.store(HOT, std::memory_order_release);
                    std::thread(&HelperPool::worker, this, &b).detach();
                }
            } else if (st == PARKED) {
                int p = PARKED;
                bool woke;
                {
                    std::lock_guard<std::mutex> lk(b.mu);
                    woke = b.state.compare_exchange_strong(p, HOT, std::memory_order_acq_rel);
                }
                if (woke) b.cv.notify_one();
            }
        }
    }
};
HelperPool& helper_pool() {
    static HelperPool* p = new HelperPool();   // leaked on purpose: helpers outlive static destruction
    return *p;
}
}  // namespace

void helpers_warm(int want, int us) { helper_pool().warm(want, (int64_t)us * 1000); }

void run_pair(const std::function<void()>& a, const std::function<void()>& b) {
    std::atomic<int> done{0};
    if (helper_pool().submit(b, &done)) {
        a();
        while (!done.load(std::memory_order_acquire)) __builtin_ia32_pause();
    } else {
        a();
        b();
    }
}

void run_many(int k, const std::function<void(int)>& fn) {
    std::atomic<int> done[HelperPool::kHelpers];
    bool sent[HelperPool::kHelpers];
    if (k > HelperPool::kHelpers + 1) k = HelperPool::kHelpers + 1;   // (callers split into at most 8)
    for (int i = 1; i < k; i++) {
        done[i - 1].store(0, std::memory_order_relaxed);
        sent[i - 1] = helper_pool().submit([&fn, i]() { fn(i); }, &done[i - 1]);
    }
    fn(0);
    for (int i = 1; i < k; i++) {
        if (!sent[i - 1]) fn(i);
        else while (!done[i - 1].load(std::memory_order_acquire)) __builtin_ia32_pause();
    }
}

namespace {

typedef __int128 i128;

struct Handle {
    int32_t t;  // record index (0 = "outer space")
    int32_t o;  // orientation 0..2
};

class DivConq {
public:
    DivConq() {}
    // An object is reused from call to call (dc_take / dc_give below): its arrays keep their capacity, so a
    // triangulation allocates nothing and touches no fresh page once the first one of its size has run.
    void reset(const float* pts, int32_t n, int par_depth, bool expect_dups) {
        expect_dups_ = expect_dups;
        pts_ = pts;
        n_ = n;
        seed_ = 1;
        par_depth_ = par_depth;
        nrec_ = 0;
        narrow_ = false;
    }
    size_t bytes_held() const {
        return (ix_.capacity() + iy_.capacity()) * 8 + rec_.capacity() * 4 + (order_.capacity() + ly_.capacity() + kd_.capacity() + rank_.capacity()) * 4 +
               (keys_.capacity() + ktmp_.capacity() + ykeys_.capacity() + ytmp_.capacity() + lx_.capacity() + lyk_.capacity() + scratch_.capacity()) * 8;
    }

    // returns triangle count, or <0 on failure
    int32_t run(int32_t* out, int32_t cap);

private:
    const float* pts_ = nullptr;
    int32_t n_ = 0;
    uint64_t seed_ = 1;
    int par_depth_ = 0;   // levels of the divide-and-conquer whose halves run on two threads
    bool expect_dups_ = false;   // the caller's points usually contain coincident ones (Matcher): no radix sort first
    std::vector<int64_t> ix_, iy_;   // exact scaled integer coordinates
    // work arrays of run()
    std::vector<int32_t> order_, ly_, kd_, rank_;
    std::vector<uint64_t> keys_, ktmp_, ykeys_, ytmp_, lx_, lyk_, scratch_;
    // one 32-byte record per triangle: vertices [0..2] (-1 = the ghost apex), neighbour
    // handles [3..5] encoded t*4+o, dead flag [6]
    std::vector<int32_t> rec_;
    int32_t nrec_ = 0;

    float X(int32_t v) const { return pts_[2 * v]; }
    float Y(int32_t v) const { return pts_[2 * v + 1]; }
    float C(int32_t v, int axis) const { return pts_[2 * v + axis]; }

    // ---- handle algebra -------------------------------------------------
    static int p1(int o) { return (0x09 >> (2 * o)) & 3; }   // (o + 1) % 3: 0->1, 1->2, 2->0
    static int p2(int o) { return (0x12 >> (2 * o)) & 3; }   // (o + 2) % 3: 0->2, 1->0, 2->1
    static Handle next(Handle h) { h.o = p1(h.o); return h; }
    static Handle prev(Handle h) { h.o = p2(h.o); return h; }
    Handle sym(Handle h) const {
        int32_t e = rec_[8 * h.t + 3 + h.o];
        Handle r = {e >> 2, e & 3};
        return r;
    }
    int32_t org(Handle h) const { return rec_[8 * h.t + p1(h.o)]; }
    int32_t dest(Handle h) const { return rec_[8 * h.t + p2(h.o)]; }
    int32_t apex(Handle h) const { return rec_[8 * h.t + h.o]; }
    void set_org(Handle h, int32_t v) { rec_[8 * h.t + p1(h.o)] = v; }
    void set_dest(Handle h, int32_t v) { rec_[8 * h.t + p2(h.o)] = v; }
    void set_apex(Handle h, int32_t v) { rec_[8 * h.t + h.o] = v; }
    void bond(Handle a, Handle b) {
        rec_[8 * a.t + 3 + a.o] = b.t * 4 + b.o;
        rec_[8 * b.t + 3 + b.o] = a.t * 4 + a.o;
    }
    // `ctr` is the caller's record counter: a subproblem of n vertices creates exactly 2n-2
    // records (2 for an edge, 4 for a triangle, 2 per merge), so the two halves of a split own
    // known, disjoint index ranges and can be triangulated by different threads while the
    // records still come out in the sequential creation order (= Triangle's output order).
    // rec_ is sized for every record up front (run()), so it never moves.
    Handle make(int32_t& ctr) {
        int32_t* r = &rec_[8 * (size_t)ctr];
        r[0] = r[1] = r[2] = -1;
        r[3] = r[4] = r[5] = 0;   // outer space, orientation 0
        r[6] = r[7] = 0;
        Handle h = {ctr++, 0};
        return h;
    }

    // ---- exact predicates -------------------------------------------------
    // 64-bit products when every scaled coordinate is below 2^14 (the in-circle
    // determinant then stays under 2^61), 128-bit otherwise
    bool narrow_ = false;
    int ccw(int32_t a, int32_t b, int32_t c) const {
        if (narrow_) {
            const int64_t l = (ix_[a] - ix_[c]) * (iy_[b] - iy_[c]);
            const int64_t r = (iy_[a] - iy_[c]) * (ix_[b] - ix_[c]);
            return l > r ? 1 : (l < r ? -1 : 0);
        }
        i128 l = (i128)(ix_[a] - ix_[c]) * (iy_[b] - iy_[c]);
        i128 r = (i128)(iy_[a] - iy_[c]) * (ix_[b] - ix_[c]);
        return l > r ? 1 : (l < r ? -1 : 0);
    }
    int incircle(int32_t a, int32_t b, int32_t c, int32_t d) const {
        if (narrow_) {
            const int64_t adx = ix_[a] - ix_[d], ady = iy_[a] - iy_[d];
            const int64_t bdx = ix_[b] - ix_[d], bdy = iy_[b] - iy_[d];
            const int64_t cdx = ix_[c] - ix_[d], cdy = iy_[c] - iy_[d];
            const int64_t al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
            const int64_t det = al * (bdx * cdy - cdx * bdy) + bl * (cdx * ady - adx * cdy) +
                                cl * (adx * bdy - bdx * ady);
            return det > 0 ? 1 : (det < 0 ? -1 : 0);
        }
        i128 adx = ix_[a] - ix_[d], ady = iy_[a] - iy_[d];
        i128 bdx = ix_[b] - ix_[d], bdy = iy_[b] - iy_[d];
        i128 cdx = ix_[c] - ix_[d], cdy = iy_[c] - iy_[d];
        i128 al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
        i128 det = al * (bdx * cdy - cdx * bdy) + bl * (cdx * ady - adx * cdy) +
                   cl * (adx * bdy - bdx * ady);
        return det > 0 ? 1 : (det < 0 ? -1 : 0);
    }

    // ---- ordering ---------------------------------------------------------
    uint32_t pick(uint32_t choices) {
        seed_ = (seed_ * 1366u + 150889u) % 714025u;
        return (uint32_t)(seed_ / (714025u / choices + 1));
    }
    bool less2(int32_t a, float p1, float p2, int axis) const {
        float c1 = C(a, axis);
        return c1 < p1 || (c1 == p1 && C(a, 1 - axis) < p2);
    }
    bool greater2(int32_t a, float p1, float p2, int axis) const {
        float c1 = C(a, axis);
        return c1 > p1 || (c1 == p1 && C(a, 1 - axis) > p2);
    }
    // Hoare partition around a pseudo-random pivot; returns the two scan ends
    void partition(int32_t* a, int32_t n, int axis, int32_t* lo, int32_t* hi) {
        int32_t pv = a[pick((uint32_t)n)];
        float p1 = C(pv, axis), p2 = C(pv, 1 - axis);
        int32_t l = -1, r = n;
        while (l < r) {
            do { l++; } while (l <= r && less2(a[l], p1, p2, axis));
            do { r--; } while (l <= r && greater2(a[r], p1, p2, axis));
            if (l < r) { int32_t t = a[l]; a[l] = a[r]; a[r] = t; }
        }
        *lo = l;
        *hi = r;
    }
    void order_pair(int32_t* a, int axis) {
        if (greater2(a[0], C(a[1], axis), C(a[1], 1 - axis), axis)) {
            int32_t t = a[0]; a[0] = a[1]; a[1] = t;
        }
    }
    void sort_xy(int32_t* a, int32_t n) {
        if (n == 2) { order_pair(a, 0); return; }
        int32_t l, r;
        partition(a, n, 0, &l, &r);
        if (l > 1) sort_xy(a, l);
        if (r < n - 2) sort_xy(a + r + 1, n - r - 1);
    }
    void select(int32_t* a, int32_t n, int32_t median, int axis) {
        if (n == 2) { order_pair(a, axis); return; }
        int32_t l, r;
        partition(a, n, axis, &l, &r);
        if (l > median) select(a, l, median, axis);
        if (r < median - 1) select(a + r + 1, n - r - 1, median - r - 1, axis);
    }
    void alternate(int32_t* a, int32_t n, int axis) {
        int32_t half = n >> 1;
        if (n <= 3) axis = 0;
        select(a, n, half, axis);
        if (n - half >= 2) {
            if (half >= 2) alternate(a, half, 1 - axis);
            alternate(a + half, n - half, 1 - axis);
        }
    }

    // ---- packed keys ---------------------------------------------------------
    // When every scaled coordinate fits 15 bits (after an offset) a point is one
    // uint64: (major << 15 | minor) << 24 | index.  Sorting then needs no float
    // look-ups; comparisons of the reference's quicksort use the 30 key bits only.
    static constexpr int kIdxBits = 24;
    uint64_t pack(int32_t v, int axis) const {
        const uint64_t a = (uint64_t)((axis == 0 ? ix_[v] : iy_[v]) + (1 << 14));
        const uint64_t b = (uint64_t)((axis == 0 ? iy_[v] : ix_[v]) + (1 << 14));
        return ((a << 15 | b) << kIdxBits) | (uint64_t)v;
    }
    // LSD radix sort on the 30 key bits (three 10-bit digits); stable
    static void radix30(std::vector<uint64_t>& a, std::vector<uint64_t>& tmp) {
        const size_t n = a.size();
        tmp.resize(n);
        uint64_t* src = a.data();
        uint64_t* dst = tmp.data();
        for (int pass = 0; pass < 3; pass++) {
            const int sh = kIdxBits + 10 * pass;
            uint32_t cnt[1025];
            memset(cnt, 0, sizeof(cnt));
            for (size_t i = 0; i < n; i++) cnt[((src[i] >> sh) & 1023) + 1]++;
            for (int d = 0; d < 1024; d++) cnt[d + 1] += cnt[d];
            for (size_t i = 0; i < n; i++) dst[cnt[(src[i] >> sh) & 1023]++] = src[i];
            std::swap(src, dst);
        }
        if (src != a.data()) memcpy(a.data(), src, n * sizeof(uint64_t));
    }
    // the reference's quicksort (vertexsort, triangle.cpp:5418-5476) on packed keys
    void sort_xy_packed(uint64_t* a, int32_t n) {
        if (n == 2) {
            if ((a[0] >> kIdxBits) > (a[1] >> kIdxBits)) std::swap(a[0], a[1]);
            return;
        }
        const uint64_t pv = a[pick((uint32_t)n)] >> kIdxBits;
        int32_t l = -1, r = n;
        // While the two scans are far apart their stops are found a block at a time without a branch per element: the
        // Hoare scans swap the k-th element >= pivot from the left with the k-th element <= pivot from the right,
        // whatever lies between, so the pairs of two blocks of kBlk positions are exactly the reference's next swaps
        // (tools: 20 000 random arrays with ties against the plain loop, pivot stream included).  The plain loop then
        // finishes from the state the blocks leave, crossing included.
        constexpr int kBlk = 16;
        while (r - l - 1 >= 2 * kBlk) {
            int32_t il[kBlk], ir[kBlk];
            int nl = 0, nr = 0;
            for (int k = 0; k < kBlk; k++) {
                il[nl] = l + 1 + k;
                nl += (a[l + 1 + k] >> kIdxBits) >= pv;
            }
            for (int k = 0; k < kBlk; k++) {
                ir[nr] = r - 1 - k;
                nr += (a[r - 1 - k] >> kIdxBits) <= pv;
            }
            if (nl == 0 || nr == 0) {   // a block without a stop: the scan passes over it
                if (nl == 0) l += kBlk;
                if (nr == 0) r -= kBlk;
                continue;
            }
            const int t = nl < nr ? nl : nr;
            for (int q = 0; q < t; q++) std::swap(a[il[q]], a[ir[q]]);
            l = il[t - 1];
            r = ir[t - 1];
        }
        while (l < r) {
            do { l++; } while (l <= r && (a[l] >> kIdxBits) < pv);
            do { r--; } while (l <= r && (a[r] >> kIdxBits) > pv);
            if (l < r) std::swap(a[l], a[r]);
        }
        if (l > 1) sort_xy_packed(a, l);
        if (r < n - 2) sort_xy_packed(a + r + 1, n - r - 1);
    }

    // The alternating-cut order is a pure function of a set of DISTINCT points (every
    // cut takes the n/2 smallest by the lexicographic key of its axis, leaves of <= 3
    // points are x-sorted), so it can be produced without the pivot stream: a
    // kd-style split over two presorted lists with stable partitions.
    // List entries are (rank in x order) << 32 | (rank in y order): membership in the lower half
    // of a cut is one compare against the pivot's rank, so the stable partition of the other
    // list is branch-free and touches nothing but the two lists.
    // `out` receives the n vertices of this sub-range, `tmp` is n entries of scratch.
    // `par`: levels whose two halves (disjoint sub-ranges of every array) run on two threads
    static void kd_order(uint64_t* lx, uint64_t* ly, int32_t n, int axis, int32_t* out, uint64_t* tmp,
                         const int32_t* by_x, int par = 0) {
        if (n <= 3) {
            for (int32_t i = 0; i < n; i++) out[i] = by_x[lx[i] >> 32];
            return;
        }
        const int32_t half = n >> 1;
        int32_t a = 0, b = half;
        if (axis == 0) {
            // lx[0 .. half) is the lower half; split ly the same way
            const uint64_t pivot = lx[half] >> 32;
            for (int32_t i = 0; i < n; i++) {
                const uint64_t e = ly[i];
                const int32_t low = (e >> 32) < pivot;
                tmp[b + ((a - b) & -low)] = e;   // a or b without a branch (random points mispredict)
                a += low;
                b += 1 - low;
            }
            memcpy(ly, tmp, sizeof(uint64_t) * n);
        } else {
            const uint64_t pivot = ly[half] & 0xffffffffu;
            for (int32_t i = 0; i < n; i++) {
                const uint64_t e = lx[i];
                const int32_t low = (e & 0xffffffffu) < pivot;
                tmp[b + ((a - b) & -low)] = e;
                a += low;
                b += 1 - low;
            }
            memcpy(lx, tmp, sizeof(uint64_t) * n);
        }
        if (par > 0 && n >= 512) {
            run_pair([&]() { kd_order(lx, ly, half, 1 - axis, out, tmp, by_x, par - 1); },
                     [&]() { kd_order(lx + half, ly + half, n - half, 1 - axis, out + half, tmp + half, by_x, par - 1); });
            return;
        }
        kd_order(lx, ly, half, 1 - axis, out, tmp, by_x);
        kd_order(lx + half, ly + half, n - half, 1 - axis, out + half, tmp + half, by_x);
    }

    void recurse(const int32_t* a, int32_t n, int axis, Handle* farleft, Handle* farright, int32_t& ctr,
                 int par_depth);
    void merge(Handle* farleft, Handle* innerleft, Handle* innerright, Handle* farright, int axis,
               int32_t& ctr);
    bool scale_coordinates();
};

// Coordinates are float; every float is a dyadic rational, so scaling all of
// them by one power of two makes the predicates exact in integers.  The
// in-circle determinant needs 4*bits+3 <= 127.
bool DivConq::scale_coordinates() {
    int min_exp = 0;  // most negative exponent of a set low bit
    float maxabs = 0.f;
    bool integral = true;
    for (int32_t i = 0; i < 2 * n_; i++) {
        float f = pts_[i];
        if (!(f == f) || std::fabs(f) > 1e9f) return false;
        maxabs = std::fabs(f) > maxabs ? std::fabs(f) : maxabs;
        if (f != (float)(int32_t)f) {
            integral = false;
            int e;
            float m = std::frexp(f, &e);  // f = m * 2^e, 0.5 <= |m| < 1
            int32_t mi = (int32_t)std::ldexp(m, 24);
            int tz = 0;
            while ((mi & 1) == 0) { mi >>= 1; tz++; }
            int low = e - 24 + tz;  // exponent of the lowest set bit
            if (low < min_exp) min_exp = low;
        }
    }
    int int_bits = 1;
    while (std::ldexp(1.0f, int_bits) <= maxabs) int_bits++;
    if (int_bits - min_exp + 1 > 30) return false;
    ix_.resize(n_);
    iy_.resize(n_);
    int64_t big = 0;
    if (integral) {   // the usual case (pixel coordinates): no scaling at all
        for (int32_t i = 0; i < n_; i++) {
            ix_[i] = (int64_t)pts_[2 * i];
            iy_[i] = (int64_t)pts_[2 * i + 1];
        }
        big = (int64_t)maxabs;
    } else {
        for (int32_t i = 0; i < n_; i++) {
            ix_[i] = (int64_t)std::ldexp((double)pts_[2 * i], -min_exp);
            iy_[i] = (int64_t)std::ldexp((double)pts_[2 * i + 1], -min_exp);
            big = std::max(big, std::max(ix_[i] < 0 ? -ix_[i] : ix_[i], iy_[i] < 0 ? -iy_[i] : iy_[i]));
        }
    }
    narrow_ = big < (1 << 14);
    return true;
}

void DivConq::recurse(const int32_t* a, int32_t n, int axis, Handle* farleft, Handle* farright, int32_t& ctr,
                      int par_depth) {
    if (n == 2) {
        // an edge: two ghost records glued along all three sides
        Handle L = make(ctr);
        set_org(L, a[0]);
        set_dest(L, a[1]);
        Handle R = make(ctr);
        set_org(R, a[1]);
        set_dest(R, a[0]);
        bond(L, R);
        L = prev(L); R = next(R);
        bond(L, R);
        L = prev(L); R = next(R);
        bond(L, R);
        *farright = R;
        *farleft = prev(R);
        return;
    }
    if (n == 3) {
        Handle mid = make(ctr), t1 = make(ctr), t2 = make(ctr), t3 = make(ctr);
        int area = ccw(a[0], a[1], a[2]);
        if (area == 0) {
            // collinear: two edges, four ghosts
            set_org(mid, a[0]); set_dest(mid, a[1]);
            set_org(t1, a[1]);  set_dest(t1, a[0]);
            set_org(t2, a[2]);  set_dest(t2, a[1]);
            set_org(t3, a[1]);  set_dest(t3, a[2]);
            bond(mid, t1);
            bond(t2, t3);
            mid = next(mid); t1 = prev(t1); t2 = next(t2); t3 = prev(t3);
            bond(mid, t3);
            bond(t1, t2);
            mid = next(mid); t1 = prev(t1); t2 = next(t2); t3 = prev(t3);
            bond(mid, t1);
            bond(t2, t3);
            *farleft = t1;
            *farright = t2;
        } else {
            // one real triangle (mid) fenced by three ghosts
            int32_t p = area > 0 ? a[1] : a[2];
            int32_t q = area > 0 ? a[2] : a[1];
            set_org(mid, a[0]); set_dest(t1, a[0]); set_org(t3, a[0]);
            set_dest(mid, p);   set_org(t1, p);     set_dest(t2, p);
            set_apex(mid, q);   set_org(t2, q);     set_dest(t3, q);
            bond(mid, t1);
            mid = next(mid);
            bond(mid, t2);
            mid = next(mid);
            bond(mid, t3);
            t1 = prev(t1); t2 = next(t2);
            bond(t1, t2);
            t1 = prev(t1); t3 = prev(t3);
            bond(t1, t3);
            t2 = next(t2); t3 = prev(t3);
            bond(t2, t3);
            *farleft = t1;
            *farright = area > 0 ? t2 : next(t1);
        }
        return;
    }
    int32_t half = n >> 1;
    Handle innerleft, innerright;
    if (par_depth > 0 && n >= 512) {
        // the right half on another thread, with the record range the sequential order gives it
        int32_t ctr_right = ctr + 2 * half - 2;
        run_pair([&]() { recurse(a, half, 1 - axis, farleft, &innerleft, ctr, par_depth - 1); },
                 [&]() { recurse(a + half, n - half, 1 - axis, &innerright, farright, ctr_right, par_depth - 1); });
        ctr = ctr_right;   // == start + 2n - 4: both halves are complete
    } else {
        recurse(a, half, 1 - axis, farleft, &innerleft, ctr, 0);
        recurse(a + half, n - half, 1 - axis, &innerright, farright, ctr, 0);
    }
    merge(farleft, &innerleft, &innerright, farright, axis, ctr);
}

void DivConq::merge(Handle* farleft, Handle* innerleft, Handle* innerright, Handle* farright,
                    int axis, int32_t& ctr) {
    int32_t il_dest = dest(*innerleft), il_apex = apex(*innerleft);
    int32_t ir_org = org(*innerright), ir_apex = apex(*innerright);

    if (axis == 1) {
        // horizontal cut: walk the four extreme handles from leftmost/rightmost
        // to bottommost/topmost (undone at the end)
        int32_t fl_pt = org(*farleft), fl_apex = apex(*farleft);
        int32_t fr_pt = dest(*farright);
        while (Y(fl_apex) < Y(fl_pt)) {
            *farleft = sym(next(*farleft));
            fl_pt = fl_apex;
            fl_apex = apex(*farleft);
        }
        Handle chk = sym(*innerleft);
        int32_t cv = apex(chk);
        while (Y(cv) > Y(il_dest)) {
            *innerleft = next(chk);
            il_apex = il_dest;
            il_dest = cv;
            chk = sym(*innerleft);
            cv = apex(chk);
        }
        while (Y(ir_apex) < Y(ir_org)) {
            *innerright = sym(next(*innerright));
            ir_org = ir_apex;
            ir_apex = apex(*innerright);
        }
        chk = sym(*farright);
        cv = apex(chk);
        while (Y(cv) > Y(fr_pt)) {
            *farright = next(chk);
            fr_pt = cv;
            chk = sym(*farright);
            cv = apex(chk);
        }
    }
    // lower common tangent
    bool moved;
    do {
        moved = false;
        if (ccw(il_dest, il_apex, ir_org) > 0) {
            *innerleft = sym(prev(*innerleft));
            il_dest = il_apex;
            il_apex = apex(*innerleft);
            moved = true;
        }
        if (ccw(ir_apex, ir_org, il_dest) > 0) {
            *innerright = sym(next(*innerright));
            ir_org = ir_apex;
            ir_apex = apex(*innerright);
            moved = true;
        }
    } while (moved);

    Handle lcand = sym(*innerleft), rcand = sym(*innerright);
    // bottom ghost of the seam
    Handle base = make(ctr);
    bond(base, *innerleft);
    base = next(base);
    bond(base, *innerright);
    base = next(base);
    set_org(base, ir_org);
    set_dest(base, il_dest);
    if (il_dest == org(*farleft)) *farleft = next(base);
    if (ir_org == dest(*farright)) *farright = prev(base);

    int32_t lowerleft = il_dest, lowerright = ir_org;
    int32_t upperleft = apex(lcand), upperright = apex(rcand);
    for (;;) {
        bool leftdone = ccw(upperleft, lowerleft, lowerright) <= 0;
        bool rightdone = ccw(upperright, lowerleft, lowerright) <= 0;
        if (leftdone && rightdone) {
            // top ghost of the seam
            Handle top = make(ctr);
            set_org(top, lowerleft);
            set_dest(top, lowerright);
            bond(top, base);
            top = next(top);
            bond(top, rcand);
            top = next(top);
            bond(top, lcand);
            if (axis == 1) {
                // restore leftmost / rightmost anchors
                int32_t fl_pt = org(*farleft);
                int32_t fr_pt = dest(*farright), fr_apex = apex(*farright);
                Handle chk = sym(*farleft);
                int32_t cv = apex(chk);
                while (X(cv) < X(fl_pt)) {
                    *farleft = prev(chk);
                    fl_pt = cv;
                    chk = sym(*farleft);
                    cv = apex(chk);
                }
                while (X(fr_apex) > X(fr_pt)) {
                    *farright = sym(prev(*farright));
                    fr_pt = fr_apex;
                    fr_apex = apex(*farright);
                }
            }
            return;
        }
        if (!leftdone) {
            // strip left-side edges that fail the in-circle test (flip in place)
            Handle nx = sym(prev(lcand));
            int32_t nap = apex(nx);
            if (nap >= 0) {
                bool bad = incircle(lowerleft, lowerright, upperleft, nap) > 0;
                while (bad) {
                    nx = next(nx);
                    Handle topc = sym(nx);
                    nx = next(nx);
                    Handle sidec = sym(nx);
                    bond(nx, topc);
                    bond(lcand, sidec);
                    lcand = next(lcand);
                    Handle outerc = sym(lcand);
                    nx = prev(nx);
                    bond(nx, outerc);
                    set_org(lcand, lowerleft);
                    set_dest(lcand, -1);
                    set_apex(lcand, nap);
                    set_org(nx, -1);
                    set_dest(nx, upperleft);
                    set_apex(nx, nap);
                    upperleft = nap;
                    nx = sidec;
                    nap = apex(nx);
                    bad = nap >= 0 && incircle(lowerleft, lowerright, upperleft, nap) > 0;
                }
            }
        }
        if (!rightdone) {
            Handle nx = sym(next(rcand));
            int32_t nap = apex(nx);
            if (nap >= 0) {
                bool bad = incircle(lowerleft, lowerright, upperright, nap) > 0;
                while (bad) {
                    nx = prev(nx);
                    Handle topc = sym(nx);
                    nx = prev(nx);
                    Handle sidec = sym(nx);
                    bond(nx, topc);
                    bond(rcand, sidec);
                    rcand = prev(rcand);
                    Handle outerc = sym(rcand);
                    nx = next(nx);
                    bond(nx, outerc);
                    set_org(rcand, -1);
                    set_dest(rcand, lowerright);
                    set_apex(rcand, nap);
                    set_org(nx, upperright);
                    set_dest(nx, -1);
                    set_apex(nx, nap);
                    upperright = nap;
                    nx = sidec;
                    nap = apex(nx);
                    bad = nap >= 0 && incircle(lowerleft, lowerright, upperright, nap) > 0;
                }
            }
        }
        if (leftdone ||
            (!rightdone && incircle(upperleft, lowerleft, lowerright, upperright) > 0)) {
            // new edge lowerleft -> upperright
            bond(base, rcand);
            base = prev(rcand);
            set_dest(base, lowerleft);
            lowerright = upperright;
            rcand = sym(base);
            upperright = apex(rcand);
        } else {
            // new edge upperleft -> lowerright (also taken on a co-circular tie)
            bond(base, lcand);
            base = next(lcand);
            set_org(base, lowerright);
            lowerleft = upperleft;
            lcand = sym(base);
            upperleft = apex(lcand);
        }
    }
}

int32_t DivConq::run(int32_t* out, int32_t cap) {
    if (n_ < 2) return 0;
    TICK(0)
    // helpers poll for the halves from here on (they are needed some tens of microseconds from now: the y-sort below,
    // then the cut order, then the recursion); a section that is already warm (the caller asked: Matcher) is left alone
    struct Warm {
        bool mine;
        explicit Warm(int depth) : mine(depth > 0 && helper_pool().warm_until.load(std::memory_order_relaxed) < mono_ns()) {
            if (mine) helpers_warm((1 << depth) - 1, 2000);
        }
        ~Warm() { if (mine) helpers_warm(0, 0); }
    } warm_guard(par_depth_);
    if (!scale_coordinates()) return SVH_ERR_UNSUPPORTED;
    // x-sort.  For distinct points the sorted order is unique, so a plain sort is
    // used; only when coincident points exist does the survivor depend on the
    // reference's pivot stream, and the mirrored quicksort is run instead.
    std::vector<int32_t>& order = order_;
    order.resize(n_);
    const bool packed = narrow_ && n_ < (1 << kIdxBits);
    std::vector<uint64_t>&keys = keys_, &ktmp = ktmp_;
    bool dup = false;
    // the y-order of ALL points, on a helper while this thread sorts by x (parallel runs; vertices that turn out to
    // be dropped as coincident are filtered from it afterwards: a stable sort, and one survivor per position)
    const bool y_ahead = packed && par_depth_ > 0 && n_ >= 512;
    std::atomic<int> y_done{0};
    bool y_async = false;
    auto y_sort_all = [this]() {
        ykeys_.resize(n_);
        for (int32_t i = 0; i < n_; i++) ykeys_[i] = pack(i, 1);
        radix30(ykeys_, ytmp_);
    };
    if (y_ahead) {
        y_async = helper_pool().submit(y_sort_all, &y_done);
        if (!y_async) y_sort_all();
    }
    if (packed) {
        keys.resize(n_);
        for (int32_t i = 0; i < n_; i++) keys[i] = pack(i, 0);
        if (expect_dups_) {
            // (the mirrored quicksort is right with and without coincident points; the radix sort ahead of it only
            // spares it when there are none -- a Matcher's vote nearly always has some)
            sort_xy_packed(keys.data(), n_);
            for (int32_t j = 1; j < n_ && !dup; j++) dup = (keys[j - 1] >> kIdxBits) == (keys[j] >> kIdxBits);
        } else {
            radix30(keys, ktmp);
            for (int32_t j = 1; j < n_ && !dup; j++) dup = (keys[j - 1] >> kIdxBits) == (keys[j] >> kIdxBits);
            if (dup) {
                for (int32_t i = 0; i < n_; i++) keys[i] = pack(i, 0);
                sort_xy_packed(keys.data(), n_);
            }
        }
        for (int32_t i = 0; i < n_; i++) order[i] = (int32_t)(keys[i] & ((1u << kIdxBits) - 1));
    } else {
        for (int32_t i = 0; i < n_; i++) order[i] = i;
        auto less_xy = [&](int32_t a, int32_t b) {
            return X(a) < X(b) || (X(a) == X(b) && Y(a) < Y(b));
        };
        std::sort(order.begin(), order.end(), less_xy);
        for (int32_t j = 1; j < n_ && !dup; j++)
            dup = X(order[j - 1]) == X(order[j]) && Y(order[j - 1]) == Y(order[j]);
        if (dup) {
            for (int32_t i = 0; i < n_; i++) order[i] = i;
            sort_xy(order.data(), n_);
        }
    }
    int32_t m = n_;
    std::vector<int32_t>& rank = rank_;
    rank.resize(n_);
    if (dup) {
        // drop coincident vertices: the first in sorted order survives
        if (y_ahead)
            for (int32_t i = 0; i < n_; i++) rank[i] = -1;   // (marks the dropped ones for the y list)
        m = 0;
        for (int32_t j = 1; j < n_; j++) {
            if (X(order[m]) == X(order[j]) && Y(order[m]) == Y(order[j])) continue;
            order[++m] = order[j];
        }
        m++;
        if (y_ahead)
            for (int32_t i = 0; i < m; i++) rank[order[i]] = 0;
    }
    if (y_async)
        while (!y_done.load(std::memory_order_acquire)) __builtin_ia32_pause();
    if (m < 2) return 0;
    TICK(1)
    {
        // alternating-cut order (triangle.cpp:5582-5604, 6198-6206): top cut by x at m/2,
        // then each half starts with a y cut
        std::vector<int32_t>&ly = ly_, &kd = kd_;
        ly.resize(m);
        kd.resize(m);
        if (y_ahead) {
            int32_t w = 0;
            for (int32_t i = 0; i < n_; i++) {
                const int32_t v = (int32_t)(ykeys_[i] & ((1u << kIdxBits) - 1));
                if (!dup || rank[v] == 0) ly[w++] = v;
            }
        } else if (packed) {
            keys.resize(m);
            for (int32_t i = 0; i < m; i++) keys[i] = pack(order[i], 1);
            radix30(keys, ktmp);
            for (int32_t i = 0; i < m; i++) ly[i] = (int32_t)(keys[i] & ((1u << kIdxBits) - 1));
        } else {
            memcpy(ly.data(), order.data(), sizeof(int32_t) * m);
            std::sort(ly.begin(), ly.end(), [&](int32_t a, int32_t b) {
                return Y(a) < Y(b) || (Y(a) == Y(b) && X(a) < X(b));
            });
        }
        // ranks: lx[i] = i << 32 | yrank, ly[j] = xrank << 32 | j
        std::vector<uint64_t>&lx = lx_, &lyk = lyk_, &scratch = scratch_;
        lx.resize(m);
        lyk.resize(m);
        scratch.resize(m);
        for (int32_t j = 0; j < m; j++) rank[ly[j]] = j;
        for (int32_t i = 0; i < m; i++) lx[i] = (uint64_t)i << 32 | (uint32_t)rank[order[i]];
        for (int32_t i = 0; i < m; i++) rank[order[i]] = i;
        for (int32_t j = 0; j < m; j++) lyk[j] = (uint64_t)rank[ly[j]] << 32 | (uint32_t)j;
        kd_order(lx.data(), lyk.data(), m, 0, kd.data(), scratch.data(), order.data(), par_depth_);
        memcpy(order.data(), kd.data(), sizeof(int32_t) * m);
    }
    TICK(2)
    // (every record is written in full by make() before anything reads it: no zero-fill)
    rec_.resize(8 * (2 * (size_t)m + 16));
    nrec_ = 0;
    make(nrec_);  // record 0 = outer space
    Handle hullleft, hullright;
    recurse(order.data(), m, 0, &hullleft, &hullright, nrec_, par_depth_);

    // peel the ghost fan off the hull
    TICK(3)
    Handle g = hullleft;
    do {
        Handle dying = next(g);
        g = sym(prev(g));
        rec_[8 * g.t + 3 + g.o] = 0;  // hull triangle now faces outer space
        g = sym(dying);
        rec_[8 * dying.t + 6] = 1;
    } while (!(g.t == hullleft.t && g.o == hullleft.o));

    TICK(4)
    int32_t count = 0;
    const int32_t nrec = nrec_;
    for (int32_t t = 1; t < nrec; t++) {
        const int32_t* r = &rec_[8 * (size_t)t];
        if (r[6]) continue;
        if (count < cap) {
            out[3 * count + 0] = r[1];
            out[3 * count + 1] = r[2];
            out[3 * count + 2] = r[0];
        }
        count++;
    }
    TICK(5)
    return count;
}

}  // namespace

// Every thread keeps the DivConq object of its last call: the arrays stay where that thread's cache has them (an object
// shared between threads was measured: the next user pulls every line out of the last user's cache -- lockstep votes
// on 16-32 pool threads 6-13 % slower than with fresh allocations).  An object that has grown beyond kKeepBytes (a
// 1920x1080 pair's support points need ~3 MB) is not kept.
int32_t delaunay(const float* pts, int32_t n, int32_t* tri, int32_t cap, int par_depth, bool expect_dups) {
    static constexpr size_t kKeepBytes = 16u << 20;
    static thread_local std::unique_ptr<DivConq> t_dc;
    std::unique_ptr<DivConq> dc = std::move(t_dc);   // (moved out: a nested call on this thread makes its own)
    if (!dc) dc.reset(new DivConq());
    dc->reset(pts, n, par_depth, expect_dups);
    const int32_t nt = dc->run(tri, cap);
    if (dc->bytes_held() <= kKeepBytes) t_dc = std::move(dc);
    return nt;
}

}  // namespace svh

extern "C" void svh_host_helper_stats(int64_t out[4]) {
    if (!out) return;
    svh::HelperPool& p = svh::helper_pool();
    out[0] = p.n_tasks.load();
    out[1] = p.n_moves.load();
    out[2] = p.n_hot.load();
    out[3] = p.n_woken.load();
}

extern "C" int32_t svh_delaunay(const float* pts, int32_t n, int32_t* tri, int32_t cap) {
    if (!pts || !tri || n < 0) return SVH_ERR_BAD_ARG;
    return svh::delaunay(pts, n, tri, cap, 0, false);
}

extern "C" int32_t svh_delaunay_mt(const float* pts, int32_t n, int32_t* tri, int32_t cap, int32_t par_depth) {
    if (!pts || !tri || n < 0 || par_depth < 0 || par_depth > 4) return SVH_ERR_BAD_ARG;
    return svh::delaunay(pts, n, tri, cap, par_depth, false);
}
