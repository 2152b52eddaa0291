// Sanitizer driver for the threaded HOST logic of the Matcher / visual-odometry engines
// (matcher_engine.cpp, vo_engine.cpp: ring buffer, pinned count read-backs, sleep-polling waits chosen by the
// number of concurrent callers, helper-pool bypass of the outlier vote, bucketing, RANSAC bookkeeping) with
// a STUB device layer: this file defines the HIP entry points those two files use (host memory, streams
// that complete after a few queries) and their kernel launchers (deterministic synthetic feature tables,
// matches and motion results of realistic sizes).  No GPU, no libamdhip64: CPU only.
//   make -C stereo-vision_amd sanitize_viso    builds it with -fsanitize=thread and with
//                                              -fsanitize=address,undefined and runs both, K = 16 sequences
// What is checked: data races and memory errors of K VisualOdometryStereo objects driven from K threads at
// once (SURVEY 8(e) "replicas only"), one thread that creates and destroys Matchers meanwhile, and two threads
// that each drive K/2 objects in lockstep through svh_vo_process_batch (recorder, helper pool, phase barriers).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../stereo-vision_amd/csrc/matcher_internal.h"
#include "../stereo-vision_amd/csrc/vo_internal.h"

// ---------------------------------------------------------------- stub HIP runtime
struct StubStream {
    std::atomic<int> pending{0};   // queries that still answer "not ready"
};
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
    memcpy(d, s, n);
    if (st) reinterpret_cast<StubStream*>(st)->pending.store(2, std::memory_order_relaxed);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = reinterpret_cast<hipStream_t>(new StubStream()); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<StubStream*>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { if (s) reinterpret_cast<StubStream*>(s)->pending.store(0, std::memory_order_relaxed); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new int(0)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<int*>(e); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(new int(0)); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.01f; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) {
    if (!s) return hipSuccess;
    StubStream* q = reinterpret_cast<StubStream*>(s);
    const int left = q->pending.load(std::memory_order_relaxed);
    if (left > 0) { q->pending.store(left - 1, std::memory_order_relaxed); return hipErrorNotReady; }
    return hipSuccess;
}
}

namespace svh {
static thread_local std::string t_err;
int fail(int code, const std::string& msg) { t_err = msg; return code; }
// fault injection under the sanitizers: SVH_SAN_FAIL_EVERY=n makes every n-th HIP call the engines check fail
// (any kind), so that their error paths -- early returns out of recording passes, helper-pool phases, the prefetch
// hand-over -- run under TSan / ASan with 16 threads around them
static const long g_fi_every = getenv("SVH_SAN_FAIL_EVERY") ? atol(getenv("SVH_SAN_FAIL_EVERY")) : 0;
static std::atomic<long> g_fi_n{0};
bool fi_armed() { return g_fi_every > 0; }
bool fi_hit(const char*) { return g_fi_every > 0 && (g_fi_n.fetch_add(1) + 1) % g_fi_every == 0; }
void report_hip_failure(const char*) {}
}  // namespace svh
extern "C" const char* svh_last_error(void) { return svh::t_err.c_str(); }
namespace svh {

// ---------------------------------------------------------------- stub launchers
static uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// (the stub launchers run at once even while a batch is being recorded: the recorder stays empty, its flush is a no-op)
void mlaunch_upload(void*, const uint8_t* pinned, uint8_t* dev, size_t bytes) { memcpy(dev, pinned, bytes); }
void mlaunch_copy(void*, void* dst, const void* src, size_t bytes, int) { memcpy(dst, src, bytes); }
void mlaunch_fill(void*, void* dst, int v, size_t bytes) { memset(dst, v, bytes); }
void mlaunch_half(void*, const uint8_t*, int, uint8_t*, int, int, int) {}
void mlaunch_filters(void*, const uint8_t*, int, int, int, uint8_t*, uint8_t*, int16_t*, int16_t*) {}
void mlaunch_half_filters(void*, const uint8_t*, int, int, int, uint8_t*, int, int, int, uint8_t*, uint8_t*) {}
int mnms_blocks(int extent, int n, int margin) { const int e = extent - 2 * margin; return e <= 0 ? 0 : (e + n) / (n + 1); }
void mlaunch_features(void*, const int16_t*, const int16_t*, const uint8_t* du, const uint8_t*, int w, int h, int, int n,
                      int, int margin, int scale, int4*, int32_t*, int32_t*, int32_t* table, int32_t* count) {
    // a plausible table: one feature per second NMS block, classes cycling, descriptor words from a hash
    const int bx = mnms_blocks(w, n, margin), by = mnms_blocks(h, n, margin);
    int k = 0;
    const uint32_t seed = mix((uint32_t)(uintptr_t)du ^ (uint32_t)n);
    for (int y = 0; y < by; y++)
        for (int x = (y & 1); x < bx; x += 2) {
            int32_t* r = table + 12 * k;
            r[0] = (margin + x * (n + 1) + (int)(mix(seed + k) % (unsigned)(n + 1))) * scale;
            r[1] = (margin + y * (n + 1) + (int)(mix(seed + 7 * k) % (unsigned)(n + 1))) * scale;
            r[2] = 0;
            r[3] = k & 3;
            for (int q = 4; q < 12; q++) r[q] = (int32_t)mix(seed + 13 * k + q);
            k++;
        }
    *count = k;
}
void mlaunch_features2(void* st, const int16_t* f1, const int16_t* f2, const uint8_t* du, const uint8_t* dv, int w, int h,
                       int bpl, int tau, int margin, int scale, int n_a, int4* sa, int32_t* fa, int32_t* oa, int32_t* ta,
                       int32_t* ca, int n_b, int4* sb, int32_t* fb, int32_t* ob, int32_t* tb, int32_t* cb, int32_t* host_counts) {
    mlaunch_features(st, f1, f2, du, dv, w, h, bpl, n_a, tau, margin, scale, sa, fa, oa, ta, ca);
    mlaunch_features(st, f1, f2, du, dv, w, h, bpl, n_b, tau, margin, scale, sb, fb, ob, tb, cb);
    host_counts[0] = *ca;
    host_counts[1] = *cb;
}
void mlaunch_bin_index(void*, const BinJobs&, int, int, int, int, int, int32_t*) {}
void mlaunch_match(void*, const MatchParams& P, const FeatView& m1p, const FeatView&, const FeatView& m1c,
                   const FeatView&, int nquery_cap, const float*, int, svh_p_match*, int32_t*, int32_t*,
                   svh_p_match* out, int32_t* out_count, int32_t* out_count_host) {
    // every third feature of the current left image "matches": consistent small flow and disparity
    const int nc = *m1c.count, np = *m1p.count;
    int k = 0;
    for (int i = 0; i < nc && k < nquery_cap; i += 3) {
        const int32_t* r = m1c.rec + 12 * i;
        svh_p_match m;
        memset(&m, 0, sizeof(m));
        const float d = 8.f + (float)(mix((uint32_t)i) % 40u), fu = (float)(mix((uint32_t)i * 3u) % 5u) - 2.f;
        m.u1c = (float)r[0]; m.v1c = (float)r[1]; m.i1c = i;
        m.u2c = m.u1c - d;   m.v2c = m.v1c;       m.i2c = i;
        m.u1p = m.u1c + fu;  m.v1p = m.v1c + 1.f; m.i1p = np > 0 ? i % np : 0;
        m.u2p = m.u1p - d;   m.v2p = m.v1p;       m.i2p = m.i1p;
        if (P.method < 2) { m.u2c = m.u2p = -1; m.v2c = m.v2p = -1; m.i2c = m.i2p = -1; }
        out[k++] = m;
    }
    *out_count = k;
    if (out_count_host) *out_count_host = k;
}
void mlaunch_refine(void*, svh_p_match* m, const int32_t* count, int, int, int, const SobelView&, const SobelView&,
                    const SobelView&, const SobelView&, int parabolic, int32_t*, svh_p_match* compacted,
                    int32_t* compacted_count) {
    if (!parabolic) return;
    memcpy(compacted, m, sizeof(svh_p_match) * (size_t)*count);
    *compacted_count = *count;
}
void vlaunch_upload(void*, const uint8_t* pinned, uint8_t* dev, size_t bytes) { memcpy(dev, pinned, bytes); }
void vlaunch_estimate(void*, const svh_p_match*, int N, const int32_t*, int, const VoCalib&, double*, int32_t*, uint8_t*,
                      double*, double*, VoResult* out, int32_t* out_inliers) {
    out->success = N >= 6;
    out->n_inliers = N / 2;
    out->best = N >= 6 ? 0 : -1;
    for (int k = 0; k < 6; k++) out->tr[k] = 1e-3 * (k + 1);
    for (int i = 0; i < N / 2; i++) out_inliers[i] = 2 * i;
}
}  // namespace svh

// ---------------------------------------------------------------- driver
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 16, frames = argc > 2 ? atoi(argv[2]) : 12;
    const int W = 640, H = 200;
    std::atomic<int> bad{0};
    std::atomic<long> matches{0}, injected{0};
    const bool inject = svh::fi_armed();
    // a negative return is a failure of the run -- unless failures are being injected: then it must be SVH_ERR_HIP
    // (or the bad-argument answer of an entry whose hand-over was lost with the failed call), and is counted
    auto check = [&](int32_t rc) {
        if (rc >= 0) return;
        if (inject && (rc == SVH_ERR_HIP || rc == SVH_ERR_BAD_ARG)) injected++;
        else bad++;
    };
    auto sequence = [&](int id) {
        svh_vo_params p;
        svh_vo_params_default(&p);
        p.f = 645.2; p.cu = 320.0; p.cv = 100.0; p.base = 0.57;
        svh_vo* vo = svh_vo_create(&p);
        if (!vo) { bad++; return; }
        std::vector<uint8_t> I1((size_t)W * H), I2((size_t)W * H);
        const int32_t dims[3] = {W, H, W};
        for (int f = 0; f < frames; f++) {
            for (size_t i = 0; i < I1.size(); i++) {
                I1[i] = (uint8_t)(svh::mix((uint32_t)(i + 977 * f + 31 * id)) >> 24);
                I2[i] = (uint8_t)(svh::mix((uint32_t)(i + 977 * f + 31 * id + 5)) >> 24);
            }
            check(svh_vo_process(vo, I1.data(), I2.data(), dims, 0));
            matches += svh_vo_num_matches(vo);
        }
        svh_vo_destroy(vo);
    };
    std::vector<std::thread> th;
    for (int k = 0; k < K; k++) th.emplace_back(sequence, k);
    // a Matcher used on its own from one more thread, created and destroyed over and over meanwhile
    th.emplace_back([&] {
        for (int r = 0; r < 3 * frames; r++) {
            svh_matcher_params mp;
            svh_matcher_params_default(&mp);
            svh_matcher* m = svh_matcher_create(&mp);
            std::vector<uint8_t> I((size_t)W * H, (uint8_t)(r * 7));
            const int32_t dims[3] = {W, H, W};
            for (int f = 0; f < 2; f++) {
                check(svh_matcher_push_back(m, I.data(), I.data(), dims, 0));
                check(svh_matcher_match_features(m, 2, nullptr));
            }
            svh_matcher_destroy(m);
        }
    });
    // two more threads, each driving K/2 objects in lockstep (svh_vo_process_batch: recorder, helper pool, phases)
    auto lockstep = [&](int id) {
        const int n = K / 2 > 1 ? K / 2 : 2;
        svh_vo_params p;
        svh_vo_params_default(&p);
        p.f = 645.2; p.cu = 320.0; p.cv = 100.0; p.base = 0.57;
        std::vector<svh_vo*> vs(n);
        for (int i = 0; i < n; i++) vs[i] = svh_vo_create(&p);
        // two sets of image buffers: a frame handed over early is read (prefetch thread) until it is TAKEN by the
        // next call, so the caller fills the other set meanwhile -- the contract of svh_vo_prefetch_batch
        std::vector<std::vector<uint8_t>> Ia[2], Ib[2];
        for (int q = 0; q < 2; q++) {
            Ia[q].assign(n, std::vector<uint8_t>((size_t)W * H));
            Ib[q] = Ia[q];
        }
        std::vector<const uint8_t*> p1(n), p2(n);
        std::vector<int32_t> ok(n);
        const int32_t dims[3] = {W, H, W};
        for (int f = 0; f < frames; f++) {
            std::vector<std::vector<uint8_t>>&I1 = Ia[f & 1], &I2 = Ib[f & 1];
            for (int i = 0; i < n; i++) {
                for (size_t j = 0; j < I1[i].size(); j++) {
                    I1[i][j] = (uint8_t)(svh::mix((uint32_t)(j + 977 * f + 31 * i + 7777 * id)) >> 24);
                    I2[i][j] = (uint8_t)(svh::mix((uint32_t)(j + 977 * f + 31 * i + 7777 * id + 5)) >> 24);
                }
                p1[i] = I1[i].data();
                p2[i] = I2[i].data();
            }
            // thread 1: images with the call; thread 2: the pipelined loop (frame f handed over one call earlier)
            if (id == 1) {
                check(svh_vo_process_batch(vs.data(), n, p1.data(), p2.data(), dims, 0, ok.data()));
            } else {
                if (f == 0) {
                    check(svh_vo_prefetch_batch(vs.data(), n, p1.data(), p2.data(), dims));
                } else {
                    // (processes frame f - 1, hands over frame f)
                    const int32_t rc = svh_vo_process_next_batch(vs.data(), n, p1.data(), p2.data(), dims, 0, ok.data());
                    check(rc);
                    // a failed call may have lost the hand-over of frame f: hand it over again (refused when it is pending)
                    if (rc < 0 && inject) (void)svh_vo_prefetch_batch(vs.data(), n, p1.data(), p2.data(), dims);
                }
            }
            for (int i = 0; i < n; i++) matches += svh_vo_num_matches(vs[i]);
        }
        for (svh_vo* v : vs) svh_vo_destroy(v);
    };
    th.emplace_back(lockstep, 1);
    th.emplace_back(lockstep, 2);
    for (std::thread& t : th) t.join();
    printf("sanitize_viso: %d sequences x %d frames + 1 Matcher thread + 2 lockstep threads, %ld matches seen, %d failures"
           ", %ld injected HIP failures reported\n", K, frames, matches.load(), bad.load(), injected.load());
    return bad.load() ? 1 : 0;
}
