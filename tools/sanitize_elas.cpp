// Sanitizer driver for the threaded HOST logic of the ELAS engine (csrc/elas_engine.cpp: lane pool, batch crew,
// double-buffered workers, streams with persistent workers that take lanes only while they have groups in flight,
// close() against blocked producers, fault drains) over a STUB device layer: this file defines the HIP entry points
// and the kernel launchers the engine uses (host memory, events that complete after a few queries, launchers that
// fill the counts the engine reads back with plausible values).  No GPU, no libamdhip64: CPU only.
//   make -C stereo-vision_amd sanitize_elas     -fsanitize=thread and -fsanitize=address,undefined, both with and
//                                               without injected HIP failures (SVH_TEST_FAIL_AT of the real library)
// What runs at once: four threads of single svh_elas_process calls, two threads of batch calls (host outputs and
// device outputs), two streams on one device with two producers and one consumer each -- one of them closed while
// a producer is blocked on its full queue -- and svh_elas_trim() from yet another thread.
// The maps are not checked (the kernels are stubs); statuses, tickets and their order are.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../stereo-vision_amd/csrc/svh_internal.h"

// ---------------------------------------------------------------- stub HIP runtime
struct StubStream {
    std::atomic<int> pending{0};   // queries that still answer "not ready"
};
struct StubEvent {
    std::atomic<int> pending{0};
};
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)16 << 30; return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
    memcpy(d, s, n);
    if (st) reinterpret_cast<StubStream*>(st)->pending.store(2, std::memory_order_relaxed);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st) {
    for (size_t r = 0; r < h; r++) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    if (st) reinterpret_cast<StubStream*>(st)->pending.store(2, std::memory_order_relaxed);
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t*, const void*) { return hipErrorInvalidValue; }   // "pageable"
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = reinterpret_cast<hipStream_t>(new StubStream()); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<StubStream*>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { if (s) reinterpret_cast<StubStream*>(s)->pending.store(0, std::memory_order_relaxed); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(new StubEvent()); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new StubEvent()); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<StubEvent*>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { reinterpret_cast<StubEvent*>(e)->pending.store(2, std::memory_order_relaxed); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { reinterpret_cast<StubEvent*>(e)->pending.store(0, std::memory_order_relaxed); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) {
    StubEvent* q = reinterpret_cast<StubEvent*>(e);
    const int left = q->pending.load(std::memory_order_relaxed);
    if (left > 0) { q->pending.store(left - 1, std::memory_order_relaxed); return hipErrorNotReady; }
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.01f; return hipSuccess; }
}

// ---------------------------------------------------------------- stub launchers
namespace svh {
bool stage_device_ok(const svh_elas_params&, const Dims&) { return true; }
bool stage_device_preferred(const svh_elas_params&, const Dims&, bool) { return true; }
void launch_stage_device(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t g, const StageDev& S, GroupHdr* hdr,
                         int32_t*, int32_t*) {
    // what k_lattice / k_delaunay / k_stage_pack leave behind: every third pair of a group has too few points
    memset(hdr, 0, sizeof(GroupHdr));
    hdr->npairs = g;
    for (int32_t j = 0; j < g; j++) {
        const bool few = j % 3 == 2;
        S.counts->nsup[j] = few ? 2 : 40;
        S.counts->ntri[2 * j] = S.counts->ntri[2 * j + 1] = few ? 0 : 60;
        S.counts->flags[j] = few ? STG_FEW : 0;
        hdr->active[j] = few ? 0 : 1;
    }
}
void launch_descriptor(const LaunchCtx&, const DevImages&, int32_t, int32_t, int32_t, int32_t, uint8_t*, bool) {}
bool descriptors_on_the_fly(const svh_elas_params&, const Dims&, int32_t, int32_t, bool) { return true; }
void launch_support(const LaunchCtx&, const svh_elas_params&, const Dims& d, int32_t g, const uint8_t*, int16_t* dcan, bool) {
    // a lattice with a block of consistent candidates: the host stage finds support points in it
    const size_t nc = (size_t)d.Wc * d.Hc;
    for (int32_t j = 0; j < g; j++)
        for (size_t i = 0; i < nc; i++) dcan[j * nc + i] = (int16_t)(j % 3 == 2 ? -1 : 20 + (int)(i % 3));
}
void launch_prior(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t, int32_t, int32_t, const GroupDev&) {}
void launch_owner(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t, int32_t, const GroupDev&) {}
bool launch_match(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t, const GroupDev&, const DevMaps*, bool,
                  const char** error) {
    if (error) *error = nullptr;
    return true;
}
void launch_lr(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t, const GroupDev&, const DevMaps&) {}
bool post_tiles_ok(const svh_elas_params&) { return true; }
void launch_gap_mean_tiles(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t, int32_t, const GroupDev&, const DevMaps&,
                           const PostScratch&) {}
void launch_segments(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t, int32_t, const GroupDev&, const DevMaps&,
                     const PostScratch&, bool) {}
void launch_gap(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t, int32_t, const GroupDev&, const DevMaps&,
                const PostScratch&) {}
void launch_adaptive_mean(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t, int32_t, const GroupDev&, const DevMaps&,
                          const PostScratch&) {}
void launch_median(const LaunchCtx&, const Dims&, int32_t, int32_t, const GroupDev&, const DevMaps&, const PostScratch&) {}
void launch_segments_label(const LaunchCtx&, const svh_elas_params&, const Dims&, int32_t, int32_t, const GroupDev&,
                           const DevMaps&, const PostScratch&) {}
}  // namespace svh

extern "C" int32_t svh_test_fail_at(const char* spec);   // test hook of the engines (csrc/svh_internal.h)
// ---------------------------------------------------------------- driver
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 12;
    // (the library no longer reads a fault specification from the environment: the driver passes its own on)
    const char* inject = getenv("SVH_TEST_FAIL_AT");
    if (inject && svh_test_fail_at(inject) != SVH_OK) { fprintf(stderr, "bad SVH_TEST_FAIL_AT\n"); return 2; }
    const int W = 320, H = 120;
    const int32_t dims[3] = {W, H, W};
    svh_elas_params prm;
    svh_elas_params_default(&prm, SVH_ELAS_ROBOTICS);
    std::atomic<int> bad{0};
    std::atomic<long> ok{0}, few{0}, injected{0};
    auto check = [&](int32_t rc) {
        if (rc == SVH_OK) ok++;
        else if (rc == SVH_ERR_FEW_SUPPORT) few++;
        else if (inject && rc == SVH_ERR_HIP) injected++;
        else bad++;
    };
    svh_elas_set_lanes(3);
    svh_elas_set_group(4);
    std::vector<uint8_t> I1((size_t)W * H, 90), I2((size_t)W * H, 91);
    std::vector<std::thread> th;
    // single calls
    for (int t = 0; t < 4; t++)
        th.emplace_back([&, t] {
            svh_elas* e = svh_elas_create(&prm);
            std::vector<float> D1((size_t)W * H), D2((size_t)W * H);
            for (int r = 0; r < 3 * rounds; r++) check(svh_elas_process(e, I1.data(), I2.data(), D1.data(), D2.data(), dims));
            svh_elas_destroy(e);
            (void)t;
        });
    // batch calls: host buffers (host stage by the automatic choice of a shallow batch: forced per thread), device buffers
    for (int t = 0; t < 2; t++)
        th.emplace_back([&, t] {
            svh_elas* e = svh_elas_create(&prm);
            const int n = 22;
            std::vector<std::vector<float>> D1(n, std::vector<float>((size_t)W * H)), D2 = D1;
            std::vector<const uint8_t*> p1(n, I1.data()), p2(n, I2.data());
            std::vector<float*> d1(n), d2(n);
            for (int i = 0; i < n; i++) { d1[i] = D1[i].data(); d2[i] = D2[i].data(); }
            std::vector<int32_t> st(n);
            std::vector<uint8_t> B1((size_t)n * W * H, 90), B2((size_t)n * W * H, 91);
            std::vector<float> O1((size_t)n * W * H), O2((size_t)n * W * H);
            for (int r = 0; r < rounds; r++) {
                int32_t rc;
                if (t == 0) rc = svh_elas_process_batch(e, n, p1.data(), p2.data(), d1.data(), d2.data(), dims, st.data());
                else rc = svh_elas_process_batch_device(e, n, B1.data(), B2.data(), (size_t)W * H, O1.data(), O2.data(),
                                                        (size_t)W * H * sizeof(float), dims, st.data());
                if (rc < 0 && !(inject && rc == SVH_ERR_HIP)) bad++;
                for (int i = 0; i < n; i++) check(st[i]);
            }
            svh_elas_destroy(e);
        });
    // two streams on the device, two producers + one consumer each; the second one is closed under a blocked producer
    for (int sidx = 0; sidx < 2; sidx++)
        th.emplace_back([&, sidx] {
            svh_elas* e = svh_elas_create(&prm);
            for (int r = 0; r < rounds; r++) {
                svh_elas_stream* s = svh_elas_stream_open(e, dims, sidx == 0 ? 0 : 3);
                if (!s) { bad++; continue; }
                const int per = 15;
                std::vector<std::vector<float>> D1(2 * per, std::vector<float>((size_t)W * H)), D2 = D1;
                std::atomic<int> pushed{0};
                auto producer = [&](int k) {
                    for (int i = 0; i < per; i++) {
                        uint64_t t = 0;
                        const int slot = k * per + i;
                        const int32_t rc = svh_elas_stream_push(s, I1.data(), I2.data(), D1[slot].data(), D2[slot].data(), &t);
                        if (rc == SVH_OK) pushed++;
                        else if (rc != SVH_ERR_BAD_ARG) bad++;     // BAD_ARG: "stream is closing" (sidx 1)
                        else break;
                    }
                };
                std::thread pa(producer, 0), pb(producer, 1);
                if (sidx == 0) {
                    // consumer: pops everything in ticket order
                    uint64_t expect = 0;
                    for (int got = 0; got < 2 * per;) {
                        uint64_t t = 0;
                        int32_t st = 0;
                        // (no timeout: gcc 11's libtsan does not intercept pthread_cond_clockwait, the timed wait of
                        // std::condition_variable, and then reports the mutex as held twice)
                        const int32_t rc = svh_elas_stream_pop(s, &t, &st, -1);
                        if (rc == SVH_ERR_EMPTY) { std::this_thread::yield(); continue; }
                        if (rc != SVH_OK || t != expect) bad++;
                        expect++;
                        got++;
                        check(st);
                    }
                    pa.join();
                    pb.join();
                    svh_elas_stream_close(s);
                } else {
                    // nobody pops: depth 3 fills, both producers block inside their next push; close() must wake them
                    // and return (a producer may not push after close() has returned: wait until the queue is full
                    // and both have had ample time to enter the push that blocks)
                    while (pushed.load() < 3) std::this_thread::sleep_for(std::chrono::milliseconds(1));
                    std::this_thread::sleep_for(std::chrono::milliseconds(20));
                    svh_elas_stream_close(s);
                    pa.join();
                    pb.join();
                    if (pushed.load() > 2 * per) bad++;
                }
            }
            svh_elas_destroy(e);
        });
    // the pool is trimmed meanwhile
    std::atomic<bool> stop{false};
    std::thread trimmer([&] {
        while (!stop.load()) {
            (void)svh_elas_trim();
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    });
    for (std::thread& t : th) t.join();
    stop.store(true);
    trimmer.join();
    (void)svh_elas_trim();
    printf("sanitize_elas: %d rounds: 4 single-call threads, 2 batch threads, 2 streams x (2 producers + consumer / close under "
           "blocked producers), trim thread: %ld ok, %ld with too few support points, %ld injected HIP failures reported, %d failures\n",
           rounds, ok.load(), few.load(), injected.load(), bad.load());
    return bad.load() ? 1 : 0;
}
