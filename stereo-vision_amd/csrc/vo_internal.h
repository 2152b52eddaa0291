// internal interface between vo_engine.cpp (host) and vo_kernels.hip (device)
#ifndef SVH_VO_INTERNAL_H
#define SVH_VO_INTERNAL_H
#include <stdint.h>
#include <stdlib.h>

#include "../../include/svh.h"
#include "svh_config.h"

namespace svh {

struct VoCalib {
    double f, cu, cv, base, inlier_threshold;
    int32_t reweighting;
};

struct VoResult {          // written by k_vo_refine into pinned host memory
    double tr[6];
    int32_t success;       // final Gauss-Newton converged on >= 6 inliers
    int32_t n_inliers;     // size of the winning hypothesis' inlier set (may be < 6)
    int32_t best;          // winning hypothesis, -1 = none
    int32_t pad_;
};

// Wait for a stream of a Matcher / visual-odometry object.  One or two threads inside the library's compute
// entries at this moment: the driver's spinning wait (lowest latency for the single-sequence case of
// stereomapper).  Three or more (K independent sequences driven concurrently on one GPU): polling with short
// sleeps, so that K host threads do not burn K cores spinning on a GPU they share.  The count is of
// concurrent callers (ActiveCaller below), not of objects that exist; SVH_MATCHER_WAIT=0/1 overrides.
int wait_stream(void* stream);   // returns a hipError_t value
// RAII marker of a thread inside a Matcher / visual-odometry compute entry (matcher_engine.cpp)
struct ActiveCaller {
    ActiveCaller();
    ~ActiveCaller();
    ActiveCaller(const ActiveCaller&) = delete;
    ActiveCaller& operator=(const ActiveCaller&) = delete;
};

// Where bucketFeatures and getRandomSample draw from.  Default: libc rand(), the process-wide stream the reference
// uses (matcher.cpp:297-343 via std::random_shuffle, viso.cpp:130-153).  An object switched to a PRIVATE stream
// (svh_vo_set_private_rand) draws from its own generator instead, which reproduces glibc's srand(seed) / rand()
// sequence (the additive-feedback TYPE_3 generator, x[i] = x[i-3] + x[i-31], output x >> 1): the object then sees
// exactly the numbers it would see with the process to itself, whatever other objects or threads do, and takes
// no lock (K threads in rand() contend for glibc's lock: ~4x the uncontended cost at K = 16).
struct RandStream {
    bool is_private = false;
    uint32_t r[34];
    int at = 0;   // next output is x[at + 344] of the recurrence, kept in a ring of 34
    void seed(uint32_t s) {
        // glibc srandom_r for TYPE_3: x[0] = seed (0 -> 1), x[i] = 16807 x[i-1] mod (2^31 - 1) by Schrage's
        // method, then 310 outputs are discarded
        uint32_t x[344];
        int32_t w = s == 0 ? 1 : (int32_t)s;
        x[0] = (uint32_t)w;
        for (int i = 1; i < 31; i++) {
            const int32_t hi = w / 127773, lo = w % 127773;
            w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            x[i] = (uint32_t)w;
        }
        for (int i = 31; i < 34; i++) x[i] = x[i - 31];
        for (int i = 34; i < 344; i++) x[i] = x[i - 31] + x[i - 3];
        for (int i = 0; i < 34; i++) r[i] = x[310 + i];   // ring slot j holds x[j + 310 + 34 k]
        at = 0;
        is_private = true;
    }
    int next() {
        if (!is_private) return rand();
        // ring position of x[n] is (n - 310) % 34; the new element replaces x[n - 34]
        const int i = at, i3 = at >= 3 ? at - 3 : at + 31, i31 = at >= 31 ? at - 31 : at + 3;
        const uint32_t v = r[i3] + r[i31];
        r[i] = v;
        at = at == 33 ? 0 : at + 1;
        return (int)(v >> 1);
    }
};

// bucketFeatures drawing from `rs` (matcher_engine.cpp)
int32_t bucket_features(svh_matcher* m, int32_t max_features, float bw, float bh, RandStream& rs);

// pinned host -> device copy by a kernel (bytes is a multiple of 16)
void vlaunch_upload(void* stream, const uint8_t* pinned, uint8_t* dev, size_t bytes);
void vlaunch_estimate(void* stream, const svh_p_match* pm, int N, const int32_t* samples, int iters,
                      const VoCalib& c, double* hyp_tr, int32_t* hyp_count, uint8_t* hyp_flags, double* Jg,
                      double* resg, VoResult* out, int32_t* out_inliers);

}  // namespace svh
#endif
