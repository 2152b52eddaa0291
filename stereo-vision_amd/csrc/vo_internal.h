// internal interface between vo_engine.cpp (host) and vo_kernels.hip (device)
#ifndef SVH_VO_INTERNAL_H
#define SVH_VO_INTERNAL_H
#include <stdint.h>

#include "../../include/svh.h"

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

// pinned host -> device copy by a kernel (bytes is a multiple of 16)
void vlaunch_upload(void* stream, const uint8_t* pinned, uint8_t* dev, size_t bytes);
void vlaunch_estimate(void* stream, const svh_p_match* pm, int N, const int32_t* samples, int iters,
                      const VoCalib& c, double* hyp_tr, int32_t* hyp_count, uint8_t* hyp_flags, double* Jg,
                      double* resg, VoResult* out, int32_t* out_inliers);

}  // namespace svh
#endif
