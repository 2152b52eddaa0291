/*
 * viso_stereo.h -- source-compatible stand-in for libviso2/src/viso_stereo.h.
 *
 * class VisualOdometryStereo : public VisualOdometry with the reference's nested
 * `parameters` (:30-44: base, ransac_iters, inlier_threshold, reweighting on top of
 * VisualOdometry::parameters), constructor and
 *     bool process(uint8_t* I1, uint8_t* I2, int32_t* dims, bool replace = false)
 * so stereomapper/visualodometrythread.cpp:19-49, 92-118 compiles unchanged and runs
 * on the MI355X: feature matching (svh_matcher_*) and the RANSAC + Gauss-Newton motion
 * estimate (viso_stereo.cpp:72-228) both execute on the device; libc rand() is
 * consumed exactly like the reference does (srand(0) in the constructor, viso.cpp:36).
 */
#ifndef VISO_STEREO_H
#define VISO_STEREO_H

#include "viso.h"

class VisualOdometryStereo : public VisualOdometry {
public:
    struct parameters : public VisualOdometry::parameters {
        double  base;
        int32_t ransac_iters;
        double  inlier_threshold;
        bool    reweighting;
        parameters() {
            base = 1.0;
            ransac_iters = 200;
            inlier_threshold = 2.0;
            reweighting = true;
        }
    };

    VisualOdometryStereo(parameters param) : VisualOdometry(to_abi(param)), _param(param) {}
    virtual ~VisualOdometryStereo() {}

    bool process(uint8_t* I1, uint8_t* I2, int32_t* dims, bool replace = false) {
        return svh_vo_process(_vo, I1, I2, dims, replace ? 1 : 0) == 1;
    }
    using VisualOdometry::process;

    // ---- extensions (not in the reference): K objects, one frame each, as ONE call (svh_vo_process_batch) ----
    // ok[i] (optional) = what process() of object i would have returned; returns the number of updated motions,
    // negative on an error.  Per object the results are those of K process() calls in this order.
    static int32_t processBatch(VisualOdometryStereo* const* vos, int32_t K, uint8_t* const* I1, uint8_t* const* I2,
                                int32_t* dims, bool replace = false, int32_t* ok = 0) {
        std::vector<svh_vo*> h((size_t)K);
        for (int32_t i = 0; i < K; i++) h[i] = vos[i]->_vo;
        return svh_vo_process_batch(h.data(), K, I1, I2, dims, replace ? 1 : 0, ok);
    }
    // the pipelined loop: prefetchBatch hands over the first frame, every processNextBatch processes the frame
    // handed over before and hands over the next one (nextI1 = nextI2 = 0 after the last frame); the images of a
    // handed-over frame stay untouched until it has been processed
    static int32_t prefetchBatch(VisualOdometryStereo* const* vos, int32_t K, uint8_t* const* I1, uint8_t* const* I2,
                                 int32_t* dims) {
        std::vector<svh_vo*> h((size_t)K);
        for (int32_t i = 0; i < K; i++) h[i] = vos[i]->_vo;
        return svh_vo_prefetch_batch(h.data(), K, I1, I2, dims);
    }
    static int32_t processNextBatch(VisualOdometryStereo* const* vos, int32_t K, uint8_t* const* nextI1,
                                    uint8_t* const* nextI2, int32_t* dims, bool replace = false, int32_t* ok = 0) {
        std::vector<svh_vo*> h((size_t)K);
        for (int32_t i = 0; i < K; i++) h[i] = vos[i]->_vo;
        return svh_vo_process_next_batch(h.data(), K, nextI1, nextI2, dims, replace ? 1 : 0, ok);
    }
    // bucketing / RANSAC samples from a private generator with glibc's srand(seed) sequence instead of the
    // process-wide rand() (seed 0 = what the reference's constructor seeds)
    void usePrivateRand(uint32_t seed = 0) { svh_vo_set_private_rand(_vo, 1, seed); }

private:
    static svh_vo_params to_abi(const parameters& p) {
        svh_vo_params q;
        svh_vo_params_default(&q);
        q.match.nms_n = p.match.nms_n;
        q.match.nms_tau = p.match.nms_tau;
        q.match.match_binsize = p.match.match_binsize;
        q.match.match_radius = p.match.match_radius;
        q.match.match_disp_tolerance = p.match.match_disp_tolerance;
        q.match.outlier_disp_tolerance = p.match.outlier_disp_tolerance;
        q.match.outlier_flow_tolerance = p.match.outlier_flow_tolerance;
        q.match.multi_stage = p.match.multi_stage;
        q.match.half_resolution = p.match.half_resolution;
        q.match.refinement = p.match.refinement;
        q.match.f = p.match.f; q.match.cu = p.match.cu; q.match.cv = p.match.cv; q.match.base = p.match.base;
        q.bucket_max_features = p.bucket.max_features;
        q.bucket_width = p.bucket.bucket_width;
        q.bucket_height = p.bucket.bucket_height;
        q.f = p.calib.f; q.cu = p.calib.cu; q.cv = p.calib.cv;
        q.base = p.base;
        q.ransac_iters = p.ransac_iters;
        q.inlier_threshold = p.inlier_threshold;
        q.reweighting = p.reweighting ? 1 : 0;
        return q;
    }
    parameters _param;
};

#endif  // VISO_STEREO_H
