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
