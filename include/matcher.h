/*
 * matcher.h -- source-compatible stand-in for libviso2/src/matcher.h.
 *
 * Same class name, nested `parameters` (libviso2/src/matcher.h:41-69) and
 * `p_match` (:87-102), same public methods:
 *     Matcher(parameters), ~Matcher(), setIntrinsics, pushBack (stereo and
 *     single-image), matchFeatures(method, Matrix* Tr_delta = 0),
 *     bucketFeatures, getMatches (by value), getGain          (:72-154)
 * so callers such as VisualOdometryStereo::process (libviso2/src/
 * viso_stereo.cpp:41-68) and stereomapper's view2d.cpp:43 compile unchanged and
 * run on the MI355X through libsvhip.so (C-ABI in svh.h).  The private half of
 * the reference class is replaced by an opaque handle.
 *
 * Behaviour kept: bad dimensions print "ERROR: Image dimension mismatch!" to
 * stderr and return (matcher.cpp:110-114); matchFeatures returns silently when
 * a needed feature table is missing (matcher.cpp:216-259).  Device failures are
 * reported on stderr; there is no CPU fallback.
 */
#ifndef __MATCHER_H__
#define __MATCHER_H__

#include <stdint.h>

#include <iostream>
#include <vector>

#include "matrix.h"
#include "svh.h"

class Matcher {
public:
    struct parameters {
        int32_t nms_n;
        int32_t nms_tau;
        int32_t match_binsize;
        int32_t match_radius;
        int32_t match_disp_tolerance;
        int32_t outlier_disp_tolerance;
        int32_t outlier_flow_tolerance;
        int32_t multi_stage;
        int32_t half_resolution;
        int32_t refinement;
        double  f, cu, cv, base;
        parameters() {
            svh_matcher_params q;
            svh_matcher_params_default(&q);
            nms_n = q.nms_n;
            nms_tau = q.nms_tau;
            match_binsize = q.match_binsize;
            match_radius = q.match_radius;
            match_disp_tolerance = q.match_disp_tolerance;
            outlier_disp_tolerance = q.outlier_disp_tolerance;
            outlier_flow_tolerance = q.outlier_flow_tolerance;
            multi_stage = q.multi_stage;
            half_resolution = q.half_resolution;
            refinement = q.refinement;
            f = cu = cv = base = 0;
        }
    };

    // field order and types as in the reference: layout-compatible with svh_p_match
    struct p_match {
        float   u1p, v1p; int32_t i1p;
        float   u2p, v2p; int32_t i2p;
        float   u1c, v1c; int32_t i1c;
        float   u2c, v2c; int32_t i2c;
        p_match() {}
        p_match(float u1p, float v1p, int32_t i1p, float u2p, float v2p, int32_t i2p, float u1c, float v1c,
                int32_t i1c, float u2c, float v2c, int32_t i2c)
            : u1p(u1p), v1p(v1p), i1p(i1p), u2p(u2p), v2p(v2p), i2p(i2p), u1c(u1c), v1c(v1c), i1c(i1c),
              u2c(u2c), v2c(v2c), i2c(i2c) {}
    };

    Matcher(parameters param) {
        svh_matcher_params q;
        q.nms_n = param.nms_n;
        q.nms_tau = param.nms_tau;
        q.match_binsize = param.match_binsize;
        q.match_radius = param.match_radius;
        q.match_disp_tolerance = param.match_disp_tolerance;
        q.outlier_disp_tolerance = param.outlier_disp_tolerance;
        q.outlier_flow_tolerance = param.outlier_flow_tolerance;
        q.multi_stage = param.multi_stage;
        q.half_resolution = param.half_resolution;
        q.refinement = param.refinement;
        q.f = param.f;
        q.cu = param.cu;
        q.cv = param.cv;
        q.base = param.base;
        _h = svh_matcher_create(&q);
    }
    ~Matcher() { svh_matcher_destroy(_h); }

    void setIntrinsics(double f, double cu, double cv, double base) {
        svh_matcher_set_intrinsics(_h, f, cu, cv, base);
    }

    void pushBack(uint8_t* I1, uint8_t* I2, int32_t* dims, const bool replace) {
        report(svh_matcher_push_back(_h, I1, I2, dims, replace ? 1 : 0), "pushBack");
    }
    void pushBack(uint8_t* I1, int32_t* dims, const bool replace) { pushBack(I1, 0, dims, replace); }

    // method: 0 = flow, 1 = stereo, 2 = quad matching
    void matchFeatures(int32_t method, Matrix* Tr_delta = 0) {
        double tr[16];
        if (Tr_delta)
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) tr[4 * i + j] = Tr_delta->_val[i][j];
        report(svh_matcher_match_features(_h, method, Tr_delta ? tr : 0), "matchFeatures");
    }

    void bucketFeatures(int32_t max_features, float bucket_width, float bucket_height) {
        svh_matcher_bucket_features(_h, max_features, bucket_width, bucket_height);
    }

    std::vector<Matcher::p_match> getMatches() {
        const int32_t n = svh_matcher_get_matches(_h, 0, 0);
        std::vector<Matcher::p_match> out(n > 0 ? n : 0);
        if (n > 0) svh_matcher_get_matches(_h, reinterpret_cast<svh_p_match*>(&out[0]), n);
        return out;
    }

    float getGain(std::vector<int32_t> inliers) {
        return svh_matcher_get_gain(_h, inliers.empty() ? 0 : &inliers[0], (int32_t)inliers.size());
    }

private:
    Matcher(const Matcher&);
    Matcher& operator=(const Matcher&);
    static void report(int32_t rc, const char* what) {
        if (rc < 0 && rc != SVH_ERR_BAD_DIMS)   // (bad dimensions: the library has printed the reference's message)
            std::cerr << "ERROR: Matcher::" << what << " failed on the device (" << rc
                      << "): " << svh_last_error() << std::endl;
    }
    svh_matcher* _h;
};

#endif
