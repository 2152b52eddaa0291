/*
 * elas.h -- source-compatible stand-in for the reference's libelas/src/elas.h.
 *
 * Same class name, nested `setting` / `parameters` (same field names and
 * defaults, libelas/src/elas.h:56-148) and the same
 *     Elas(parameters), ~Elas(), process(I1,I2,D1,D2,dims)
 * (libelas/src/elas.h:151-165), so callers such as
 * stereomapper/stereothread.cpp:76-114 and libelas/src/main.cpp:55-64 compile
 * unchanged and run on the MI355X through libsvhip.so (C-ABI in svh.h).  The
 * private half of the reference class (SSE types, stage functions) is replaced
 * by an opaque handle.
 *
 * Behaviour kept from the reference: inputs are consumed before process()
 * returns; D1/D2 are caller-allocated, tightly packed; with fewer than three
 * support points a message goes to stdout and D1/D2 stay untouched
 * (elas.cpp:69-75).  A HIP failure prints to stderr and likewise leaves the
 * outputs untouched; there is no CPU fallback.
 */
#ifndef __ELAS_H__
#define __ELAS_H__

#include <stdint.h>
#include <iostream>

#include "svh.h"

class Elas {
public:
    enum setting { ROBOTICS, MIDDLEBURY };

    struct parameters {
        int32_t disp_min;
        int32_t disp_max;
        float   support_threshold;
        int32_t support_texture;
        int32_t candidate_stepsize;
        int32_t incon_window_size;
        int32_t incon_threshold;
        int32_t incon_min_support;
        bool    add_corners;
        int32_t grid_size;
        float   beta;
        float   gamma;
        float   sigma;
        float   sradius;
        int32_t match_texture;
        int32_t lr_threshold;
        float   speckle_sim_threshold;
        int32_t speckle_size;
        int32_t ipol_gap_width;
        bool    filter_median;
        bool    filter_adaptive_mean;
        bool    postprocess_only_left;
        bool    subsampling;   // D1/D2 are then width/2 x height/2 (rounded towards zero)

        // the two presets live in the library (svh_elas_params_default)
        parameters(setting s = ROBOTICS) {
            svh_elas_params q;
            svh_elas_params_default(&q, s == ROBOTICS ? SVH_ELAS_ROBOTICS : SVH_ELAS_MIDDLEBURY);
            disp_min = q.disp_min;
            disp_max = q.disp_max;
            support_threshold = q.support_threshold;
            support_texture = q.support_texture;
            candidate_stepsize = q.candidate_stepsize;
            incon_window_size = q.incon_window_size;
            incon_threshold = q.incon_threshold;
            incon_min_support = q.incon_min_support;
            add_corners = q.add_corners != 0;
            grid_size = q.grid_size;
            beta = q.beta;
            gamma = q.gamma;
            sigma = q.sigma;
            sradius = q.sradius;
            match_texture = q.match_texture;
            lr_threshold = q.lr_threshold;
            speckle_sim_threshold = q.speckle_sim_threshold;
            speckle_size = q.speckle_size;
            ipol_gap_width = q.ipol_gap_width;
            filter_median = q.filter_median != 0;
            filter_adaptive_mean = q.filter_adaptive_mean != 0;
            postprocess_only_left = q.postprocess_only_left != 0;
            subsampling = q.subsampling != 0;
        }
    };

    Elas(parameters param) : _param(param), _h(0) {}
    ~Elas() {
        if (_h) svh_elas_destroy(_h);
    }

    // dims[0] = width, dims[1] = height, dims[2] = bytes per line of I1 and I2
    void process(uint8_t* I1, uint8_t* I2, float* D1, float* D2, const int32_t* dims) {
        svh_elas_params q;
        q.disp_min = _param.disp_min;
        q.disp_max = _param.disp_max;
        q.support_threshold = _param.support_threshold;
        q.support_texture = _param.support_texture;
        q.candidate_stepsize = _param.candidate_stepsize;
        q.incon_window_size = _param.incon_window_size;
        q.incon_threshold = _param.incon_threshold;
        q.incon_min_support = _param.incon_min_support;
        q.add_corners = _param.add_corners;
        q.grid_size = _param.grid_size;
        q.beta = _param.beta;
        q.gamma = _param.gamma;
        q.sigma = _param.sigma;
        q.sradius = _param.sradius;
        q.match_texture = _param.match_texture;
        q.lr_threshold = _param.lr_threshold;
        q.speckle_sim_threshold = _param.speckle_sim_threshold;
        q.speckle_size = _param.speckle_size;
        q.ipol_gap_width = _param.ipol_gap_width;
        q.filter_median = _param.filter_median;
        q.filter_adaptive_mean = _param.filter_adaptive_mean;
        q.postprocess_only_left = _param.postprocess_only_left;
        q.subsampling = _param.subsampling;
        if (_h) svh_elas_destroy(_h);
        _h = svh_elas_create(&q);
        int32_t rc = _h ? svh_elas_process(_h, I1, I2, D1, D2, dims) : SVH_ERR_BAD_ARG;
        if (rc < 0)
            std::cerr << "ERROR: Elas::process failed on the device (" << rc
                      << "): " << svh_last_error() << std::endl;
    }

private:
    Elas(const Elas&);
    Elas& operator=(const Elas&);
    parameters _param;
    svh_elas*  _h;
};

#endif
