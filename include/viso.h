/*
 * viso.h -- source-compatible stand-in for libviso2/src/viso.h (class VisualOdometry).
 *
 * Same nested types (calibration :30-44, bucketing :46-58, parameters :60-65) and the
 * public surface callers use (:87-130): process(p_matched), getDeltaMotion,
 * calculateRollPitchYaw / Velocity / AltitudeFromTransformation (viso.cpp:157-183),
 * getMatches, getNumberOfMatches, getNumberOfInliers, getInlierIndices, getGain and
 * operator<<.  The reference class is abstract (estimateMotion is pure virtual) and only
 * its stereo subclass is on this repository's path, so the state lives in a svh_vo handle
 * created by VisualOdometryStereo (viso_stereo.h); see svh.h for the C-ABI.
 */
#ifndef VISO_H
#define VISO_H

#include <math.h>
#include <stdint.h>

#include <iostream>
#include <vector>

#include "matcher.h"
#include "matrix.h"
#include "svh.h"

class VisualOdometry {
public:
    struct calibration {
        double f, cu, cv;
        calibration() { f = 1; cu = 0; cv = 0; }
    };
    struct bucketing {
        int32_t max_features;
        double  bucket_width, bucket_height;
        bucketing() { max_features = 2; bucket_width = 50; bucket_height = 50; }
    };
    struct parameters {
        Matcher::parameters match;
        bucketing           bucket;
        calibration         calib;
    };

    virtual ~VisualOdometry() { svh_vo_destroy(_vo); }

    // motion from given matches (viso.h:87-91)
    bool process(std::vector<Matcher::p_match> p_matched) {
        return svh_vo_process_matches(_vo, reinterpret_cast<const svh_p_match*>(p_matched.data()),
                                      (int32_t)p_matched.size()) == 1;
    }

    Matrix getDeltaMotion() const {
        double T[16];
        svh_vo_get_motion(_vo, T);
        return Matrix(4, 4, T);
    }

    // viso.cpp:157-183
    void calculateRollPitchYawFromTransformation(double& roll, double& pitch, double& yaw) const {
        double T[16];
        svh_vo_get_motion(_vo, T);
        roll = atan2(-T[1], T[0]);
        pitch = atan2(-T[6], T[10]);
        yaw = atan2(T[2], sqrt(T[0] * T[0] + T[1] * T[1]));
    }
    void calculateVelocityFromTransformation(double& velocity) const {
        double T[16];
        svh_vo_get_motion(_vo, T);
        velocity = sqrt(T[3] * T[3] + T[11] * T[11]);
    }
    void calculateAltitudeFromTransformation(double& altitude) const {
        double T[16];
        svh_vo_get_motion(_vo, T);
        altitude = T[7];
    }

    std::vector<Matcher::p_match> getMatches() {
        std::vector<Matcher::p_match> v((size_t)svh_vo_get_matches(_vo, 0, 0));
        if (!v.empty()) svh_vo_get_matches(_vo, reinterpret_cast<svh_p_match*>(v.data()), (int32_t)v.size());
        return v;
    }
    int32_t getNumberOfMatches() { return svh_vo_num_matches(_vo); }
    int32_t getNumberOfInliers() { return svh_vo_get_inliers(_vo, 0, 0); }
    std::vector<int32_t> getInlierIndices() {
        std::vector<int32_t> v((size_t)svh_vo_get_inliers(_vo, 0, 0));
        if (!v.empty()) svh_vo_get_inliers(_vo, v.data(), (int32_t)v.size());
        return v;
    }
    float getGain(std::vector<int32_t> inliers) {
        return svh_vo_get_gain(_vo, inliers.data(), (int32_t)inliers.size());
    }

    friend std::ostream& operator<<(std::ostream& os, VisualOdometry& viso) {
        Matrix p = viso.getDeltaMotion();
        os << p._val[0][0] << " " << p._val[0][1] << " " << p._val[0][2] << " " << p._val[0][3] << " ";
        os << p._val[1][0] << " " << p._val[1][1] << " " << p._val[1][2] << " " << p._val[1][3] << " ";
        os << p._val[2][0] << " " << p._val[2][1] << " " << p._val[2][2] << " " << p._val[2][3];
        return os;
    }

protected:
    explicit VisualOdometry(const svh_vo_params& q) : _vo(svh_vo_create(&q)) {}
    svh_vo* _vo;

private:
    VisualOdometry(const VisualOdometry&);
    VisualOdometry& operator=(const VisualOdometry&);
};

#endif  // VISO_H
