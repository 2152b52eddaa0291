// TEST INFRASTRUCTURE -- not part of the product.
//
// C-ABI harness around the *reference* libviso2 Matcher, compiled from the
// sources where they lie (/root/reference/libviso2/src) by oracle/Makefile into
// oracle/_ref/libref_viso.so.  It calls the reference's own functions --
// public ones, and the private stages in the order matchFeatures runs them
// (libviso2/src/matcher.cpp:209-293) -- and copies intermediates out.
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <iostream>
#include <vector>

#define private public   // reach the private stage functions
#define protected public
#include "matcher.h"
#include "viso_stereo.h"
#undef protected
#undef private
#include "filter.h"
#include "triangle.h"

#include "../include/svh.h"

static Matcher::parameters to_ref(const svh_matcher_params* p) {
    Matcher::parameters r;
    r.nms_n = p->nms_n;
    r.nms_tau = p->nms_tau;
    r.match_binsize = p->match_binsize;
    r.match_radius = p->match_radius;
    r.match_disp_tolerance = p->match_disp_tolerance;
    r.outlier_disp_tolerance = p->outlier_disp_tolerance;
    r.outlier_flow_tolerance = p->outlier_flow_tolerance;
    r.multi_stage = p->multi_stage;
    r.half_resolution = p->half_resolution;
    r.refinement = p->refinement;
    r.f = p->f;
    r.cu = p->cu;
    r.cv = p->cv;
    r.base = p->base;
    return r;
}

struct ref_matcher {
    Matcher* m;
    std::vector<Matcher::p_match> stage[SVH_M_STAGE_COUNT];
    std::vector<float> ranges;
};

static void copy_matches(const std::vector<Matcher::p_match>& v, std::vector<Matcher::p_match>& o) {
    o = v;
}

extern "C" {

// deterministic "uninitialised" memory, see oracle/ref_elas_harness.cpp
static int g_zero_alloc = 0;
void ref_init(int deterministic) { g_zero_alloc = deterministic; }
void* __libc_malloc(size_t);
void* __libc_memalign(size_t, size_t);
void* malloc(size_t n) {
    void* p = __libc_malloc(n);
    if (p && g_zero_alloc) memset(p, 0, n);
    return p;
}
int posix_memalign(void** out, size_t align, size_t n) {
    void* p = __libc_memalign(align, n);
    if (!p) return 12;
    if (g_zero_alloc) memset(p, 0, n);
    *out = p;
    return 0;
}

void ref_matcher_params_default(svh_matcher_params* p) {
    Matcher::parameters r;
    p->nms_n = r.nms_n;
    p->nms_tau = r.nms_tau;
    p->match_binsize = r.match_binsize;
    p->match_radius = r.match_radius;
    p->match_disp_tolerance = r.match_disp_tolerance;
    p->outlier_disp_tolerance = r.outlier_disp_tolerance;
    p->outlier_flow_tolerance = r.outlier_flow_tolerance;
    p->multi_stage = r.multi_stage;
    p->half_resolution = r.half_resolution;
    p->refinement = r.refinement;
    p->f = p->cu = p->cv = p->base = 0;   // the reference leaves these unset
}

ref_matcher* ref_matcher_create(const svh_matcher_params* p) {
    ref_matcher* h = new ref_matcher();
    h->m = new Matcher(to_ref(p));
    return h;
}
void ref_matcher_destroy(ref_matcher* h) {
    delete h->m;
    delete h;
}
void ref_matcher_set_intrinsics(ref_matcher* h, double f, double cu, double cv, double base) {
    h->m->setIntrinsics(f, cu, cv, base);
}
void ref_matcher_push_back(ref_matcher* h, const uint8_t* I1, const uint8_t* I2, const int32_t* dims,
                           int32_t replace) {
    int32_t d[3] = {dims[0], dims[1], dims[2]};
    h->m->pushBack(const_cast<uint8_t*>(I1), const_cast<uint8_t*>(I2), d, replace != 0);
}

static Matrix* make_tr(const double* Tr, Matrix& store) {
    if (!Tr) return 0;
    store = Matrix(4, 4, Tr);
    return &store;
}

// the reference's public entry point, untouched
void ref_matcher_match(ref_matcher* h, int32_t method, const double* Tr) {
    Matrix T;
    h->m->matchFeatures(method, make_tr(Tr, T));
    h->stage[SVH_M_SPARSE] = h->m->_p_matched_1;
    h->stage[SVH_M_DENSE] = h->m->_p_matched_2;
}

// matchFeatures (matcher.cpp:261-282, multi_stage branch) stage by stage
void ref_matcher_match_staged(ref_matcher* h, int32_t method, const double* Tr) {
    Matcher* m = h->m;
    Matrix T;
    Matrix* Trp = make_tr(Tr, T);
    for (int s = 0; s < SVH_M_STAGE_COUNT; s++) h->stage[s].clear();
    h->ranges.clear();
    m->_p_matched_1.clear();
    m->_p_matched_2.clear();
    if (m->_param.multi_stage) {
        m->matching(m->_m1p1, m->_m2p1, m->_m1c1, m->_m2c1, m->_n1p1, m->_n2p1, m->_n1c1, m->_n2c1,
                    m->_p_matched_1, method, false, Trp);
        h->stage[SVH_M_SPARSE_RAW] = m->_p_matched_1;
        m->removeOutliers(m->_p_matched_1, method);
        h->stage[SVH_M_SPARSE] = m->_p_matched_1;
        m->computePriorStatistics(m->_p_matched_1, method);
        for (size_t i = 0; i < m->_ranges.size(); i++) {
            const Matcher::range& r = m->_ranges[i];
            for (int k = 0; k < 4; k++) h->ranges.push_back(r.u_min[k]);
            for (int k = 0; k < 4; k++) h->ranges.push_back(r.u_max[k]);
            for (int k = 0; k < 4; k++) h->ranges.push_back(r.v_min[k]);
            for (int k = 0; k < 4; k++) h->ranges.push_back(r.v_max[k]);
        }
        m->matching(m->_m1p2, m->_m2p2, m->_m1c2, m->_m2c2, m->_n1p2, m->_n2p2, m->_n1c2, m->_n2c2,
                    m->_p_matched_2, method, true, Trp);
    } else {
        m->matching(m->_m1p2, m->_m2p2, m->_m1c2, m->_m2c2, m->_n1p2, m->_n2p2, m->_n1c2, m->_n2c2,
                    m->_p_matched_2, method, false, Trp);
    }
    h->stage[SVH_M_DENSE_RAW] = m->_p_matched_2;
    if (m->_param.refinement > 0) m->refinement(m->_p_matched_2, method);
    h->stage[SVH_M_DENSE_REFINED] = m->_p_matched_2;
    m->removeOutliers(m->_p_matched_2, method);
    h->stage[SVH_M_DENSE] = m->_p_matched_2;
}

int64_t ref_matcher_get_stage(ref_matcher* h, int32_t stage, void* buf, int64_t cap) {
    if (stage == SVH_M_RANGES) {
        int64_t n = (int64_t)(h->ranges.size() * sizeof(float));
        if (buf && cap >= n && n) memcpy(buf, h->ranges.data(), n);
        return n;
    }
    if (stage < 0 || stage >= SVH_M_STAGE_COUNT) return -1;
    const std::vector<Matcher::p_match>& v = h->stage[stage];
    int64_t n = (int64_t)(v.size() * sizeof(svh_p_match));
    if (buf && cap >= n) {
        svh_p_match* o = (svh_p_match*)buf;
        for (size_t i = 0; i < v.size(); i++) {
            o[i].u1p = v[i].u1p; o[i].v1p = v[i].v1p; o[i].i1p = v[i].i1p;
            o[i].u2p = v[i].u2p; o[i].v2p = v[i].v2p; o[i].i2p = v[i].i2p;
            o[i].u1c = v[i].u1c; o[i].v1c = v[i].v1c; o[i].i1c = v[i].i1c;
            o[i].u2c = v[i].u2c; o[i].v2c = v[i].v2c; o[i].i2c = v[i].i2c;
        }
    }
    return n;
}

int32_t ref_matcher_get_matches(ref_matcher* h, svh_p_match* out, int32_t cap) {
    std::vector<Matcher::p_match> v = h->m->getMatches();
    for (int32_t i = 0; i < (int32_t)v.size() && i < cap; i++) {
        out[i].u1p = v[i].u1p; out[i].v1p = v[i].v1p; out[i].i1p = v[i].i1p;
        out[i].u2p = v[i].u2p; out[i].v2p = v[i].v2p; out[i].i2p = v[i].i2p;
        out[i].u1c = v[i].u1c; out[i].v1c = v[i].v1c; out[i].i1c = v[i].i1c;
        out[i].u2c = v[i].u2c; out[i].v2c = v[i].v2c; out[i].i2c = v[i].i2c;
    }
    return (int32_t)v.size();
}

int32_t ref_matcher_get_features(ref_matcher* h, int32_t table, int32_t* out, int32_t cap) {
    Matcher* m = h->m;
    int32_t* t[8] = {m->_m1p1, m->_m1p2, m->_m2p1, m->_m2p2, m->_m1c1, m->_m1c2, m->_m2c1, m->_m2c2};
    int32_t n[8] = {m->_n1p1, m->_n1p2, m->_n2p1, m->_n2p2, m->_n1c1, m->_n1c2, m->_n2c1, m->_n2c2};
    if (table < 0 || table > 7) return -1;
    if (out && t[table]) memcpy(out, t[table], sizeof(int32_t) * 12 * std::min(n[table], cap));
    return t[table] ? n[table] : 0;
}

// filter images of the current left frame (see svh_matcher_get_filter)
int64_t ref_matcher_get_filter(ref_matcher* h, int32_t which, void* buf, int64_t cap, int32_t* dims3) {
    Matcher* m = h->m;
    int32_t full[3] = {m->_dims_c[0], m->_dims_c[1], m->_dims_c[2]};
    int32_t mt[3] = {full[0], full[1], full[2]};
    if (m->_param.half_resolution) m->getHalfResolutionDimensions(full, mt);
    const uint8_t* src = 0;
    const int32_t* d = mt;
    switch (which) {
        case 0: src = m->_I1c_du; break;
        case 1: src = m->_I1c_dv; break;
        case 2: src = m->_I1c_du_full; d = full; break;
        case 3: src = m->_I1c_dv_full; d = full; break;
        default: break;
    }
    if (which == 4 || which == 5) {
        // the reference frees f1/f2 inside computeFeatures: recompute them the same way
        uint8_t* I = m->_I1c;
        uint8_t* Ih = m->_param.half_resolution ? m->createHalfResolutionImage(I, full) : I;
        int64_t n = (int64_t)mt[2] * mt[1] * 2;
        if (dims3) memcpy(dims3, mt, 12);
        if (buf && cap >= n) {
            int16_t* tmp = (int16_t*)_mm_malloc(n, 16);
            if (which == 4) filter::blob5x5(Ih, tmp, mt[2], mt[1]);
            else            filter::checkerboard5x5(Ih, tmp, mt[2], mt[1]);
            memcpy(buf, tmp, n);
            _mm_free(tmp);
        }
        if (Ih != I) _mm_free(Ih);
        return n;
    }
    if (!src) return -1;
    int64_t n = (int64_t)d[2] * d[1];
    if (dims3) memcpy(dims3, d, 12);
    if (buf && cap >= n) memcpy(buf, src, n);
    return n;
}

int32_t ref_matcher_bucket(ref_matcher* h, int32_t max_features, float bw, float bh) {
    h->m->bucketFeatures(max_features, bw, bh);
    return (int32_t)h->m->_p_matched_2.size();
}

float ref_matcher_gain(ref_matcher* h, const int32_t* inl, int32_t n) {
    return h->m->getGain(std::vector<int32_t>(inl, inl + n));
}

// Matcher::nonMaximumSuppression on caller-supplied filter images
int32_t ref_matcher_nms(const svh_matcher_params* p, const int16_t* f1, const int16_t* f2,
                        const int32_t* dims, int32_t nms_n, int32_t* out, int32_t cap) {
    Matcher m(to_ref(p));
    std::vector<Matcher::maximum> mx;
    m.nonMaximumSuppression(const_cast<int16_t*>(f1), const_cast<int16_t*>(f2), dims, mx, nms_n);
    for (int32_t i = 0; i < (int32_t)mx.size() && i < cap; i++) {
        out[4 * i + 0] = mx[i].u;
        out[4 * i + 1] = mx[i].v;
        out[4 * i + 2] = mx[i].val;
        out[4 * i + 3] = mx[i].c;
    }
    return (int32_t)mx.size();
}

void ref_sobel5x5(const uint8_t* in, uint8_t* du, uint8_t* dv, int32_t bpl, int32_t h) {
    filter::sobel5x5(in, du, dv, bpl, h);
}
void ref_blob5x5(const uint8_t* in, int16_t* out, int32_t bpl, int32_t h) { filter::blob5x5(in, out, bpl, h); }
void ref_checkerboard5x5(const uint8_t* in, int16_t* out, int32_t bpl, int32_t h) {
    filter::checkerboard5x5(in, out, bpl, h);
}

// libviso2's copy of Triangle (3 trivial diffs vs libelas')
int32_t ref_viso_triangulate(const float* pts, int32_t n, int32_t* tri, int32_t cap) {
    struct triangulateio in, out;
    memset(&in, 0, sizeof(in));
    memset(&out, 0, sizeof(out));
    in.numberofpoints = n;
    in.pointlist = (float*)malloc(sizeof(float) * 2 * n);
    memcpy(in.pointlist, pts, sizeof(float) * 2 * n);
    char sw[] = "zQB";
    triangulate(sw, &in, &out, NULL);
    int32_t nt = out.numberoftriangles;
    for (int32_t i = 0; i < nt && i < cap; i++)
        for (int k = 0; k < 3; k++) tri[3 * i + k] = out.trianglelist[3 * i + k];
    free(in.pointlist);
    free(out.pointlist);
    free(out.trianglelist);
    return nt;
}


// ---- VisualOdometryStereo (viso_stereo.cpp, viso.cpp) ------------------------------------
struct ref_vo { VisualOdometryStereo* vo; };

static VisualOdometryStereo::parameters to_ref_vo(const svh_vo_params* p) {
    VisualOdometryStereo::parameters r;
    r.match = to_ref(&p->match);
    r.bucket.max_features = p->bucket_max_features;
    r.bucket.bucket_width = p->bucket_width;
    r.bucket.bucket_height = p->bucket_height;
    r.calib.f = p->f; r.calib.cu = p->cu; r.calib.cv = p->cv;
    r.base = p->base;
    r.ransac_iters = p->ransac_iters;
    r.inlier_threshold = p->inlier_threshold;
    r.reweighting = p->reweighting != 0;
    return r;
}
void ref_vo_params_default(svh_vo_params* p) {
    VisualOdometryStereo::parameters r;
    ref_matcher_params_default(&p->match);
    p->bucket_max_features = r.bucket.max_features;
    p->bucket_width = r.bucket.bucket_width;
    p->bucket_height = r.bucket.bucket_height;
    p->f = r.calib.f; p->cu = r.calib.cu; p->cv = r.calib.cv;
    p->base = r.base;
    p->ransac_iters = r.ransac_iters;
    p->inlier_threshold = r.inlier_threshold;
    p->reweighting = r.reweighting ? 1 : 0;
}
ref_vo* ref_vo_create(const svh_vo_params* p) {
    ref_vo* h = new ref_vo();
    h->vo = new VisualOdometryStereo(to_ref_vo(p));   // calls srand(0)
    return h;
}
void ref_vo_destroy(ref_vo* h) { delete h->vo; delete h; }
int32_t ref_vo_process(ref_vo* h, const uint8_t* I1, const uint8_t* I2, const int32_t* dims, int32_t replace) {
    int32_t d[3] = {dims[0], dims[1], dims[2]};
    return h->vo->process(const_cast<uint8_t*>(I1), const_cast<uint8_t*>(I2), d, replace != 0) ? 1 : 0;
}
static std::vector<Matcher::p_match> from_abi(const svh_p_match* m, int32_t n) {
    std::vector<Matcher::p_match> v(n);
    for (int32_t i = 0; i < n; i++) {
        v[i].u1p = m[i].u1p; v[i].v1p = m[i].v1p; v[i].i1p = m[i].i1p;
        v[i].u2p = m[i].u2p; v[i].v2p = m[i].v2p; v[i].i2p = m[i].i2p;
        v[i].u1c = m[i].u1c; v[i].v1c = m[i].v1c; v[i].i1c = m[i].i1c;
        v[i].u2c = m[i].u2c; v[i].v2c = m[i].v2c; v[i].i2c = m[i].i2c;
    }
    return v;
}
int32_t ref_vo_estimate_motion(ref_vo* h, const svh_p_match* m, int32_t n, double* tr6) {
    std::vector<double> tr = h->vo->estimateMotion(from_abi(m, n));
    if (tr.size() != 6) return 0;
    for (int i = 0; i < 6; i++) tr6[i] = tr[i];
    return 1;
}
int32_t ref_vo_process_matches(ref_vo* h, const svh_p_match* m, int32_t n) {
    return h->vo->VisualOdometry::process(from_abi(m, n)) ? 1 : 0;
}
void ref_vo_get_motion(ref_vo* h, double* Tr16) {
    Matrix T = h->vo->getDeltaMotion();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) Tr16[4 * i + j] = T._val[i][j];
}
int32_t ref_vo_get_inliers(ref_vo* h, int32_t* out, int32_t cap) {
    std::vector<int32_t> v = h->vo->getInlierIndices();
    for (int32_t i = 0; i < (int32_t)v.size() && i < cap; i++) out[i] = v[i];
    return (int32_t)v.size();
}
int32_t ref_vo_num_matches(ref_vo* h) { return h->vo->getNumberOfMatches(); }
int32_t ref_vo_get_matches(ref_vo* h, svh_p_match* out, int32_t cap) {
    ref_matcher tmp;
    tmp.m = h->vo->_matcher;
    return ref_matcher_get_matches(&tmp, out, cap);
}
float ref_vo_get_gain(ref_vo* h, const int32_t* inl, int32_t n) {
    return h->vo->getGain(std::vector<int32_t>(inl, inl + n));
}
void ref_srand(uint32_t s) { srand(s); }


// ---- coefficient matrices of stereomapper's map fusion, computed with the reference's own Matrix
// (stereothread.cpp:196-199, 306-314, 444-455): pins oracle/map_oracle.cpp's orc_map_coeffs
void ref_map_coeffs(const double* H16, float f, float cu, float cv, float* hcf12, float* hfc4, float* pfc12) {
    Matrix Ht(4, 4);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) Ht._val[i][j] = H16[4 * i + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) hcf12[4 * i + j] = Ht._val[i][j];
    Matrix H = Matrix::inv(Ht);
    for (int j = 0; j < 4; j++) hfc4[j] = H._val[2][j];
    Matrix K(3, 3);
    K._val[0][0] = f;
    K._val[1][1] = f;
    K._val[0][2] = cu;
    K._val[1][2] = cv;
    K._val[2][2] = 1;
    Matrix P = K * H.getMat(0, 0, 2, 3);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) pfc12[4 * i + j] = P._val[i][j];
}

}  // extern "C"
