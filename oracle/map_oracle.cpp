// TEST INFRASTRUCTURE -- CPU restatement of stereomapper's map fusion:
//   StereoThread::createCurrentMap                  stereomapper/stereothread.cpp:180-255
//   StereoThread::addDisparityMapToReconstruction   stereomapper/stereothread.cpp:290-437
// Only tests/ may load this (through liboracle.so); the product never does.
//
// PARITY UNPINNED for the per-pixel part.  The reference code lives inside Qt classes that cannot
// be compiled here, and it is not well defined as written: `_previous_map3d = current_map3d;
// releaseMap(current_map3d);` (:432-433) frees the five buffers the previous map keeps pointing
// at, so the next frame reads freed (possibly re-used) memory.  This file restates the INTENDED
// behaviour: the previous map is the current map of the frame before (values as left by the
// fusion), everything else line by line.  Stated choices where C++ overloads decide the
// arithmetic: `fabs(float)` is the float overload (<cmath> via Qt headers), so the closeness test
// adds three floats; never-written X/Y/Z cells (malloc'ed, only read where D > 0) are 0.
// What IS pinned: the coefficient computation (Matrix::inv, K * H[0:3,0:4]) against the
// reference's own matrix.cpp through oracle/_ref (tests/test_map.py).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "oracle.h"

namespace {

// Matrix::solve with B = identity (Matrix::inv, libviso2/src/matrix.cpp:593-604, 648-760):
// Gauss-Jordan, full pivoting, ">=" pivot search, rows swapped, no column unscrambling of B
bool invert(const double* M, double* out, int n) {
    std::vector<double> A(M, M + n * n), B(n * n, 0.0);
    for (int i = 0; i < n; i++) B[i * n + i] = 1.0;
    std::vector<int> ipiv(n, 0);
    for (int it = 0; it < n; it++) {
        double big = 0.0;
        int irow = 0, icol = 0;
        for (int j = 0; j < n; j++)
            if (ipiv[j] != 1)
                for (int k = 0; k < n; k++)
                    if (ipiv[k] == 0 && fabs(A[j * n + k]) >= big) {
                        big = fabs(A[j * n + k]);
                        irow = j;
                        icol = k;
                    }
        ++ipiv[icol];
        if (irow != icol)
            for (int l = 0; l < n; l++) {
                std::swap(A[irow * n + l], A[icol * n + l]);
                std::swap(B[irow * n + l], B[icol * n + l]);
            }
        if (fabs(A[icol * n + icol]) < 1e-20) {   // Matrix::inv returns B as it stands then
            memcpy(out, B.data(), sizeof(double) * n * n);
            return false;
        }
        const double pivinv = 1.0 / A[icol * n + icol];
        A[icol * n + icol] = 1.0;
        for (int l = 0; l < n; l++) A[icol * n + l] *= pivinv;
        for (int l = 0; l < n; l++) B[icol * n + l] *= pivinv;
        for (int ll = 0; ll < n; ll++)
            if (ll != icol) {
                const double dum = A[ll * n + icol];
                A[ll * n + icol] = 0.0;
                for (int l = 0; l < n; l++) A[ll * n + l] -= A[icol * n + l] * dum;
                for (int l = 0; l < n; l++) B[ll * n + l] -= B[icol * n + l] * dum;
            }
    }
    memcpy(out, B.data(), sizeof(double) * n * n);
    return true;
}

struct Map3d {
    std::vector<float> I, D, X, Y, Z;
    int32_t w = 0, h = 0;
    bool present = false;
};

}  // namespace

struct orc_map {
    orc_map_params p;
    Map3d prev;
    std::vector<float> pts[2];   // x y z val
};

extern "C" {

// hcf: rows 0..2 of H_total; hfc: row 2 of inv(H_total); pfc: K * inv(H_total)[0:3, 0:4]
// (stereothread.cpp:196-199, 306-314; K from :450-455), all narrowed to float as there
void orc_map_coeffs(const orc_map_params* p, const double* H, float* hcf12, float* hfc4, float* pfc12) {
    for (int i = 0; i < 12; i++) hcf12[i] = (float)H[i];
    double Hi[16];
    invert(H, Hi, 4);
    for (int j = 0; j < 4; j++) hfc4[j] = (float)Hi[8 + j];
    // _f, _cu, _cv are float members; _K holds them as FLOAT (double)
    const double K[9] = {(double)p->f, 0, (double)p->cu, 0, (double)p->f, (double)p->cv, 0, 0, 1};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;   // Matrix::operator*: zero-initialised sum, k ascending
            for (int k = 0; k < 3; k++) s += K[i * 3 + k] * Hi[k * 4 + j];
            pfc12[i * 4 + j] = (float)s;
        }
}

orc_map* orc_map_create(const orc_map_params* p) {
    orc_map* m = new orc_map();
    m->p = *p;
    return m;
}
void orc_map_destroy(orc_map* m) { delete m; }
void orc_map_clear(orc_map* m) {   // clearReconstruction, stereothread.cpp:460-470
    m->prev = Map3d();
    m->pts[0].clear();
    m->pts[1].clear();
}

void orc_map_add(orc_map* m, const float* D1, const uint8_t* I1, const int32_t* dims, const double* H,
                 float gain) {
    const orc_map_params& p = m->p;
    const int32_t w = dims[0], h = dims[1], step = dims[2];
    float hcf[12], hfc[4], pfc[12];
    orc_map_coeffs(&p, H, hcf, hfc, pfc);
    // ---- createCurrentMap (:180-255)
    Map3d cur;
    cur.w = w;
    cur.h = h;
    cur.present = true;
    const size_t n = (size_t)w * h;
    cur.I.resize(n);
    cur.D.assign(D1, D1 + n);
    cur.X.assign(n, 0.f);
    cur.Y.assign(n, 0.f);
    cur.Z.assign(n, 0.f);
    for (int32_t v = 0; v < h; v++)
        for (int32_t u = 0; u < w; u++) cur.I[v * w + u] = (float)(((float)I1[v * step + u]) / 255.0);
    for (int32_t u = 0; u < w; u++)
        for (int32_t v = 0; v < h; v++) {
            const int32_t a = v * w + u;
            const float d = cur.D[a];
            if (d > 0) {
                const float z = (p.f * p.base) / d;
                if ((z > 0.1) && (z < p.max_dist)) {
                    const float x = ((float)u - p.cu) * p.base / d;
                    const float y = ((float)v - p.cv) * p.base / d;
                    cur.X[a] = hcf[0] * x + hcf[1] * y + hcf[2] * z + hcf[3];
                    cur.Y[a] = hcf[4] * x + hcf[5] * y + hcf[6] * z + hcf[7];
                    cur.Z[a] = hcf[8] * x + hcf[9] * y + hcf[10] * z + hcf[11];
                } else {
                    cur.D[a] = -1;
                }
            }
        }
    const int32_t margin = std::min(std::min(200, w / 2), h / 2);
    float gain_inv = 1;
    if (gain) gain_inv = 1.0 / gain;
    for (int32_t i = 0; i < margin; i++) {
        const float g = ((float)(margin - i) * gain_inv + (float)i * 1.0) / (float)margin;
        for (int32_t u = margin; u < w - margin; u++) {
            float& a = cur.I[i * w + u];
            a = std::min(std::max(g * a, (float)0), (float)1);
            float& b = cur.I[(h - i - 1) * w + u];
            b = std::min(std::max(g * b, (float)0), (float)1);
        }
        for (int32_t v = margin; v < h - margin; v++) {
            float& a = cur.I[v * w + i];
            a = std::min(std::max(g * a, (float)0), (float)1);
            float& b = cur.I[v * w + w - i - 1];
            b = std::min(std::max(g * b, (float)0), (float)1);
        }
    }
    // ---- association with the previous map (:297-398)
    if (m->prev.present) {
        Map3d& pr = m->prev;
        std::vector<float>& out = m->pts[0];
        out.clear();
        for (int32_t u = 0; u < pr.w; u++)
            for (int32_t v = 0; v < pr.h; v++) {
                const int32_t a = v * pr.w + u;
                const float d = pr.D[a];
                if (!(d > 0)) continue;
                const float x = pr.X[a], y = pr.Y[a], z = pr.Z[a];
                const float z2 = hfc[0] * x + hfc[1] * y + hfc[2] * z + hfc[3];
                bool added = false;
                if ((z2 > 0.1) && (z2 < p.max_dist)) {
                    const float w2 = pfc[8] * x + pfc[9] * y + pfc[10] * z + pfc[11];
                    const int32_t u2 = (int32_t)((pfc[0] * x + pfc[1] * y + pfc[2] * z + pfc[3]) / w2);
                    const int32_t v2 = (int32_t)((pfc[4] * x + pfc[5] * y + pfc[6] * z + pfc[7]) / w2);
                    if (u2 >= 0 && u2 < cur.w && v2 >= 0 && v2 < cur.h) {
                        const int32_t a2 = v2 * cur.w + u2;
                        if (cur.D[a2] > 0) {
                            if (fabsf(x - cur.X[a2]) + fabsf(y - cur.Y[a2]) + fabsf(z - cur.Z[a2]) < 0.2) {
                                cur.X[a2] = (cur.X[a2] + x) / 2.0;
                                cur.Y[a2] = (cur.Y[a2] + y) / 2.0;
                                cur.Z[a2] = (cur.Z[a2] + z) / 2.0;
                                cur.I[a2] = (cur.I[a2] + pr.I[a]) / 2.0;
                                added = true;
                            }
                        } else {
                            cur.X[a2] = x;
                            cur.Y[a2] = y;
                            cur.Z[a2] = z;
                            cur.I[a2] = pr.I[a];
                            cur.D[a2] = 1;
                            added = true;
                        }
                    }
                }
                if (added) {
                    pr.D[a] = -1;
                } else {
                    out.push_back(x);
                    out.push_back(y);
                    out.push_back(z);
                    out.push_back(pr.I[a]);
                }
            }
    } else {
        m->pts[0].clear();
    }
    // ---- current point cloud (:410-428)
    std::vector<float>& pc = m->pts[1];
    pc.clear();
    for (int32_t u = 0; u < w; u++)
        for (int32_t v = 0; v < h; v++) {
            const int32_t a = v * w + u;
            if (cur.D[a] > 0) {
                pc.push_back(cur.X[a]);
                pc.push_back(cur.Y[a]);
                pc.push_back(cur.Z[a]);
                pc.push_back(cur.I[a]);
            }
        }
    m->prev = cur;   // the intended "_previous_map3d = current_map3d" (:432)
}

int64_t orc_map_points(const orc_map* m, int32_t which, float* out, int64_t cap) {
    const std::vector<float>& v = m->pts[which ? 1 : 0];
    const int64_t n = (int64_t)(v.size() / 4);
    if (out)
        for (int64_t i = 0; i < n && i < cap; i++) memcpy(out + 4 * i, &v[4 * i], 16);
    return n;
}

// colour-coded disparity, stereothread.cpp:117-147 (untouched cells -- none for finite input -- are 0)
void orc_disparity_colormap(const float* D, int64_t n, float* rgb) {
    const float d_max = 200;
    for (int64_t i = 0; i < n; i++) {
        float* c = rgb + 3 * i;
        c[0] = c[1] = c[2] = 0;
        const float val = std::min(D[i] / d_max, (float)1.0);
        if (val <= 0) continue;
        const float h2 = 6.0 * (1.0 - val);
        const float x = 1.0 * (1.0 - fabs(fmodf(h2, (float)2.0) - 1.0));
        if (0 <= h2 && h2 < 1)       { c[0] = 1; c[1] = x; c[2] = 0; }
        else if (1 <= h2 && h2 < 2)  { c[0] = x; c[1] = 1; c[2] = 0; }
        else if (2 <= h2 && h2 < 3)  { c[0] = 0; c[1] = 1; c[2] = x; }
        else if (3 <= h2 && h2 < 4)  { c[0] = 0; c[1] = x; c[2] = 1; }
        else if (4 <= h2 && h2 < 5)  { c[0] = x; c[1] = 0; c[2] = 1; }
        else if (5 <= h2 && h2 <= 6) { c[0] = 1; c[1] = 0; c[2] = x; }
    }
}

// the maps after the last frame (I, D, X, Y, Z planes of w*h floats): tests compare them too
void orc_map_planes(const orc_map* m, float* out5) {
    const size_t n = (size_t)m->prev.w * m->prev.h;
    const std::vector<float>* pl[5] = {&m->prev.I, &m->prev.D, &m->prev.X, &m->prev.Y, &m->prev.Z};
    for (int k = 0; k < 5; k++) memcpy(out5 + k * n, pl[k]->data(), n * sizeof(float));
}

}  // extern "C"
