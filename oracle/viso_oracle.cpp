// TEST INFRASTRUCTURE -- scalar CPU restatement of libviso2's Matcher
// (feature extraction, circular matching, outlier filter, refinement).
// See oracle/oracle.h for the rules.  Written from the algorithm with plain
// loops; every function cites the reference lines it restates (paths relative
// to the reference checkout).  Pinned bit-for-bit -- feature tables, match
// indices and coordinates -- against oracle/_ref/libref_viso.so and the golden
// quad fixture (tests/test_oracle_viso.py).
//
// Not restated: Shewchuk's Triangle (supplied as a callback, like for ELAS).
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

inline int32_t iabs(int32_t x) { return x < 0 ? -x : x; }
inline uint8_t sat_u8(int32_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

struct Feat {
    int32_t v[12];  // u, v, 0, class, d1..d8 (matcher.cpp:854-877)
};

struct Image {
    int32_t w = 0, h = 0, bpl = 0;
    std::vector<uint8_t> px;
};

// one camera image of one frame with everything pushBack derives from it
struct View {
    bool valid = false;
    Image I;                          // full resolution copy (matcher.cpp:171-196)
    int32_t mw = 0, mh = 0, mbpl = 0; // matching resolution
    std::vector<uint8_t> du, dv;      // Sobel 5x5 at matching resolution
    std::vector<uint8_t> du_full, dv_full;
    std::vector<int16_t> f1, f2;      // blob / checkerboard (kept for inspection)
    std::vector<Feat> sparse, dense;
};

// ---------------------------------------------------------------------------
// M2  filter::sobel5x5   libviso2/src/filter.cpp:474 (+306-361, 154-222, 93-152)
//   du: vertical [1 4 6 4 1], horizontal [1 2 0 -2 -1];  dv: vertical
//   [1 2 0 -2 -1], horizontal [1 4 6 4 1]; >>7 (arithmetic), +128, saturate.
//   Defined on rows 2..h-3, cols 2..w-3; 0 elsewhere.
// ---------------------------------------------------------------------------
void sobel5x5(const uint8_t* I, int32_t w, int32_t h, int32_t bpl, uint8_t* du, uint8_t* dv) {
    memset(du, 0, (size_t)bpl * h);
    memset(dv, 0, (size_t)bpl * h);
    std::vector<int32_t> S(w), T(w);
    for (int32_t v = 2; v < h - 2; v++) {
        const uint8_t* r0 = I + (size_t)(v - 2) * bpl;
        const uint8_t* r1 = r0 + bpl;
        const uint8_t* r2 = r1 + bpl;
        const uint8_t* r3 = r2 + bpl;
        const uint8_t* r4 = r3 + bpl;
        for (int32_t u = 0; u < w; u++) {
            S[u] = r0[u] + 4 * r1[u] + 6 * r2[u] + 4 * r3[u] + r4[u];
            T[u] = r0[u] + 2 * r1[u] - 2 * r3[u] - r4[u];
        }
        for (int32_t u = 2; u < w - 2; u++) {
            du[(size_t)v * bpl + u] = sat_u8(((S[u - 2] + 2 * S[u - 1] - 2 * S[u + 1] - S[u + 2]) >> 7) + 128);
            dv[(size_t)v * bpl + u] =
                sat_u8(((T[u - 2] + 4 * T[u - 1] + 6 * T[u] + 4 * T[u + 1] + T[u + 2]) >> 7) + 128);
        }
    }
}

// M3  filter::blob5x5 (filter.cpp:507-532): -sum5x5 + 2*sum3x3 + 7*centre
//     filter::checkerboard5x5 (filter.cpp:492-497, 364-417): [1 1 0 -1 -1]^T x [1 1 0 -1 -1]
void blob_checker(const uint8_t* I, int32_t w, int32_t h, int32_t bpl, int16_t* f1, int16_t* f2) {
    memset(f1, 0, sizeof(int16_t) * (size_t)bpl * h);
    memset(f2, 0, sizeof(int16_t) * (size_t)bpl * h);
    static const int sg[5] = {1, 1, 0, -1, -1};
    for (int32_t v = 2; v < h - 2; v++)
        for (int32_t u = 2; u < w - 2; u++) {
            int32_t s5 = 0, s3 = 0, ck = 0;
            for (int dy = -2; dy <= 2; dy++)
                for (int dx = -2; dx <= 2; dx++) {
                    int32_t p = I[(size_t)(v + dy) * bpl + u + dx];
                    s5 += p;
                    if (dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1) s3 += p;
                    ck += sg[dy + 2] * sg[dx + 2] * p;
                }
            f1[(size_t)v * bpl + u] = (int16_t)(-s5 + 2 * s3 + 7 * I[(size_t)v * bpl + u]);
            f2[(size_t)v * bpl + u] = (int16_t)ck;
        }
}

struct Maximum {
    int32_t u, v, val, c;
};

// M4  Matcher::nonMaximumSuppression   matcher.cpp:395-530
void nms(const int16_t* f1, const int16_t* f2, int32_t w, int32_t h, int32_t bpl, int32_t n,
         int32_t tau, int32_t margin, std::vector<Maximum>& out) {
    for (int32_t i = n + margin; i < w - n - margin; i += n + 1)
        for (int32_t j = n + margin; j < h - n - margin; j += n + 1) {
            const int16_t* F[2] = {f1, f2};
            int32_t mini[2], minj[2], maxi[2], maxj[2], minv[2], maxv[2];
            for (int k = 0; k < 2; k++) {
                mini[k] = maxi[k] = i;
                minj[k] = maxj[k] = j;
                minv[k] = maxv[k] = F[k][(size_t)j * bpl + i];
            }
            for (int32_t i2 = i; i2 <= i + n; i2++)
                for (int32_t j2 = j; j2 <= j + n; j2++)
                    for (int k = 0; k < 2; k++) {
                        int32_t c = F[k][(size_t)j2 * bpl + i2];
                        if (c < minv[k]) {
                            mini[k] = i2; minj[k] = j2; minv[k] = c;
                        } else if (c > maxv[k]) {
                            maxi[k] = i2; maxj[k] = j2; maxv[k] = c;
                        }
                    }
            // class order: f1 min, f1 max, f2 min, f2 max
            for (int k = 0; k < 2; k++)
                for (int mm = 0; mm < 2; mm++) {
                    const bool is_min = mm == 0;
                    const int32_t ci = is_min ? mini[k] : maxi[k], cj = is_min ? minj[k] : maxj[k];
                    const int32_t cv = is_min ? minv[k] : maxv[k];
                    bool ok = true;
                    for (int32_t i2 = ci - n; ok && i2 <= std::min(ci + n, w - 1 - margin); i2++)
                        for (int32_t j2 = cj - n; j2 <= std::min(cj + n, h - 1 - margin); j2++) {
                            int32_t c = F[k][(size_t)j2 * bpl + i2];
                            bool beats = is_min ? c < cv : c > cv;
                            if (beats && (i2 < i || i2 > i + n || j2 < j || j2 > j + n)) {
                                ok = false;
                                break;
                            }
                        }
                    if (!ok) continue;
                    if (is_min ? cv <= -tau : cv >= tau) out.push_back({ci, cj, cv, 2 * k + mm});
                }
        }
}

// M5  Matcher::computeDescriptor   matcher.cpp:534-579: 16 (du,dv) pairs around (u, v-1)
void descriptor32(const uint8_t* du, const uint8_t* dv, int32_t bpl, int32_t u, int32_t v, uint8_t* d) {
    const ptrdiff_t m1 = (ptrdiff_t)(v - 1) * bpl + u;
    const ptrdiff_t m3 = m1 - 2 * bpl, m5 = m3 - 2 * bpl, p1 = m1 + 2 * bpl, p3 = p1 + 2 * bpl,
                    p5 = p3 + 2 * bpl;
    const ptrdiff_t at[16] = {m1 - 3, p1 - 3, m1 - 1, p1 - 1, m1 + 3, p1 + 3, m1 + 1, p1 + 1,
                              m5 - 1, p5 - 1, m5 + 1, p5 + 1, m3 - 5, p3 - 5, m3 + 5, p3 + 5};
    for (int k = 0; k < 16; k++) {
        d[2 * k] = du[at[k]];
        d[2 * k + 1] = dv[at[k]];
    }
}

// Matcher::computeSmallDescriptor   matcher.cpp:583-611 (the ELAS descriptor)
void descriptor16(const uint8_t* du, const uint8_t* dv, int32_t bpl, int32_t u, int32_t v, uint8_t* d) {
    const ptrdiff_t a2 = (ptrdiff_t)v * bpl + u, a1 = a2 - bpl, a0 = a1 - bpl, a3 = a2 + bpl, a4 = a3 + bpl;
    d[0] = du[a0]; d[1] = du[a1 - 2]; d[2] = du[a1]; d[3] = du[a1 + 2];
    d[4] = du[a2 - 1]; d[5] = du[a2]; d[6] = du[a2]; d[7] = du[a2 + 1];
    d[8] = du[a3 - 2]; d[9] = du[a3]; d[10] = du[a3 + 2]; d[11] = du[a4];
    d[12] = dv[a1]; d[13] = dv[a2 - 1]; d[14] = dv[a2 + 1]; d[15] = dv[a3];
}

inline int32_t sad_bytes(const uint8_t* a, const uint8_t* b, int n) {
    int32_t s = 0;
    for (int i = 0; i < n; i++) s += iabs((int32_t)a[i] - (int32_t)b[i]);
    return s;
}

typedef svh_p_match PM;

struct Range {
    float u_min[4], u_max[4], v_min[4], v_max[4];
};

}  // namespace

struct orc_matcher {
    svh_matcher_params p;
    int32_t margin;
    View prev[2], cur[2];     // [0] left, [1] right
    int32_t dims_p[3], dims_c[3];
    std::vector<PM> m1, m2;   // _p_matched_1, _p_matched_2
    std::vector<Range> ranges;
    std::vector<PM> stage[SVH_M_STAGE_COUNT];
    orc_triangulate_fn tri_fn;
};

namespace {

// M1..M5  Matcher::computeFeatures   matcher.cpp:780-878
void compute_features(orc_matcher* m, View& V) {
    const svh_matcher_params& p = m->p;
    const Image& I = V.I;
    const uint8_t* Im = I.px.data();
    std::vector<uint8_t> half;
    V.mw = I.w; V.mh = I.h; V.mbpl = I.bpl;
    if (p.half_resolution) {
        // createHalfResolutionImage / getHalfResolutionDimensions (matcher.cpp:751-776)
        V.mw = I.w / 2;
        V.mh = I.h / 2;
        V.mbpl = V.mw + 15 - (V.mw - 1) % 16;
        half.assign((size_t)V.mbpl * V.mh, 0);
        for (int32_t v = 0; v < V.mh; v++)
            for (int32_t u = 0; u < V.mw; u++)
                half[(size_t)v * V.mbpl + u] =
                    (uint8_t)(((int32_t)I.px[(size_t)(2 * v) * I.bpl + 2 * u] + I.px[(size_t)(2 * v) * I.bpl + 2 * u + 1] +
                               I.px[(size_t)(2 * v + 1) * I.bpl + 2 * u] + I.px[(size_t)(2 * v + 1) * I.bpl + 2 * u + 1]) / 4);
        Im = half.data();
        V.du_full.resize((size_t)I.bpl * I.h);
        V.dv_full.resize((size_t)I.bpl * I.h);
        sobel5x5(I.px.data(), I.w, I.h, I.bpl, V.du_full.data(), V.dv_full.data());
    } else {
        V.du_full.clear();
        V.dv_full.clear();
    }
    const size_t mn = (size_t)V.mbpl * V.mh;
    V.du.resize(mn); V.dv.resize(mn); V.f1.resize(mn); V.f2.resize(mn);
    sobel5x5(Im, V.mw, V.mh, V.mbpl, V.du.data(), V.dv.data());
    blob_checker(Im, V.mw, V.mh, V.mbpl, V.f1.data(), V.f2.data());

    const int32_t s = p.half_resolution ? 2 : 1;
    for (int pass = 0; pass < 2; pass++) {
        std::vector<Feat>& out = pass == 0 ? V.sparse : V.dense;
        out.clear();
        if (pass == 0 && !p.multi_stage) continue;
        int32_t n = p.nms_n;
        if (pass == 0) {  // matcher.cpp:824-828
            n = p.nms_n * 3;
            if (n > 10) n = std::max(p.nms_n, 10);
        }
        std::vector<Maximum> mx;
        nms(V.f1.data(), V.f2.data(), V.mw, V.mh, V.mbpl, n, p.nms_tau, m->margin, mx);
        out.resize(mx.size());
        for (size_t i = 0; i < mx.size(); i++) {
            Feat& f = out[i];
            f.v[0] = mx[i].u * s;
            f.v[1] = mx[i].v * s;
            f.v[2] = 0;
            f.v[3] = mx[i].c;
            descriptor32(V.du.data(), V.dv.data(), V.mbpl, mx[i].u, mx[i].v, (uint8_t*)&f.v[4]);
        }
    }
    V.valid = true;
}

// M6  Matcher::createIndexVector   matcher.cpp:1036-1057
void index_vector(const svh_matcher_params& p, const std::vector<Feat>& t, int32_t ub, int32_t vb,
                  std::vector<std::vector<int32_t>>& k) {
    k.assign((size_t)4 * ub * vb, std::vector<int32_t>());
    for (int32_t i = 0; i < (int32_t)t.size(); i++) {
        int32_t u_bin = std::min((int32_t)floor((float)t[i].v[0] / (float)p.match_binsize), ub - 1);
        int32_t v_bin = std::min((int32_t)floor((float)t[i].v[1] / (float)p.match_binsize), vb - 1);
        k[((size_t)t[i].v[3] * vb + v_bin) * ub + u_bin].push_back(i);
    }
}

// M7  Matcher::findMatch   matcher.cpp:1061-1157
int32_t find_match(const orc_matcher* m, const std::vector<Feat>& t1, int32_t i1,
                   const std::vector<Feat>& t2, const std::vector<std::vector<int32_t>>& k2, int32_t ub,
                   int32_t vb, int32_t stat_bin, int32_t stage, bool flow, bool use_prior, double u_ = -1,
                   double v_ = -1) {
    const svh_matcher_params& p = m->p;
    int32_t min_ind = 0;
    double min_cost = 10000000;
    const int32_t u1 = t1[i1].v[0], v1 = t1[i1].v[1], c = t1[i1].v[3];
    const uint8_t* d1 = (const uint8_t*)&t1[i1].v[4];
    float u_min, u_max, v_min, v_max;
    if (use_prior) {
        u_min = u1 + m->ranges[stat_bin].u_min[stage];
        u_max = u1 + m->ranges[stat_bin].u_max[stage];
        v_min = v1 + m->ranges[stat_bin].v_min[stage];
        v_max = v1 + m->ranges[stat_bin].v_max[stage];
    } else {
        u_min = u1 - p.match_radius;
        u_max = u1 + p.match_radius;
        v_min = v1 - p.match_radius;
        v_max = v1 + p.match_radius;
    }
    if (!flow) {
        v_min = v1 - p.match_disp_tolerance;
        v_max = v1 + p.match_disp_tolerance;
    }
    auto bin = [&](float x, int32_t nb) {
        return std::min(std::max((int32_t)floor(x / (float)p.match_binsize), 0), nb - 1);
    };
    const int32_t ub0 = bin(u_min, ub), ub1 = bin(u_max, ub), vb0 = bin(v_min, vb), vb1 = bin(v_max, vb);
    for (int32_t u_bin = ub0; u_bin <= ub1; u_bin++)
        for (int32_t v_bin = vb0; v_bin <= vb1; v_bin++) {
            const std::vector<int32_t>& lst = k2[((size_t)c * vb + v_bin) * ub + u_bin];
            for (size_t q = 0; q < lst.size(); q++) {
                const int32_t i2 = lst[q];
                const int32_t u2 = t2[i2].v[0], v2 = t2[i2].v[1];
                if (u2 >= u_min && u2 <= u_max && v2 >= v_min && v2 <= v_max) {
                    double cost = (double)sad_bytes(d1, (const uint8_t*)&t2[i2].v[4], 32);
                    if (u_ >= 0 && v_ >= 0) {
                        double du = (double)u2 - u_, dv = (double)v2 - v_;
                        cost += 4 * sqrt(du * du + dv * dv);
                    }
                    if (cost < min_cost) {
                        min_ind = i2;
                        min_cost = cost;
                    }
                }
            }
        }
    return min_ind;
}

PM make_match(float u1p, float v1p, int32_t i1p, float u2p, float v2p, int32_t i2p, float u1c, float v1c,
              int32_t i1c, float u2c, float v2c, int32_t i2c) {
    PM q;
    q.u1p = u1p; q.v1p = v1p; q.i1p = i1p; q.u2p = u2p; q.v2p = v2p; q.i2p = i2p;
    q.u1c = u1c; q.v1c = v1c; q.i1c = i1c; q.u2c = u2c; q.v2c = v2c; q.i2c = i2c;
    return q;
}

// M8  Matcher::matching   matcher.cpp:1161-1379
void matching(orc_matcher* m, bool dense, std::vector<PM>& out, int32_t method, bool use_prior,
              const double* Tr) {
    const svh_matcher_params& p = m->p;
    const std::vector<Feat>& m1p = dense ? m->prev[0].dense : m->prev[0].sparse;
    const std::vector<Feat>& m2p = dense ? m->prev[1].dense : m->prev[1].sparse;
    const std::vector<Feat>& m1c = dense ? m->cur[0].dense : m->cur[0].sparse;
    const std::vector<Feat>& m2c = dense ? m->cur[1].dense : m->cur[1].sparse;
    const int32_t ub = (int32_t)ceil((float)m->dims_c[0] / (float)p.match_binsize);
    const int32_t vb = (int32_t)ceil((float)m->dims_c[1] / (float)p.match_binsize);
    std::vector<std::vector<int32_t>> k1p, k2p, k1c, k2c;
    auto stat = [&](int32_t u, int32_t v) {
        int32_t u_bin = std::min((int32_t)floor((float)u / (float)p.match_binsize), ub - 1);
        int32_t v_bin = std::min((int32_t)floor((float)v / (float)p.match_binsize), vb - 1);
        return v_bin * ub + u_bin;
    };
    if (method == 0) {
        index_vector(p, m1p, ub, vb, k1p);
        index_vector(p, m1c, ub, vb, k1c);
        std::vector<uint8_t> M((size_t)m->dims_c[0] * m->dims_c[1], 0);
        for (int32_t i1c = 0; i1c < (int32_t)m1c.size(); i1c++) {
            const int32_t u1c = m1c[i1c].v[0], v1c = m1c[i1c].v[1];
            const int32_t sb = stat(u1c, v1c);
            int32_t i1p = find_match(m, m1c, i1c, m1p, k1p, ub, vb, sb, 0, true, use_prior);
            int32_t i1c2 = find_match(m, m1p, i1p, m1c, k1c, ub, vb, sb, 1, true, use_prior);
            if (i1c2 == i1c) {
                size_t a = (size_t)v1c * m->dims_c[0] + u1c;
                if (!M[a]) {
                    out.push_back(make_match(m1p[i1p].v[0], m1p[i1p].v[1], i1p, -1, -1, -1, u1c, v1c, i1c, -1, -1, -1));
                    M[a] = 1;
                }
            }
        }
    } else if (method == 1) {
        index_vector(p, m1c, ub, vb, k1c);
        index_vector(p, m2c, ub, vb, k2c);
        std::vector<uint8_t> M((size_t)m->dims_c[0] * m->dims_c[1], 0);
        for (int32_t i1c = 0; i1c < (int32_t)m1c.size(); i1c++) {
            const int32_t u1c = m1c[i1c].v[0], v1c = m1c[i1c].v[1];
            const int32_t sb = stat(u1c, v1c);
            int32_t i2c = find_match(m, m1c, i1c, m2c, k2c, ub, vb, sb, 0, false, use_prior);
            int32_t i1c2 = find_match(m, m2c, i2c, m1c, k1c, ub, vb, sb, 1, false, use_prior);
            if (i1c2 == i1c) {
                const int32_t u2c = m2c[i2c].v[0], v2c = m2c[i2c].v[1];
                if (u1c >= u2c) {
                    size_t a = (size_t)v1c * m->dims_c[0] + u1c;
                    if (!M[a]) {
                        out.push_back(make_match(-1, -1, -1, -1, -1, -1, u1c, v1c, i1c, u2c, v2c, i2c));
                        M[a] = 1;
                    }
                }
            }
        }
    } else {
        index_vector(p, m1p, ub, vb, k1p);
        index_vector(p, m2p, ub, vb, k2p);
        index_vector(p, m1c, ub, vb, k1c);
        index_vector(p, m2c, ub, vb, k2c);
        for (int32_t i1p = 0; i1p < (int32_t)m1p.size(); i1p++) {
            const int32_t u1p = m1p[i1p].v[0], v1p = m1p[i1p].v[1];
            const int32_t sb = stat(u1p, v1p);
            const int32_t i2p = find_match(m, m1p, i1p, m2p, k2p, ub, vb, sb, 0, false, use_prior);
            const int32_t u2p = m2p[i2p].v[0], v2p = m2p[i2p].v[1];
            int32_t i2c;
            if (Tr) {
                // predicted position in the current right image (matcher.cpp:1312-1327)
                double d = std::max((double)u1p - (double)u2p, 1.0);
                double x1p = ((double)u1p - p.cu) * p.base / d;
                double y1p = ((double)v1p - p.cv) * p.base / d;
                double z1p = p.f * p.base / d;
                double x2c = Tr[0] * x1p + Tr[1] * y1p + Tr[2] * z1p + Tr[3] - p.base;
                double y2c = Tr[4] * x1p + Tr[5] * y1p + Tr[6] * z1p + Tr[7];
                double z2c = Tr[8] * x1p + Tr[9] * y1p + Tr[10] * z1p + Tr[11];
                double u2c_ = p.f * x2c / z2c + p.cu;
                double v2c_ = p.f * y2c / z2c + p.cv;
                i2c = find_match(m, m2p, i2p, m2c, k2c, ub, vb, sb, 1, true, use_prior, u2c_, v2c_);
            } else {
                i2c = find_match(m, m2p, i2p, m2c, k2c, ub, vb, sb, 1, true, use_prior);
            }
            const int32_t i1c = find_match(m, m2c, i2c, m1c, k1c, ub, vb, sb, 2, false, use_prior);
            int32_t i1p2;
            if (Tr) i1p2 = find_match(m, m1c, i1c, m1p, k1p, ub, vb, sb, 3, true, use_prior, u1p, v1p);
            else    i1p2 = find_match(m, m1c, i1c, m1p, k1p, ub, vb, sb, 3, true, use_prior);
            if (i1p2 == i1p) {
                const int32_t u2c = m2c[i2c].v[0], v2c = m2c[i2c].v[1];
                const int32_t u1c = m1c[i1c].v[0], v1c = m1c[i1c].v[1];
                if (u1p >= u2p && u1c >= u2c)
                    out.push_back(make_match(u1p, v1p, i1p, u2p, v2p, i2p, u1c, v1c, i1c, u2c, v2c, i2c));
            }
        }
    }
}

// M9  Matcher::removeOutliers   matcher.cpp:1383-1570
void remove_outliers(orc_matcher* m, std::vector<PM>& pm, int32_t method) {
    if (pm.size() <= 3) return;
    const svh_matcher_params& p = m->p;
    const int32_t n = (int32_t)pm.size();
    std::vector<float> pts((size_t)2 * n);
    for (int32_t i = 0; i < n; i++) {
        pts[2 * i] = pm[i].u1c;
        pts[2 * i + 1] = pm[i].v1c;
    }
    std::vector<int32_t> tri((size_t)3 * (2 * n + 16));
    int32_t nt = m->tri_fn(pts.data(), n, tri.data(), 2 * n + 16);
    if (nt < 0) nt = 0;
    std::vector<int32_t> sup(n, 0);
    auto flow_ok = [&](const PM& a, const PM& b) {
        float au = a.u1c - a.u1p, av = a.v1c - a.v1p, bu = b.u1c - b.u1p, bv = b.v1c - b.v1p;
        return fabs(au - bu) + fabs(av - bv) < p.outlier_flow_tolerance;
    };
    auto disp_ok = [&](const PM& a, const PM& b, bool prev) {
        float da = prev ? a.u1p - a.u2p : a.u1c - a.u2c, db = prev ? b.u1p - b.u2p : b.u1c - b.u2c;
        return fabs(da - db) < p.outlier_disp_tolerance;
    };
    for (int32_t t = 0; t < nt; t++) {
        const int32_t c[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
        const int e[3][2] = {{0, 1}, {1, 2}, {0, 2}};
        for (int k = 0; k < 3; k++) {
            const PM& a = pm[c[e[k][0]]];
            const PM& b = pm[c[e[k][1]]];
            bool ok;
            if (method == 0) ok = flow_ok(a, b);
            else if (method == 1) ok = disp_ok(a, b, false);
            else ok = disp_ok(a, b, true) && flow_ok(a, b);
            if (ok) {
                sup[c[e[k][0]]]++;
                sup[c[e[k][1]]]++;
            }
        }
    }
    std::vector<PM> keep;
    for (int32_t i = 0; i < n; i++)
        if (sup[i] >= 4) keep.push_back(pm[i]);
    pm.swap(keep);
}

// M10  Matcher::computePriorStatistics   matcher.cpp:882-1032
void prior_statistics(orc_matcher* m, const std::vector<PM>& pm, int32_t method) {
    const svh_matcher_params& p = m->p;
    const int32_t ub = (int32_t)ceil((float)m->dims_c[0] / (float)p.match_binsize);
    const int32_t vb = (int32_t)ceil((float)m->dims_c[1] / (float)p.match_binsize);
    const int32_t stages = method == 2 ? 4 : 2;
    struct Delta { float v[8]; };
    std::vector<std::vector<Delta>> acc((size_t)ub * vb);
    for (size_t q = 0; q < pm.size(); q++) {
        const PM& it = pm[q];
        Delta d;
        for (int k = 0; k < 8; k++) d.v[k] = 0;   // unused slots are never read
        if (method == 0) {
            d.v[0] = it.u1p - it.u1c; d.v[1] = it.v1p - it.v1c;
            d.v[2] = it.u1c - it.u1p; d.v[3] = it.v1c - it.v1p;
        } else if (method == 1) {
            d.v[0] = it.u2c - it.u1c; d.v[1] = 0;
            d.v[2] = it.u1c - it.u2c; d.v[3] = 0;
        } else {
            d.v[0] = it.u2p - it.u1p; d.v[1] = 0;
            d.v[2] = it.u2c - it.u2p; d.v[3] = it.v2c - it.v2p;
            d.v[4] = it.u1c - it.u2c; d.v[5] = 0;
            d.v[6] = it.u1p - it.u1c; d.v[7] = it.v1p - it.v1c;
        }
        const float ru = method < 2 ? it.u1c : it.u1p, rv = method < 2 ? it.v1c : it.v1p;
        auto clampb = [](int32_t x, int32_t nb) { return std::min(std::max(x, 0), nb - 1); };
        const int32_t cu = (int32_t)floor(ru / (float)p.match_binsize);
        const int32_t cv = (int32_t)floor(rv / (float)p.match_binsize);
        for (int32_t v_bin = clampb(cv - 1, vb); v_bin <= clampb(cv + 1, vb); v_bin++)
            for (int32_t u_bin = clampb(cu - 1, ub); u_bin <= clampb(cu + 1, ub); u_bin++)
                acc[(size_t)v_bin * ub + u_bin].push_back(d);
    }
    m->ranges.clear();
    for (int32_t v_bin = 0; v_bin < vb; v_bin++)
        for (int32_t u_bin = 0; u_bin < ub; u_bin++) {
            float lo[8], hi[8];
            const std::vector<Delta>& a = acc[(size_t)v_bin * ub + u_bin];
            for (int k = 0; k < 8; k++) {
                lo[k] = a.empty() ? (float)-p.match_radius : (float)+1000000;
                hi[k] = a.empty() ? (float)+p.match_radius : (float)-1000000;
            }
            for (size_t q = 0; q < a.size(); q++)
                for (int k = 0; k < stages * 2; k++) {
                    if (a[q].v[k] < lo[k]) lo[k] = a[q].v[k];
                    if (a[q].v[k] > hi[k]) hi[k] = a[q].v[k];
                }
            Range r;
            memset(&r, 0, sizeof(r));
            for (int i = 0; i < stages; i++) {
                for (int ax = 0; ax < 2; ax++) {
                    float span = hi[i * 2 + ax] - lo[i * 2 + ax];
                    if (span < 20) {
                        lo[i * 2 + ax] -= ceil((20 - span) / 2);
                        hi[i * 2 + ax] += ceil((20 - span) / 2);
                    }
                }
                r.u_min[i] = lo[i * 2]; r.u_max[i] = hi[i * 2];
                r.v_min[i] = lo[i * 2 + 1]; r.v_max[i] = hi[i * 2 + 1];
            }
            m->ranges.push_back(r);
        }
}

// M11  Matcher::relocateMinimum   matcher.cpp:1666-1711
void relocate(const orc_matcher* m, const View& V1, const int32_t* dims1, const View& V2,
              const int32_t* dims2, float u1, float v1, float& u2, float& v2) {
    const bool half = m->p.half_resolution != 0;
    const uint8_t* du1 = half ? V1.du_full.data() : V1.du.data();
    const uint8_t* dv1 = half ? V1.dv_full.data() : V1.dv.data();
    const uint8_t* du2 = half ? V2.du_full.data() : V2.du.data();
    const uint8_t* dv2 = half ? V2.dv_full.data() : V2.dv.data();
    if (u2 - 2 < m->margin || u2 + 2 > dims2[0] - 1 - m->margin || v2 - 2 < m->margin ||
        v2 + 2 > dims2[1] - 1 - m->margin)
        return;
    uint8_t ref[16], d[16];
    descriptor16(du1, dv1, dims1[2], (int32_t)u1, (int32_t)v1, ref);
    int32_t best = 0, best_cost = 0;
    for (int32_t k = 0; k < 25; k++) {
        descriptor16(du2, dv2, dims2[2], (int32_t)u2 + k % 5 - 2, (int32_t)v2 + k / 5 - 2, d);
        int32_t c = sad_bytes(ref, d, 16);
        if (k == 0 || c < best_cost) {
            best = k;
            best_cost = c;
        }
    }
    u2 += (float)(best % 5) - 2.0;
    v2 += (float)(best / 5) - 2.0;
}

// Matrix::solve for an n x n system with one right-hand side   libviso2/src/matrix.cpp:648-760
// (Gauss-Jordan, full pivoting, ">=" pivot search: the last maximum wins)
bool gauss_jordan(double* A, double* B, int n, double eps = 1e-20) {
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
        if (irow != icol) {
            for (int l = 0; l < n; l++) std::swap(A[irow * n + l], A[icol * n + l]);
            std::swap(B[irow], B[icol]);
        }
        if (fabs(A[icol * n + icol]) < eps) return false;
        const double pivinv = 1.0 / A[icol * n + icol];
        A[icol * n + icol] = 1.0;
        for (int l = 0; l < n; l++) A[icol * n + l] *= pivinv;
        B[icol] *= pivinv;
        for (int ll = 0; ll < n; ll++)
            if (ll != icol) {
                const double dum = A[ll * n + icol];
                A[ll * n + icol] = 0.0;
                for (int l = 0; l < n; l++) A[ll * n + l] -= A[icol * n + l] * dum;
                B[ll] -= B[icol] * dum;
            }
    }
    return true;
}

// design matrix of the 3x3 quadratic fit (matcher.cpp:1725-1733)
const double kFitA[9][6] = {{1, 1, 1, -1, -1, 1}, {0, 1, 0, 0, -1, 1}, {1, 1, -1, 1, -1, 1},
                            {1, 0, 0, -1, 0, 1},  {0, 0, 0, 0, 0, 1},  {1, 0, 0, 1, 0, 1},
                            {1, 1, -1, -1, 1, 1}, {0, 1, 0, 0, 1, 1},  {1, 1, 1, 1, 1, 1}};

// M11'  Matcher::parabolicFitting   matcher.cpp:1574-1662; false drops the match
bool parabolic(const orc_matcher* m, const View& V1, const int32_t* dims1, const View& V2,
               const int32_t* dims2, float u1, float v1, float& u2, float& v2) {
    const bool half = m->p.half_resolution != 0;
    const uint8_t* du1 = half ? V1.du_full.data() : V1.du.data();
    const uint8_t* dv1 = half ? V1.dv_full.data() : V1.dv.data();
    const uint8_t* du2 = half ? V2.du_full.data() : V2.du.data();
    const uint8_t* dv2 = half ? V2.dv_full.data() : V2.dv.data();
    if (u2 - 3 < m->margin || u2 + 3 > dims2[0] - 1 - m->margin || v2 - 3 < m->margin ||
        v2 + 3 > dims2[1] - 1 - m->margin)
        return false;
    uint8_t ref[16], d[16];
    descriptor16(du1, dv1, dims1[2], (int32_t)u1, (int32_t)v1, ref);
    int32_t cost[49];
    for (int32_t k = 0; k < 49; k++) {
        descriptor16(du2, dv2, dims2[2], (int32_t)u2 + k % 7 - 3, (int32_t)v2 + k / 7 - 3, d);
        cost[k] = sad_bytes(ref, d, 16);
    }
    int32_t min_ind = 0, min_cost = cost[0];
    for (int32_t i = 1; i < 49; i++)
        if (cost[i] < min_cost) {
            min_ind = i;
            min_cost = cost[i];
        }
    const int32_t du = min_ind % 7, dv = min_ind / 7;
    if (du == 0 || du == 6 || dv == 0 || dv == 6) return false;
    // b = At * c ; solve AtA * x = b   (Matrix::operator*: k ascending, zero-initialised sums)
    double c[9], b[6], AtA[36];
    for (int i = -1; i <= 1; i++)
        for (int j = -1; j <= 1; j++) c[(i + 1) * 3 + (j + 1)] = cost[(dv + i) * 7 + (du + j)];
    for (int r = 0; r < 6; r++) {
        b[r] = 0;
        for (int k = 0; k < 9; k++) b[r] += kFitA[k][r] * c[k];
        for (int q = 0; q < 6; q++) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += kFitA[k][r] * kFitA[k][q];
            AtA[r * 6 + q] = s;
        }
    }
    if (!gauss_jordan(AtA, b, 6)) return false;
    float divisor = (b[2] * b[2] - 4.0 * b[0] * b[1]);
    if (fabs(divisor) < 1e-8 || fabs(b[2]) < 1e-8) return false;
    float ddv = (2.0 * b[0] * b[4] - b[2] * b[3]) / divisor;
    float ddu = -(b[4] + 2.0 * b[1] * ddv) / b[2];
    if (fabs(ddu) >= 1.0 || fabs(ddv) >= 1.0) return false;
    u2 += (float)du - 3.0 + ddu;
    v2 += (float)dv - 3.0 + ddv;
    return true;
}

// Matcher::refinement   matcher.cpp:1715-1821 (1: pixel relocation, 2: parabolic fit)
void refine(orc_matcher* m, std::vector<PM>& pm, int32_t method) {
    const bool sub = m->p.refinement == 2;
    std::vector<PM> keep;
    for (size_t q = 0; q < pm.size(); q++) {
        PM it = pm[q];
        bool ok = true;
        if (method == 0 || method == 2) {
            if (sub) ok = parabolic(m, m->cur[0], m->dims_c, m->prev[0], m->dims_p, it.u1c, it.v1c, it.u1p, it.v1p);
            else relocate(m, m->cur[0], m->dims_c, m->prev[0], m->dims_p, it.u1c, it.v1c, it.u1p, it.v1p);
        }
        if (ok && (method == 1 || method == 2)) {
            if (sub) ok = parabolic(m, m->cur[0], m->dims_c, m->cur[1], m->dims_c, it.u1c, it.v1c, it.u2c, it.v2c);
            else relocate(m, m->cur[0], m->dims_c, m->cur[1], m->dims_c, it.u1c, it.v1c, it.u2c, it.v2c);
        }
        if (ok && method == 2) {
            if (sub) ok = parabolic(m, m->cur[0], m->dims_c, m->prev[1], m->dims_p, it.u1c, it.v1c, it.u2p, it.v2p);
            else relocate(m, m->cur[0], m->dims_c, m->prev[1], m->dims_p, it.u1c, it.v1c, it.u2p, it.v2p);
        }
        if (ok) keep.push_back(it);
    }
    pm.swap(keep);
}

}  // namespace

extern "C" {

void orc_matcher_params_default(svh_matcher_params* p) {
    // Matcher::parameters::parameters()   matcher.h:56-68
    p->nms_n = 3; p->nms_tau = 50; p->match_binsize = 50; p->match_radius = 200;
    p->match_disp_tolerance = 2; p->outlier_disp_tolerance = 5; p->outlier_flow_tolerance = 5;
    p->multi_stage = 1; p->half_resolution = 1; p->refinement = 1;
    p->f = p->cu = p->cv = p->base = 0;
}

orc_matcher* orc_matcher_create(const svh_matcher_params* p) {
    // Matcher::Matcher   matcher.cpp:33-63
    orc_matcher* m = new orc_matcher();
    m->p = *p;
    m->margin = 8 + 1;
    if (p->half_resolution) m->p.match_radius /= 2;
    memset(m->dims_p, 0, sizeof(m->dims_p));
    memset(m->dims_c, 0, sizeof(m->dims_c));
    m->tri_fn = 0;
    return m;
}
void orc_matcher_destroy(orc_matcher* m) { delete m; }
void orc_matcher_set_triangulator(orc_matcher* m, orc_triangulate_fn fn) { m->tri_fn = fn; }
void orc_matcher_set_intrinsics(orc_matcher* m, double f, double cu, double cv, double base) {
    m->p.f = f; m->p.cu = cu; m->p.cv = cv; m->p.base = base;
}

// M0  Matcher::pushBack   matcher.cpp:102-205
int32_t orc_matcher_push_back(orc_matcher* m, const uint8_t* I1, const uint8_t* I2, const int32_t* dims,
                              int32_t replace) {
    const int32_t w = dims[0], h = dims[1], bpl = dims[2];
    if (w <= 0 || h <= 0 || bpl < w || I1 == 0) {
        fprintf(stderr, "ERROR: Image dimension mismatch!\n");
        return 1;
    }
    if (!replace) {
        for (int k = 0; k < 2; k++) {
            m->prev[k] = View();
            std::swap(m->prev[k], m->cur[k]);
        }
        memcpy(m->dims_p, m->dims_c, sizeof(m->dims_p));
    } else {
        m->cur[0] = View();
        m->cur[1] = View();
    }
    m->dims_c[0] = w;
    m->dims_c[1] = h;
    m->dims_c[2] = w + 16 - w % 16;   // +16 even when w % 16 == 0 (matcher.cpp:173)
    const uint8_t* src[2] = {I1, I2};
    for (int k = 0; k < 2; k++) {
        if (!src[k]) continue;
        View& V = m->cur[k];
        V.I.w = w; V.I.h = h; V.I.bpl = m->dims_c[2];
        V.I.px.assign((size_t)V.I.bpl * h, 0);
        for (int32_t v = 0; v < h; v++) memcpy(&V.I.px[(size_t)v * V.I.bpl], src[k] + (size_t)v * bpl, w);
        compute_features(m, V);
    }
    return 0;
}

// Matcher::matchFeatures   matcher.cpp:209-293
int32_t orc_matcher_match_features(orc_matcher* m, int32_t method, const double* Tr) {
    const svh_matcher_params& p = m->p;
    auto missing = [&](const View& V, bool dense) {
        return !V.valid || (dense ? V.dense.empty() : V.sparse.empty());
    };
    // sanity checks: return silently, old matches stay (matcher.cpp:216-259)
    const bool need_1p = method == 0 || method == 2, need_2p = method == 2;
    const bool need_1c = true, need_2c = method == 1 || method == 2;
    for (int dense = 1; dense >= (p.multi_stage ? 0 : 1); dense--) {
        if (need_1p && missing(m->prev[0], dense)) return 0;
        if (need_2p && missing(m->prev[1], dense)) return 0;
        if (need_1c && missing(m->cur[0], dense)) return 0;
        if (need_2c && missing(m->cur[1], dense)) return 0;
    }
    for (int s = 0; s < SVH_M_STAGE_COUNT; s++) m->stage[s].clear();
    m->m1.clear();
    m->m2.clear();
    if (p.multi_stage) {
        matching(m, false, m->m1, method, false, Tr);
        m->stage[SVH_M_SPARSE_RAW] = m->m1;
        remove_outliers(m, m->m1, method);
        m->stage[SVH_M_SPARSE] = m->m1;
        prior_statistics(m, m->m1, method);
        matching(m, true, m->m2, method, true, Tr);
    } else {
        matching(m, true, m->m2, method, false, Tr);
    }
    m->stage[SVH_M_DENSE_RAW] = m->m2;
    if (p.refinement > 0) refine(m, m->m2, method);
    m->stage[SVH_M_DENSE_REFINED] = m->m2;
    remove_outliers(m, m->m2, method);
    m->stage[SVH_M_DENSE] = m->m2;
    return 0;
}

// Matcher::bucketFeatures   matcher.cpp:297-343 (same libstdc++ random_shuffle / rand())
int32_t orc_matcher_bucket_features(orc_matcher* m, int32_t max_features, float bw, float bh) {
    float u_max = 0, v_max = 0;
    for (size_t i = 0; i < m->m2.size(); i++) {
        if (m->m2[i].u1c > u_max) u_max = m->m2[i].u1c;
        if (m->m2[i].v1c > v_max) v_max = m->m2[i].v1c;
    }
    const int32_t cols = (int32_t)floor(u_max / bw) + 1, rows = (int32_t)floor(v_max / bh) + 1;
    std::vector<std::vector<PM>> b((size_t)cols * rows);
    for (size_t i = 0; i < m->m2.size(); i++) {
        int32_t u = (int32_t)floor(m->m2[i].u1c / bw), v = (int32_t)floor(m->m2[i].v1c / bh);
        b[(size_t)v * cols + u].push_back(m->m2[i]);
    }
    m->m2.clear();
    for (size_t i = 0; i < b.size(); i++) {
        std::random_shuffle(b[i].begin(), b[i].end());
        int32_t k = 0;
        for (size_t q = 0; q < b[i].size(); q++) {
            m->m2.push_back(b[i][q]);
            if (++k >= max_features) break;
        }
    }
    return (int32_t)m->m2.size();
}

// Matcher::getGain + mean   matcher.cpp:347-389, 1825-1837
float orc_matcher_get_gain(orc_matcher* m, const int32_t* inl, int32_t n) {
    if (!m->prev[0].valid || !m->cur[0].valid || m->m2.empty() || n == 0) return 1;
    auto meanf = [](const Image& I, int32_t u0, int32_t u1, int32_t v0, int32_t v1) {
        float s = 0;
        for (int32_t v = v0; v <= v1; v++)
            for (int32_t u = u0; u <= u1; u++) s += (float)I.px[(size_t)v * I.bpl + u];
        return s /= (float)((u1 - u0 + 1) * (v1 - v0 + 1));
    };
    auto cl = [](int32_t x, int32_t hi) { return std::min(std::max(x, 0), hi); };
    float gain = 0;
    int32_t num = 0;
    for (int32_t q = 0; q < n; q++) {
        if (inl[q] >= (int32_t)m->m2.size()) continue;
        const PM& it = m->m2[inl[q]];
        float mp = meanf(m->prev[0].I, cl((int32_t)it.u1p - 3, m->dims_p[0]), cl((int32_t)it.u1p + 3, m->dims_p[0]),
                         cl((int32_t)it.v1p - 3, m->dims_p[1]), cl((int32_t)it.v1p + 3, m->dims_p[1]));
        float mc = meanf(m->cur[0].I, cl((int32_t)it.u1c - 3, m->dims_p[0]), cl((int32_t)it.u1c + 3, m->dims_p[0]),
                         cl((int32_t)it.v1c - 3, m->dims_p[1]), cl((int32_t)it.v1c + 3, m->dims_p[1]));
        if (mp > 10) {
            gain += mc / mp;
            num++;
        }
    }
    return num > 0 ? gain / (float)num : 1;
}

int32_t orc_matcher_get_matches(orc_matcher* m, svh_p_match* out, int32_t cap) {
    for (int32_t i = 0; i < (int32_t)m->m2.size() && i < cap; i++) out[i] = m->m2[i];
    return (int32_t)m->m2.size();
}

int32_t orc_matcher_get_features(orc_matcher* m, int32_t table, int32_t* out, int32_t cap) {
    if (table < 0 || table > 7) return -1;
    const View& V = (table < 4 ? m->prev : m->cur)[(table >> 1) & 1];
    const std::vector<Feat>& t = (table & 1) ? V.dense : V.sparse;
    if (out) memcpy(out, t.data(), sizeof(Feat) * std::min<size_t>(t.size(), cap));
    return (int32_t)t.size();
}

int64_t orc_matcher_get_stage(orc_matcher* m, int32_t stage, void* buf, int64_t cap) {
    if (stage == SVH_M_RANGES) {
        int64_t n = (int64_t)(m->ranges.size() * sizeof(Range));
        if (buf && cap >= n && n) memcpy(buf, m->ranges.data(), n);
        return n;
    }
    if (stage < 0 || stage >= SVH_M_STAGE_COUNT) return -1;
    int64_t n = (int64_t)(m->stage[stage].size() * sizeof(PM));
    if (buf && cap >= n && n) memcpy(buf, m->stage[stage].data(), n);
    return n;
}

int64_t orc_matcher_get_filter(orc_matcher* m, int32_t which, void* buf, int64_t cap, int32_t* dims3) {
    const View& V = m->cur[0];
    if (!V.valid) return -1;
    const void* src = 0;
    int64_t n = 0;
    int32_t d[3] = {V.mw, V.mh, V.mbpl};
    switch (which) {
        case 0: src = V.du.data(); n = V.du.size(); break;
        case 1: src = V.dv.data(); n = V.dv.size(); break;
        case 2: src = V.du_full.data(); n = V.du_full.size(); d[0] = V.I.w; d[1] = V.I.h; d[2] = V.I.bpl; break;
        case 3: src = V.dv_full.data(); n = V.dv_full.size(); d[0] = V.I.w; d[1] = V.I.h; d[2] = V.I.bpl; break;
        case 4: src = V.f1.data(); n = V.f1.size() * 2; break;
        case 5: src = V.f2.data(); n = V.f2.size() * 2; break;
        default: return -1;
    }
    if (dims3) memcpy(dims3, d, 12);
    if (buf && cap >= n && n) memcpy(buf, src, n);
    return n;
}

}  // extern "C"

// ===========================================================================
// VisualOdometryStereo -- restated from
//   libviso2/src/viso_stereo.cpp:41-68 (process), :72-228 (estimateMotion),
//   :232-255 (getInlier), :259-323 (updateParameters), :327-337
//   (computeObservations), :341-485 (computeResidualsAndJacobian) and
//   libviso2/src/viso.cpp:28-37, 47-64, 68-96, 130-153.
// Plain doubles, the reference's operation order (sums run over the rows in
// ascending order, no FMA: the file is built -ffp-contract=off), libm sin/cos.
// ===========================================================================
struct orc_vo {
    svh_vo_params p;
    orc_matcher* matcher;
    double Tr[16];
    bool Tr_valid;
    std::vector<int32_t> inliers;
    std::vector<PM> matched;
};

namespace {

typedef orc_vo VoState;

enum VoResult { VO_UPDATED, VO_FAILED, VO_CONVERGED };

struct VoWork {
    const PM* pm;
    int32_t n;
    std::vector<double> X, Y, Z, J, predict, observe, residual;
};

// viso_stereo.cpp:327-337 + :341-485: observations, predictions, residuals and Jacobian rows
// of the `active` matches for the motion tr = (rx,ry,rz,tx,ty,tz)
void vo_linearise(const svh_vo_params& P, VoWork& w, const double* tr, const std::vector<int32_t>& active) {
    const double rx = tr[0], ry = tr[1], rz = tr[2], tx = tr[3], ty = tr[4], tz = tr[5];
    const double sx = sin(rx), cx = cos(rx), sy = sin(ry), cy = cos(ry), sz = sin(rz), cz = cos(rz);
    // R = Rx * Ry * Rz and its partial derivatives
    const double r00 = +cy * cz, r01 = -cy * sz, r02 = +sy;
    const double r10 = +sx * sy * cz + cx * sz, r11 = -sx * sy * sz + cx * cz, r12 = -sx * cy;
    const double r20 = -cx * sy * cz + sx * sz, r21 = +cx * sy * sz + sx * cz, r22 = +cx * cy;
    const double ax10 = +cx * sy * cz - sx * sz, ax11 = -cx * sy * sz - sx * cz, ax12 = -cx * cy;
    const double ax20 = +sx * sy * cz + cx * sz, ax21 = -sx * sy * sz + cx * cz, ax22 = -sx * cy;
    const double ay00 = -sy * cz, ay01 = +sy * sz, ay02 = +cy;
    const double ay10 = +sx * cy * cz, ay11 = -sx * cy * sz, ay12 = +sx * sy;
    const double ay20 = -cx * cy * cz, ay21 = +cx * cy * sz, ay22 = -cx * sy;
    const double az00 = -cy * sz, az01 = -cy * cz;
    const double az10 = -sx * sy * sz + cx * cz, az11 = -sx * sy * cz - cx * sz;
    const double az20 = +cx * sy * sz + sx * cz, az21 = +cx * sy * cz - sx * sz;
    const double f = P.f, cu = P.cu, cv = P.cv;
    for (size_t i = 0; i < active.size(); i++) {
        const PM& m = w.pm[active[i]];
        double* ob = &w.observe[4 * i];
        ob[0] = m.u1c; ob[1] = m.v1c; ob[2] = m.u2c; ob[3] = m.v2c;
        const double X1p = w.X[active[i]], Y1p = w.Y[active[i]], Z1p = w.Z[active[i]];
        const double X1c = r00 * X1p + r01 * Y1p + r02 * Z1p + tx;
        const double Y1c = r10 * X1p + r11 * Y1p + r12 * Z1p + ty;
        const double Z1c = r20 * X1p + r21 * Y1p + r22 * Z1p + tz;
        double weight = 1.0;
        if (P.reweighting) weight = 1.0 / (fabs(ob[0] - cu) / fabs(cu) + 0.05);
        const double X2c = X1c - P.base;
        for (int j = 0; j < 6; j++) {
            double dX = 0, dY = 0, dZ = 0;
            switch (j) {
                case 0: dX = 0;
                        dY = ax10 * X1p + ax11 * Y1p + ax12 * Z1p;
                        dZ = ax20 * X1p + ax21 * Y1p + ax22 * Z1p; break;
                case 1: dX = ay00 * X1p + ay01 * Y1p + ay02 * Z1p;
                        dY = ay10 * X1p + ay11 * Y1p + ay12 * Z1p;
                        dZ = ay20 * X1p + ay21 * Y1p + ay22 * Z1p; break;
                case 2: dX = az00 * X1p + az01 * Y1p;
                        dY = az10 * X1p + az11 * Y1p;
                        dZ = az20 * X1p + az21 * Y1p; break;
                case 3: dX = 1; break;
                case 4: dY = 1; break;
                case 5: dZ = 1; break;
            }
            double* Jr = &w.J[(4 * i) * 6 + j];
            Jr[0]  = weight * f * (dX * Z1c - X1c * dZ) / (Z1c * Z1c);
            Jr[6]  = weight * f * (dY * Z1c - Y1c * dZ) / (Z1c * Z1c);
            Jr[12] = weight * f * (dX * Z1c - X2c * dZ) / (Z1c * Z1c);
            Jr[18] = Jr[6];
        }
        double* pr = &w.predict[4 * i];
        pr[0] = f * X1c / Z1c + cu;
        pr[1] = f * Y1c / Z1c + cv;
        pr[2] = f * X2c / Z1c + cu;
        pr[3] = pr[1];
        for (int k = 0; k < 4; k++) w.residual[4 * i + k] = weight * (ob[k] - pr[k]);
    }
}

// viso_stereo.cpp:259-323: one Gauss-Newton step, normal equations by ascending-row sums
VoResult vo_update(const svh_vo_params& P, VoWork& w, const std::vector<int32_t>& active, double* tr,
                   double step, double eps) {
    if (active.size() < 3) return VO_FAILED;
    vo_linearise(P, w, tr, active);
    double A[36], B[6];
    const int32_t rows = 4 * (int32_t)active.size();
    for (int m = 0; m < 6; m++) {
        for (int n = 0; n < 6; n++) {
            double a = 0;
            for (int32_t i = 0; i < rows; i++) a += w.J[i * 6 + m] * w.J[i * 6 + n];
            A[m * 6 + n] = a;
        }
        double b = 0;
        for (int32_t i = 0; i < rows; i++) b += w.J[i * 6 + m] * w.residual[i];
        B[m] = b;
    }
    if (!gauss_jordan(A, B, 6)) return VO_FAILED;
    bool converged = true;
    for (int m = 0; m < 6; m++) {
        tr[m] += step * B[m];
        if (fabs(B[m]) > eps) converged = false;
    }
    return converged ? VO_CONVERGED : VO_UPDATED;
}

// viso_stereo.cpp:72-228.  Returns true and tr[6] on success (the reference's 6-vector), false
// for its empty vector; `inliers` is left as the reference leaves _inliers.
bool vo_estimate(VoState* s, const PM* pm, int32_t N, double* tr_out) {
    const svh_vo_params& P = s->p;
    if (N < 6) return false;               // _inliers is NOT cleared on this path (:91-94)
    VoWork w;
    w.pm = pm; w.n = N;
    w.X.resize(N); w.Y.resize(N); w.Z.resize(N);
    w.J.assign((size_t)4 * N * 6, 0.0);
    w.predict.assign((size_t)4 * N, 0.0);
    w.observe.assign((size_t)4 * N, 0.0);
    w.residual.assign((size_t)4 * N, 0.0);
    for (int32_t i = 0; i < N; i++) {
        const double d = std::max(pm[i].u1p - pm[i].u2p, 0.0001f);   // float max, then widened
        w.X[i] = (pm[i].u1p - P.cu) * P.base / d;
        w.Y[i] = (pm[i].v1p - P.cv) * P.base / d;
        w.Z[i] = P.f * P.base / d;
    }
    std::vector<double> best;     // empty until a hypothesis wins
    double cur[6];
    s->inliers.clear();
    std::vector<int32_t> all(N);
    for (int32_t i = 0; i < N; i++) all[i] = i;
    for (int32_t k = 0; k < P.ransac_iters; k++) {
        // getRandomSample(N, 3), viso.cpp:130-153: draw without replacement through libc rand()
        std::vector<int32_t> pool(all), active;
        for (int q = 0; q < 3; q++) {
            const int32_t j = rand() % (int32_t)pool.size();
            active.push_back(pool[j]);
            pool.erase(pool.begin() + j);
        }
        for (int i = 0; i < 6; i++) cur[i] = 0;
        VoResult res = VO_UPDATED;
        int32_t iter = 0;
        while (res == VO_UPDATED) {
            res = vo_update(P, w, active, cur, 1, 1e-6);
            if (iter++ > 20 || res == VO_CONVERGED) break;
        }
        if (res != VO_FAILED) {
            // getInlier, :232-255
            vo_linearise(P, w, cur, all);
            std::vector<int32_t> in;
            const double thr = P.inlier_threshold * P.inlier_threshold;
            for (int32_t i = 0; i < N; i++) {
                const double e0 = w.observe[4 * i + 0] - w.predict[4 * i + 0];
                const double e1 = w.observe[4 * i + 1] - w.predict[4 * i + 1];
                const double e2 = w.observe[4 * i + 2] - w.predict[4 * i + 2];
                const double e3 = w.observe[4 * i + 3] - w.predict[4 * i + 3];
                if (e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3 < thr) in.push_back(i);
            }
            if (in.size() > s->inliers.size()) {
                s->inliers = in;
                best.assign(cur, cur + 6);
            }
        }
    }
    bool success = true;
    if (s->inliers.size() >= 6) {
        int32_t iter = 0;
        VoResult res = VO_UPDATED;
        while (res == VO_UPDATED) {
            res = vo_update(P, w, s->inliers, best.data(), 1, 1e-8);
            if (iter++ > 100 || res == VO_CONVERGED) break;
        }
        if (res != VO_CONVERGED) success = false;
    } else {
        success = false;
    }
    if (!success) return false;
    for (int i = 0; i < 6; i++) tr_out[i] = best[i];
    return true;
}

// viso.cpp:68-96
void vo_vector_to_matrix(const double* tr, double* T) {
    const double sx = sin(tr[0]), cx = cos(tr[0]), sy = sin(tr[1]), cy = cos(tr[1]);
    const double sz = sin(tr[2]), cz = cos(tr[2]);
    T[0] = +cy * cz;                T[1] = -cy * sz;                T[2] = +sy;       T[3] = tr[3];
    T[4] = +sx * sy * cz + cx * sz; T[5] = -sx * sy * sz + cx * cz; T[6] = -sx * cy;  T[7] = tr[4];
    T[8] = -cx * sy * cz + sx * sz; T[9] = +cx * sy * sz + sx * cz; T[10] = +cx * cy; T[11] = tr[5];
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

// viso.cpp:47-64
bool vo_update_motion(VoState* s) {
    double tr[6];
    if (!vo_estimate(s, s->matched.data(), (int32_t)s->matched.size(), tr)) return false;
    vo_vector_to_matrix(tr, s->Tr);
    s->Tr_valid = true;
    return true;
}

void vo_fetch_matches(VoState* s) {
    const int32_t n = orc_matcher_get_matches(s->matcher, 0, 0);
    s->matched.resize(n);
    if (n) orc_matcher_get_matches(s->matcher, s->matched.data(), n);
}

}  // namespace

extern "C" {

void orc_vo_params_default(svh_vo_params* p) {
    orc_matcher_params_default(&p->match);
    p->bucket_max_features = 2; p->bucket_width = 50; p->bucket_height = 50;   // viso.h:50-55
    p->f = 1; p->cu = 0; p->cv = 0;                                            // viso.h:38-43
    p->base = 1.0; p->ransac_iters = 200; p->inlier_threshold = 2.0; p->reweighting = 1;
}

orc_vo* orc_vo_create(const svh_vo_params* p) {
    VoState* s = new VoState();
    s->p = *p;
    s->matcher = orc_matcher_create(&p->match);
    orc_matcher_set_intrinsics(s->matcher, p->f, p->cu, p->cv, p->base);
    for (int i = 0; i < 16; i++) s->Tr[i] = (i % 5 == 0) ? 1.0 : 0.0;
    s->Tr_valid = false;
    srand(0);                                                                  // viso.cpp:36
    return s;
}
void orc_vo_destroy(orc_vo* s) { orc_matcher_destroy(s->matcher); delete s; }
void orc_vo_set_triangulator(orc_vo* s, orc_triangulate_fn fn) { orc_matcher_set_triangulator(s->matcher, fn); }

// viso_stereo.cpp:41-68
int32_t orc_vo_process(orc_vo* s, const uint8_t* I1, const uint8_t* I2, const int32_t* dims, int32_t replace) {
    const svh_vo_params& P = s->p;
    orc_matcher_push_back(s->matcher, I1, I2, dims, replace);
    if (!s->Tr_valid) {
        orc_matcher_match_features(s->matcher, 2, 0);
        orc_matcher_bucket_features(s->matcher, P.bucket_max_features, (float)P.bucket_width, (float)P.bucket_height);
        vo_fetch_matches(s);
        vo_update_motion(s);
    }
    orc_matcher_match_features(s->matcher, 2, s->Tr_valid ? s->Tr : 0);
    orc_matcher_bucket_features(s->matcher, P.bucket_max_features, (float)P.bucket_width, (float)P.bucket_height);
    vo_fetch_matches(s);
    return vo_update_motion(s) ? 1 : 0;
}
int32_t orc_vo_process_matches(orc_vo* s, const svh_p_match* m, int32_t n) {
    s->matched.assign(m, m + n);
    return vo_update_motion(s) ? 1 : 0;
}
int32_t orc_vo_estimate_motion(orc_vo* s, const svh_p_match* m, int32_t n, double* tr6) {
    return vo_estimate(s, m, n, tr6) ? 1 : 0;
}
void orc_vo_get_motion(orc_vo* s, double* Tr16) { memcpy(Tr16, s->Tr, sizeof(s->Tr)); }
int32_t orc_vo_get_inliers(orc_vo* s, int32_t* out, int32_t cap) {
    for (int32_t i = 0; i < (int32_t)s->inliers.size() && i < cap && out; i++) out[i] = s->inliers[i];
    return (int32_t)s->inliers.size();
}
int32_t orc_vo_num_matches(orc_vo* s) { return (int32_t)s->matched.size(); }
int32_t orc_vo_get_matches(orc_vo* s, svh_p_match* out, int32_t cap) { return orc_matcher_get_matches(s->matcher, out, cap); }
float orc_vo_get_gain(orc_vo* s, const int32_t* inl, int32_t n) { return orc_matcher_get_gain(s->matcher, inl, n); }

}  // extern "C"
