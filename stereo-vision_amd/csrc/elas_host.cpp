// Host-resident stages of the ELAS pipeline (SURVEY 8a, E4 second half .. E9).
//
// These stages are tiny (a 249x75 lattice, ~1.5 k support points, ~3 k
// triangles), strictly serial in the reference and decide nothing about
// throughput, so they stay on the CPU between the two device phases.  What
// matters is that they reproduce the reference's observable behaviour:
//   * the consistency / redundancy filters mutate D_can in place while scanning
//     u-major (elas.cpp:174-279), and row 0 / column 0 of D_can hold the valid
//     disparity 0 left by calloc (elas.cpp:464-479);
//   * planes come from a double-precision Gauss-Jordan with full pivoting whose
//     pivot search uses ">=" (matrix.cpp:414-501);
//   * the grid dilation is a flat pointer walk that wraps columns
//     (elas.cpp:732-751);
//   * triangle edges are float lines evaluated without FMA (elas.cpp:1060-1067).
// This translation unit is compiled with -ffp-contract=off.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "svh_internal.h"

namespace svh {

Dims make_dims(const svh_elas_params& p, int32_t W, int32_t H) {
    Dims d;
    d.W = W;
    d.H = H;
    d.DW = p.subsampling ? W / 2 : W;
    d.DH = p.subsampling ? H / 2 : H;
    d.step = p.candidate_stepsize;
    if (p.subsampling) d.step += d.step % 2;
    if (d.step < 1) d.step = 1;
    d.Wc = (W + d.step - 1) / d.step;   // count of u = 0, step, 2*step, ... < W
    d.Hc = (H + d.step - 1) / d.step;
    d.gw = (int32_t)ceil((float)W / (float)p.grid_size);
    d.gh = (int32_t)ceil((float)H / (float)p.grid_size);
    return d;
}

// ---------------------------------------------------------------------------
// support points from the candidate lattice
// ---------------------------------------------------------------------------
static inline bool similar(int16_t a, int16_t b, int32_t thr) {
    return b >= 0 && abs((int)a - (int)b) <= thr;
}

void support_from_candidates(const svh_elas_params& p, const Dims& d, int16_t* dc,
                             std::vector<int32_t>& support) {
    const int32_t Wc = d.Wc, Hc = d.Hc;
    // removeInconsistentSupportPoints (elas.cpp:174-209): sequential and in place,
    // so an earlier invalidation lowers the count of a later cell.
    const int32_t ws = p.incon_window_size;
    for (int32_t uc = 0; uc < Wc; uc++) {
        const int32_t ulo = std::max(uc - ws, 0), uhi = std::min(uc + ws, Wc - 1);
        for (int32_t vc = 0; vc < Hc; vc++) {
            const int16_t dv = dc[vc * Wc + uc];
            if (dv < 0) continue;
            const int32_t vlo = std::max(vc - ws, 0), vhi = std::min(vc + ws, Hc - 1);
            int32_t cnt = 0;
            for (int32_t v2 = vlo; v2 <= vhi; v2++) {
                const int16_t* row = dc + v2 * Wc;
                for (int32_t u2 = ulo; u2 <= uhi; u2++) cnt += similar(dv, row[u2], p.incon_threshold);
            }
            if (cnt < p.incon_min_support) dc[vc * Wc + uc] = -1;
        }
    }
    // removeRedundantSupportPoints (elas.cpp:213-279), vertical then horizontal,
    // max distance 5, threshold 1 (elas.cpp:501-502)
    for (int pass = 0; pass < 2; pass++) {
        const int32_t su = pass == 0 ? 0 : 1, sv = pass == 0 ? 1 : 0;
        for (int32_t uc = 0; uc < Wc; uc++)
            for (int32_t vc = 0; vc < Hc; vc++) {
                const int16_t dv = dc[vc * Wc + uc];
                if (dv < 0) continue;
                bool both = true;
                for (int dir = -1; dir <= 1 && both; dir += 2) {
                    bool found = false;
                    for (int32_t j = 1; j <= 5; j++) {
                        int32_t u2 = uc + dir * su * j, v2 = vc + dir * sv * j;
                        if (u2 < 0 || v2 < 0 || u2 >= Wc || v2 >= Hc) break;
                        if (similar(dv, dc[v2 * Wc + u2], 1)) {
                            found = true;
                            break;
                        }
                    }
                    both = found;
                }
                if (both) dc[vc * Wc + uc] = -1;
            }
    }
    // lattice -> list, u-major from 1 (elas.cpp:505-517)
    support.clear();
    for (int32_t uc = 1; uc < Wc; uc++)
        for (int32_t vc = 1; vc < Hc; vc++) {
            int16_t dv = dc[vc * Wc + uc];
            if (dv >= 0) {
                support.push_back(uc * d.step);
                support.push_back(vc * d.step);
                support.push_back(dv);
            }
        }
    // addCornerSupportPoints (elas.cpp:283-318)
    if (p.add_corners) {
        const int32_t n = (int32_t)(support.size() / 3);
        int32_t cu[4] = {0, 0, d.W - 1, d.W - 1};
        int32_t cv[4] = {0, d.H - 1, 0, d.H - 1};
        int32_t cd[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            int32_t best = 10000000;
            for (int32_t j = 0; j < n; j++) {
                int32_t du = cu[i] - support[3 * j], dv = cv[i] - support[3 * j + 1];
                int32_t dist = du * du + dv * dv;
                if (dist < best) {
                    best = dist;
                    cd[i] = support[3 * j + 2];
                }
            }
        }
        for (int i = 0; i < 4; i++) {
            support.push_back(cu[i]);
            support.push_back(cv[i]);
            support.push_back(cd[i]);
        }
        for (int i = 2; i < 4; i++) {  // the two right-image corners
            support.push_back(cu[i] + cd[i]);
            support.push_back(cv[i]);
            support.push_back(cd[i]);
        }
    }
}

// ---------------------------------------------------------------------------
// 3x3 Gauss-Jordan with full pivoting (matrix.cpp:414-501), one RHS
// ---------------------------------------------------------------------------
static bool solve3(double A[3][3], double B[3]) {
    bool used[3] = {false, false, false};
    for (int it = 0; it < 3; it++) {
        double big = 0.0;
        int pr = 0, pc = 0;
        for (int j = 0; j < 3; j++) {
            if (used[j]) continue;
            for (int k = 0; k < 3; k++)
                if (!used[k] && fabs(A[j][k]) >= big) {  // ">=": last maximum wins
                    big = fabs(A[j][k]);
                    pr = j;
                    pc = k;
                }
        }
        used[pc] = true;
        if (pr != pc) {
            for (int l = 0; l < 3; l++) std::swap(A[pr][l], A[pc][l]);
            std::swap(B[pr], B[pc]);
        }
        if (fabs(A[pc][pc]) < 1e-20) return false;
        const double inv = 1.0 / A[pc][pc];
        A[pc][pc] = 1.0;
        for (int l = 0; l < 3; l++) A[pc][l] *= inv;
        B[pc] *= inv;
        for (int r = 0; r < 3; r++) {
            if (r == pc) continue;
            const double f = A[r][pc];
            A[r][pc] = 0.0;
            for (int l = 0; l < 3; l++) A[r][l] -= A[pc][l] * f;
            B[r] -= B[pc] * f;
        }
    }
    return true;
}

static void plane_fit(const int32_t* s0, const int32_t* s1, const int32_t* s2, bool right,
                      float out[3]) {
    const int32_t* s[3] = {s0, s1, s2};
    double A[3][3], B[3];
    for (int r = 0; r < 3; r++) {
        A[r][0] = right ? s[r][0] - s[r][2] : s[r][0];
        A[r][1] = s[r][1];
        A[r][2] = 1.0;
        B[r] = s[r][2];
    }
    if (solve3(A, B)) {
        out[0] = (float)B[0];
        out[1] = (float)B[1];
        out[2] = (float)B[2];
    } else {
        out[0] = out[1] = out[2] = 0.f;
    }
}

// rasterisation record of one triangle on one side (elas.cpp:1006-1072)
static TriRaster make_raster(const int32_t* sup, const int32_t* c, const float* pl, bool right) {
    TriRaster r;
    float tu[3], tv[3];
    for (int k = 0; k < 3; k++) {
        const int32_t* s = sup + 3 * c[k];
        tu[k] = right ? (float)(s[0] - s[2]) : (float)s[0];
        tv[k] = (float)s[1];
    }
    // the reference's exchange sort (not stable: keep the same loop shape)
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < j; k++)
            if (tu[k] > tu[j]) {
                std::swap(tu[j], tu[k]);
                std::swap(tv[j], tv[k]);
            }
    const float Au = tu[0], Av = tv[0], Bu = tu[1], Bv = tv[1], Cu = tu[2], Cv = tv[2];
    r.uA = (int32_t)Au;
    r.uB = (int32_t)Bu;
    r.uC = (int32_t)Cu;
    float ABa = 0, ACa = 0, BCa = 0;
    if (r.uA != r.uB) ABa = (Av - Bv) / (Au - Bu);
    if (r.uA != r.uC) ACa = (Av - Cv) / (Au - Cu);
    if (r.uB != r.uC) BCa = (Bv - Cv) / (Bu - Cu);
    r.ABa = ABa;
    r.ACa = ACa;
    r.BCa = BCa;
    r.ABb = Av - ABa * Au;
    r.ACb = Av - ACa * Au;
    r.BCb = Bv - BCa * Bu;
    const float pa = right ? pl[3] : pl[0];
    const float pd = right ? pl[0] : pl[3];
    r.pa = pa;
    r.pb = right ? pl[4] : pl[1];
    r.pc = right ? pl[5] : pl[2];
    r.valid = (fabs(pa) < 0.7 && fabs(pd) < 0.7) ? 1 : 0;
    return r;
}

// createGrid (elas.cpp:684-780) into a compact per-cell list
static void build_grid(const svh_elas_params& p, const Dims& d, const std::vector<int32_t>& sup,
                       bool right, std::vector<int32_t>& off, std::vector<uint16_t>& list) {
    const int32_t gw = d.gw, gh = d.gh;
    const int32_t cells = gw * gh;
    const int32_t words = (p.disp_max + 1 + 63) / 64;
    std::vector<uint64_t> seed((size_t)cells * words, 0), dil((size_t)cells * words, 0);
    const int32_t n = (int32_t)(sup.size() / 3);
    for (int32_t i = 0; i < n; i++) {
        const int32_t xc = sup[3 * i], yc = sup[3 * i + 1], dc = sup[3 * i + 2];
        int32_t x;
        if (!right) x = (int32_t)floor((float)(xc / p.grid_size));            // int division first
        else        x = (int32_t)floor((float)(xc - dc) / (float)p.grid_size);
        const int32_t y = (int32_t)floor((float)yc / (float)p.grid_size);
        if (x < 0 || x >= gw || y < 0 || y >= gh) continue;
        for (int32_t dd = std::max(dc - 1, 0); dd <= std::min(dc + 1, p.disp_max); dd++)
            seed[(size_t)(y * gw + x) * words + (dd >> 6)] |= 1ull << (dd & 63);
    }
    // flat 3x3 dilation over cells gw+1 .. cells-gw-2: columns wrap, border rows stay empty
    const int32_t nb[9] = {-gw - 1, -gw, -gw + 1, -1, 0, 1, gw - 1, gw, gw + 1};
    for (int32_t c = gw + 1; c <= cells - gw - 2; c++)
        for (int32_t w = 0; w < words; w++) {
            uint64_t o = 0;
            for (int k = 0; k < 9; k++) o |= seed[(size_t)(c + nb[k]) * words + w];
            dil[(size_t)c * words + w] = o;
        }
    off.assign(cells + 1, 0);
    list.clear();
    for (int32_t c = 0; c < cells; c++) {
        off[c] = (int32_t)list.size();
        for (int32_t w = 0; w < words; w++) {
            uint64_t b = dil[(size_t)c * words + w];
            while (b) {
                int bit = __builtin_ctzll(b);
                list.push_back((uint16_t)(w * 64 + bit));
                b &= b - 1;
            }
        }
    }
    off[cells] = (int32_t)list.size();
}

bool build_prior(const svh_elas_params& p, const Dims& d, HostPrior& hp) {
    const int32_t n = (int32_t)(hp.support.size() / 3);
    std::vector<float> pts((size_t)2 * n);
    for (int side = 0; side < 2; side++) {
        for (int32_t i = 0; i < n; i++) {
            pts[2 * i] = (float)(side ? hp.support[3 * i] - hp.support[3 * i + 2] : hp.support[3 * i]);
            pts[2 * i + 1] = (float)hp.support[3 * i + 1];
        }
        std::vector<int32_t>& tri = hp.tri[side];
        tri.resize((size_t)3 * (2 * n + 8));
        int32_t nt = delaunay(pts.data(), n, tri.data(), 2 * n + 8);
        if (nt < 0) return false;
        tri.resize((size_t)3 * nt);
        // computeDisparityPlanes (elas.cpp:605-680): both plane sets for every list
        hp.planes[side].resize((size_t)6 * nt);
        hp.raster[side].resize(nt);
        for (int32_t t = 0; t < nt; t++) {
            const int32_t* c = &tri[3 * t];
            float* pl = &hp.planes[side][6 * t];
            plane_fit(&hp.support[3 * c[0]], &hp.support[3 * c[1]], &hp.support[3 * c[2]], false, pl);
            plane_fit(&hp.support[3 * c[0]], &hp.support[3 * c[1]], &hp.support[3 * c[2]], true, pl + 3);
            hp.raster[side][t] = make_raster(hp.support.data(), c, pl, side == 1);
        }
        build_grid(p, d, hp.support, side == 1, hp.cell_off[side], hp.cell_d[side]);
    }
    // prior table and plane radius (elas.cpp:984-993), float math
    const int32_t disp_num = p.disp_max + 1;
    const float two_sigma_squared = 2 * p.sigma * p.sigma;
    hp.P.resize(disp_num);
    for (int32_t dd = 0; dd < disp_num; dd++) {
        float tmp = -logf(p.gamma + expf(-dd * dd / two_sigma_squared)) + logf(p.gamma);
        hp.P[dd] = (int32_t)(tmp / p.beta);
    }
    hp.plane_radius = (int32_t)std::max((float)ceilf(p.sigma * p.sradius), 2.0f);
    return true;
}

void expand_grid(const svh_elas_params& p, const Dims& d, const HostPrior& hp, int side,
                 std::vector<int32_t>& grid) {
    const int32_t DN = p.disp_max + 2;
    const int32_t cells = d.gw * d.gh;
    grid.assign((size_t)cells * DN, 0);
    for (int32_t c = 0; c < cells; c++) {
        const int32_t b = hp.cell_off[side][c], e = hp.cell_off[side][c + 1];
        grid[(size_t)c * DN] = e - b;
        for (int32_t i = b; i < e; i++) grid[(size_t)c * DN + 1 + (i - b)] = hp.cell_d[side][i];
    }
}

}  // namespace svh
