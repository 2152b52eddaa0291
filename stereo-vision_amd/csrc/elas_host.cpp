// Host-resident stages of the ELAS pipeline (SURVEY 8a, E4 second half .. E9).
//
// These stages are tiny (a 249x75 lattice, ~1.5 k support points, ~3 k
// triangles), strictly serial in the reference and decide nothing about
// throughput, so they stay on the CPU between the two device phases.  What
// matters is that they reproduce the reference's observable behaviour:
//   * the consistency / redundancy filters mutate D_can in place while scanning
//     u-major (elas.cpp:174-279), and row 0 / column 0 of D_can hold the valid
//     disparity 0 left by calloc (elas.cpp:464-479);
//   * the triangulation reproduces Triangle's output order (delaunay.cpp).
// Plane fits, triangle edge lines and the disparity grid are evaluated on the
// device (k_prior / k_grid_* in elas_kernels.hip) from the support points and
// triangle lists uploaded here.  Compiled with -ffp-contract=off.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <thread>

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
    d.gwords = (p.disp_max + 1 + 31) / 32;
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
            // only "fewer than incon_min_support" matters: stop counting once reached
            int32_t cnt = 0;
            for (int32_t v2 = vlo; v2 <= vhi && cnt < p.incon_min_support; v2++) {
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

// E7  Elas::computeDelaunayTriangulation   elas.cpp:534-600, on (u,v) and (u-d,v)
// E7: the two triangulations (left coordinates, right coordinates u - d).  `parallel`: run them
// on two threads -- for the single-pair call, whose host stage is on the critical path; batch
// workers already keep every core busy and stay sequential.
bool triangulate_support(HostPrior& hp, bool parallel) {
    const int32_t n = (int32_t)(hp.support.size() / 3);
    bool ok[2] = {true, true};
    auto one = [&](int side) {
        std::vector<float> pts((size_t)2 * n);
        for (int32_t i = 0; i < n; i++) {
            pts[2 * i] = (float)(side ? hp.support[3 * i] - hp.support[3 * i + 2] : hp.support[3 * i]);
            pts[2 * i + 1] = (float)hp.support[3 * i + 1];
        }
        std::vector<int32_t>& tri = hp.tri[side];
        tri.resize((size_t)3 * (2 * n + 8));
        const int32_t nt = delaunay(pts.data(), n, tri.data(), 2 * n + 8);
        if (nt < 0) {
            ok[side] = false;
            return;
        }
        tri.resize((size_t)3 * nt);
    };
    if (parallel && n >= 256) {
        run_pair([&]() { one(0); }, [&]() { one(1); });
    } else {
        one(0);
        one(1);
    }
    return ok[0] && ok[1];
}

// prior table and plane radius (elas.cpp:984-993): float math with the float
// overloads of exp/log, as the reference resolves them
void prior_table(const svh_elas_params& p, std::vector<int32_t>& P, int32_t* plane_radius) {
    const int32_t disp_num = p.disp_max + 1;
    const float two_sigma_squared = 2 * p.sigma * p.sigma;
    P.resize(disp_num);
    for (int32_t dd = 0; dd < disp_num; dd++) {
        float tmp = -logf(p.gamma + expf(-dd * dd / two_sigma_squared)) + logf(p.gamma);
        P[dd] = (int32_t)(tmp / p.beta);
    }
    *plane_radius = (int32_t)std::max((float)ceilf(p.sigma * p.sradius), 2.0f);
}

// device bit sets -> the reference's grid layout (elas.cpp:753-775), for the tap
void expand_grid(const svh_elas_params& p, const Dims& d, const uint32_t* mask,
                 std::vector<int32_t>& grid) {
    const int32_t DN = p.disp_max + 2;
    const int32_t cells = d.gw * d.gh;
    grid.assign((size_t)cells * DN, 0);
    for (int32_t c = 0; c < cells; c++) {
        int32_t n = 0;
        for (int32_t dd = 0; dd <= p.disp_max; dd++)
            if (mask[(size_t)c * d.gwords + (dd >> 5)] >> (dd & 31) & 1u)
                grid[(size_t)c * DN + 1 + n++] = dd;
        grid[(size_t)c * DN] = n;
    }
}

}  // namespace svh
