// TEST INFRASTRUCTURE -- scalar CPU restatement of libelas' Elas::process.
// See oracle/oracle.h for the rules; every function cites the reference lines
// it restates.  Written from the algorithm, not from the reference's SSE code:
// plain loops, explicit integer/float semantics.
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cstdio>
#include <vector>

namespace {

inline int32_t iabs(int32_t x) { return x < 0 ? -x : x; }

inline uint8_t sat_u8(int32_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// sum of |a[i]-b[i]| over 16 bytes == extract(0)+extract(4) of psadbw (elas.cpp:406-414)
inline int32_t sad16(const uint8_t* a, const uint8_t* b) {
    int32_t s = 0;
    for (int i = 0; i < 16; i++) s += iabs((int32_t)a[i] - (int32_t)b[i]);
    return s;
}

// descriptor texture: sum |desc[i]-128| (elas.cpp:358-362, 851-855)
inline int32_t texture16(const uint8_t* a) {
    int32_t s = 0;
    for (int i = 0; i < 16; i++) s += iabs((int32_t)a[i] - 128);
    return s;
}

// x86 cvttss2si semantics of (uint32_t)f followed by assignment to int32_t
// (elas.cpp:1081-1082): convert through int64 and wrap.
inline int32_t f2u2i(float f) { return (int32_t)(uint32_t)(int64_t)f; }

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------
// E1  filter::sobel3x3   libelas/src/filter.cpp:408-416 (+372-405, 227-267, 176-222)
//   column pass: S = I[v-1]+2I[v]+I[v+1], T = I[v-1]-I[v+1]           (int16)
//   row pass   : du = sat(((S[u-1]-S[u+1])>>2)+128)  ("out_v", 1 0 -1)
//                dv = sat(((T[u-1]+2T[u]+T[u+1])>>2)+128) ("out_h", 1 2 1)
//   called as sobel3x3(I, I_du, I_dv, bpl, H) (descriptor.cpp:33).
// ---------------------------------------------------------------------------
void orc_sobel3x3(const uint8_t* I, uint8_t* du, uint8_t* dv, int32_t w, int32_t h, int32_t bpl) {
    memset(du, 0, (size_t)bpl * h);
    memset(dv, 0, (size_t)bpl * h);
    std::vector<int32_t> S(w), T(w);
    for (int32_t v = 1; v < h - 1; v++) {
        const uint8_t* r0 = I + (size_t)(v - 1) * bpl;
        const uint8_t* r1 = I + (size_t)v * bpl;
        const uint8_t* r2 = I + (size_t)(v + 1) * bpl;
        for (int32_t u = 0; u < w; u++) {
            S[u] = (int32_t)r0[u] + 2 * (int32_t)r1[u] + (int32_t)r2[u];
            T[u] = (int32_t)r0[u] - (int32_t)r2[u];
        }
        for (int32_t u = 1; u < w - 1; u++) {
            du[(size_t)v * bpl + u] = sat_u8(((S[u - 1] - S[u + 1]) >> 2) + 128);
            dv[(size_t)v * bpl + u] = sat_u8(((T[u - 1] + 2 * T[u] + T[u + 1]) >> 2) + 128);
        }
    }
}

// ---------------------------------------------------------------------------
// E2  Descriptor::createDescriptor   libelas/src/descriptor.cpp:48-121
//   16 bytes per pixel, stride = width (not bpl); rows [3,h-3) (every row, or
//   even rows from 4 when half), cols [3,w-3); everything else 0.
// ---------------------------------------------------------------------------
void orc_descriptor(const uint8_t* I, int32_t w, int32_t h, int32_t bpl, int32_t half,
                    uint8_t* desc) {
    std::vector<uint8_t> du((size_t)bpl * h), dv((size_t)bpl * h);
    orc_sobel3x3(I, du.data(), dv.data(), w, h, bpl);
    memset(desc, 0, (size_t)16 * w * h);
    const int32_t v0 = half ? 4 : 3, vs = half ? 2 : 1;
    for (int32_t v = v0; v < h - 3; v += vs) {
        const uint8_t* u0 = du.data() + (size_t)(v - 2) * bpl;
        const uint8_t* u1 = du.data() + (size_t)(v - 1) * bpl;
        const uint8_t* u2 = du.data() + (size_t)v * bpl;
        const uint8_t* u3 = du.data() + (size_t)(v + 1) * bpl;
        const uint8_t* u4 = du.data() + (size_t)(v + 2) * bpl;
        const uint8_t* w1 = dv.data() + (size_t)(v - 1) * bpl;
        const uint8_t* w2 = dv.data() + (size_t)v * bpl;
        const uint8_t* w3 = dv.data() + (size_t)(v + 1) * bpl;
        for (int32_t u = 3; u < w - 3; u++) {
            uint8_t* d = desc + ((size_t)v * w + u) * 16;
            d[0] = u0[u];
            d[1] = u1[u - 2];
            d[2] = u1[u];
            d[3] = u1[u + 2];
            d[4] = u2[u - 1];
            d[5] = u2[u];
            d[6] = u2[u];
            d[7] = u2[u + 1];
            d[8] = u3[u - 2];
            d[9] = u3[u];
            d[10] = u3[u + 2];
            d[11] = u4[u];
            d[12] = w1[u];
            d[13] = w2[u - 1];
            d[14] = w2[u + 1];
            d[15] = w3[u];
        }
    }
}

// ---------------------------------------------------------------------------
// E3  Elas::computeMatchingDisparity   libelas/src/elas.cpp:322-445
// ---------------------------------------------------------------------------
static int32_t matching_disparity(const svh_elas_params* p, int32_t u, int32_t v,
                                  const uint8_t* desc1, const uint8_t* desc2, int32_t w,
                                  int32_t h, bool right_image) {
    const int32_t u_step = 2, v_step = 2, window = 3;
    if (!(u >= window + u_step && u <= w - window - 1 - u_step && v >= window + v_step &&
          v <= h - window - 1 - v_step))
        return -1;
    const uint8_t* own = right_image ? desc2 : desc1;
    const uint8_t* oth = right_image ? desc1 : desc2;
    const size_t row = (size_t)16 * w;
    const uint8_t* own_px = own + row * v + (size_t)16 * u;
    if (texture16(own_px) < p->support_texture) return -1;

    int32_t dmin = std::max(p->disp_min, 0);
    int32_t dmax = right_image ? std::min(p->disp_max, w - u - window - u_step)
                               : std::min(p->disp_max, u - window - u_step);
    if (dmax - dmin < 10) return -1;

    const ptrdiff_t off[4] = {-16 * u_step - (ptrdiff_t)row * v_step, +16 * u_step - (ptrdiff_t)row * v_step,
                              -16 * u_step + (ptrdiff_t)row * v_step, +16 * u_step + (ptrdiff_t)row * v_step};
    int32_t e1 = 32767, d1 = -1, e2 = 32767, d2 = -1;
    for (int32_t d = dmin; d <= dmax; d++) {
        int32_t uw = right_image ? u + d : u - d;
        const uint8_t* oth_px = oth + row * v + (size_t)16 * uw;
        int32_t e = 0;
        for (int k = 0; k < 4; k++) e += sad16(own_px + off[k], oth_px + off[k]);
        if (e < e1) {
            e2 = e1;
            d2 = d1;
            e1 = e;
            d1 = d;
        } else if (e < e2) {
            e2 = e;
            d2 = d;
        }
    }
    if (d1 >= 0 && d2 >= 0 && (float)e1 < p->support_threshold * (float)e2) return d1;
    return -1;
}

void orc_dcan_dims(const svh_elas_params* p, int32_t w, int32_t h, int32_t* wc, int32_t* hc) {
    // elas.cpp:453-463
    int32_t step = p->candidate_stepsize;
    if (p->subsampling) step += step % 2;
    int32_t a = 0, b = 0;
    for (int32_t u = 0; u < w; u += step) a++;
    for (int32_t v = 0; v < h; v += step) b++;
    *wc = a;
    *hc = b;
}

// E4 (first half)  Elas::computeSupportMatches   elas.cpp:449-493
//   D_can is calloc'ed: row 0 / column 0 keep the *valid* disparity 0.
void orc_support_candidates(const svh_elas_params* p, const uint8_t* desc1, const uint8_t* desc2,
                            int32_t w, int32_t h, int16_t* dcan) {
    int32_t step = p->candidate_stepsize;
    if (p->subsampling) step += step % 2;
    int32_t wc, hc;
    orc_dcan_dims(p, w, h, &wc, &hc);
    memset(dcan, 0, sizeof(int16_t) * wc * hc);
    for (int32_t uc = 1; uc < wc; uc++) {
        int32_t u = uc * step;
        for (int32_t vc = 1; vc < hc; vc++) {
            int32_t v = vc * step;
            int16_t out = -1;
            int32_t d = matching_disparity(p, u, v, desc1, desc2, w, h, false);
            if (d >= 0) {
                int32_t d2 = matching_disparity(p, u - d, v, desc1, desc2, w, h, true);
                if (d2 >= 0 && iabs(d - d2) <= p->lr_threshold) out = (int16_t)d;
            }
            dcan[vc * wc + uc] = out;
        }
    }
}

// E5  Elas::removeInconsistentSupportPoints   elas.cpp:174-209 (in place, u-major)
static void remove_inconsistent(const svh_elas_params* p, int16_t* dc, int32_t wc, int32_t hc) {
    const int32_t ws = p->incon_window_size;
    for (int32_t uc = 0; uc < wc; uc++)
        for (int32_t vc = 0; vc < hc; vc++) {
            int16_t d = dc[vc * wc + uc];
            if (d < 0) continue;
            int32_t support = 0;
            for (int32_t u2 = uc - ws; u2 <= uc + ws; u2++)
                for (int32_t v2 = vc - ws; v2 <= vc + ws; v2++)
                    if (u2 >= 0 && v2 >= 0 && u2 < wc && v2 < hc) {
                        int16_t d2 = dc[v2 * wc + u2];
                        if (d2 >= 0 && iabs(d - d2) <= p->incon_threshold) support++;
                    }
            if (support < p->incon_min_support) dc[vc * wc + uc] = -1;
        }
}

// E6  Elas::removeRedundantSupportPoints   elas.cpp:213-279 (in place, u-major)
static void remove_redundant(int16_t* dc, int32_t wc, int32_t hc, int32_t max_dist,
                             int32_t thresh, bool vertical) {
    const int32_t du[2] = {vertical ? 0 : -1, vertical ? 0 : +1};
    const int32_t dv[2] = {vertical ? -1 : 0, vertical ? +1 : 0};
    for (int32_t uc = 0; uc < wc; uc++)
        for (int32_t vc = 0; vc < hc; vc++) {
            int16_t d = dc[vc * wc + uc];
            if (d < 0) continue;
            bool redundant = true;
            for (int i = 0; i < 2 && redundant; i++) {
                int32_t u2 = uc, v2 = vc;
                bool support = false;
                for (int32_t j = 0; j < max_dist; j++) {
                    u2 += du[i];
                    v2 += dv[i];
                    if (u2 < 0 || v2 < 0 || u2 >= wc || v2 >= hc) break;
                    int16_t d2 = dc[v2 * wc + u2];
                    if (d2 >= 0 && iabs(d - d2) <= thresh) {
                        support = true;
                        break;
                    }
                }
                if (!support) redundant = false;
            }
            if (redundant) dc[vc * wc + uc] = -1;
        }
}

// E4 (second half)  elas.cpp:495-523 + addCornerSupportPoints elas.cpp:283-318
int32_t orc_support_filter(const svh_elas_params* p, int16_t* dcan, int32_t w, int32_t h,
                           int32_t* support, int32_t cap) {
    int32_t step = p->candidate_stepsize;
    if (p->subsampling) step += step % 2;
    int32_t wc, hc;
    orc_dcan_dims(p, w, h, &wc, &hc);
    remove_inconsistent(p, dcan, wc, hc);
    remove_redundant(dcan, wc, hc, 5, 1, true);
    remove_redundant(dcan, wc, hc, 5, 1, false);
    std::vector<int32_t> s;
    for (int32_t uc = 1; uc < wc; uc++)
        for (int32_t vc = 1; vc < hc; vc++)
            if (dcan[vc * wc + uc] >= 0) {
                s.push_back(uc * step);
                s.push_back(vc * step);
                s.push_back(dcan[vc * wc + uc]);
            }
    if (p->add_corners) {
        int32_t b[6][3] = {{0, 0, 0}, {0, h - 1, 0}, {w - 1, 0, 0}, {w - 1, h - 1, 0}, {0, 0, 0}, {0, 0, 0}};
        const size_t n = s.size() / 3;
        for (int i = 0; i < 4; i++) {
            int32_t best = 10000000;
            for (size_t j = 0; j < n; j++) {
                int32_t du = b[i][0] - s[3 * j], dv = b[i][1] - s[3 * j + 1];
                int32_t dist = du * du + dv * dv;
                if (dist < best) {
                    best = dist;
                    b[i][2] = s[3 * j + 2];
                }
            }
        }
        for (int i = 0; i < 2; i++) {
            b[4 + i][0] = b[2 + i][0] + b[2 + i][2];
            b[4 + i][1] = b[2 + i][1];
            b[4 + i][2] = b[2 + i][2];
        }
        for (int i = 0; i < 6; i++)
            for (int k = 0; k < 3; k++) s.push_back(b[i][k]);
    }
    int32_t n = (int32_t)(s.size() / 3);
    for (int32_t i = 0; i < n && i < cap; i++)
        for (int k = 0; k < 3; k++) support[3 * i + k] = s[3 * i + k];
    return n;
}

// ---------------------------------------------------------------------------
// Matrix::solve for a 3x3 system with one right-hand side
//   libelas/src/matrix.cpp:414-501 (Gauss-Jordan, full pivoting, eps 1e-20,
//   ">=" in the pivot search so the LAST maximal element wins)
// ---------------------------------------------------------------------------
static bool gauss_jordan3(double A[3][3], double B[3]) {
    const double eps = 1e-20;
    int ipiv[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) {
        double big = 0.0;
        int irow = 0, icol = 0;
        for (int j = 0; j < 3; j++)
            if (ipiv[j] != 1)
                for (int k = 0; k < 3; k++)
                    if (ipiv[k] == 0)
                        if (fabs(A[j][k]) >= big) {
                            big = fabs(A[j][k]);
                            irow = j;
                            icol = k;
                        }
        ++ipiv[icol];
        if (irow != icol) {
            for (int l = 0; l < 3; l++) std::swap(A[irow][l], A[icol][l]);
            std::swap(B[irow], B[icol]);
        }
        if (fabs(A[icol][icol]) < eps) return false;
        double pivinv = 1.0 / A[icol][icol];
        A[icol][icol] = 1.0;
        for (int l = 0; l < 3; l++) A[icol][l] *= pivinv;
        B[icol] *= pivinv;
        for (int ll = 0; ll < 3; ll++)
            if (ll != icol) {
                double dum = A[ll][icol];
                A[ll][icol] = 0.0;
                for (int l = 0; l < 3; l++) A[ll][l] -= A[icol][l] * dum;
                B[ll] -= B[icol] * dum;
            }
    }
    return true;  // the column unscramble (matrix.cpp:487-493) only touches A
}

// E8  Elas::computeDisparityPlanes   elas.cpp:605-680
void orc_planes(const int32_t* sup, const int32_t* tri, int32_t ntri, float* planes) {
    for (int32_t i = 0; i < ntri; i++) {
        const int32_t c[3] = {tri[3 * i], tri[3 * i + 1], tri[3 * i + 2]};
        for (int side = 0; side < 2; side++) {
            double A[3][3], B[3];
            for (int r = 0; r < 3; r++) {
                int32_t u = sup[3 * c[r]], v = sup[3 * c[r] + 1], d = sup[3 * c[r] + 2];
                A[r][0] = side == 0 ? u : u - d;
                A[r][1] = v;
                A[r][2] = 1;
                B[r] = d;
            }
            float* o = planes + 6 * i + 3 * side;
            if (gauss_jordan3(A, B)) {
                o[0] = (float)B[0];
                o[1] = (float)B[1];
                o[2] = (float)B[2];
            } else {
                o[0] = o[1] = o[2] = 0.f;
            }
        }
    }
}

void orc_grid_dims(const svh_elas_params* p, int32_t w, int32_t h, int32_t* gw, int32_t* gh) {
    // elas.cpp:98-99
    *gw = (int32_t)ceil((float)w / (float)p->grid_size);
    *gh = (int32_t)ceil((float)h / (float)p->grid_size);
}

// E9  Elas::createGrid   elas.cpp:684-780
void orc_grid(const svh_elas_params* p, const int32_t* sup, int32_t nsup, int32_t w, int32_t h,
              int32_t right, int32_t* grid) {
    int32_t gw, gh;
    orc_grid_dims(p, w, h, &gw, &gh);
    const int32_t D = p->disp_max + 1;  // entries per cell in temp
    const size_t cells = (size_t)gw * gh;
    std::vector<uint8_t> t1(cells * D, 0), t2(cells * D, 0);
    for (int32_t i = 0; i < nsup; i++) {
        int32_t xc = sup[3 * i], yc = sup[3 * i + 1], dc = sup[3 * i + 2];
        int32_t dlo = std::max(dc - 1, 0), dhi = std::min(dc + 1, p->disp_max);
        for (int32_t d = dlo; d <= dhi; d++) {
            int32_t x;
            if (!right) x = (int32_t)floor((float)(xc / p->grid_size));  // integer division first
            else        x = (int32_t)floor((float)(xc - dc) / (float)p->grid_size);
            int32_t y = (int32_t)floor((float)yc / (float)p->grid_size);
            if (x >= 0 && x < gw && y >= 0 && y < gh) t1[((size_t)y * gw + x) * D + d] = 1;
        }
    }
    // 3x3 dilation done as a flat walk over cells gw+1 .. cells-gw-2 (elas.cpp:732-751):
    // columns wrap around, first/last grid rows stay empty.
    const ptrdiff_t nb[9] = {-gw - 1, -gw, -gw + 1, -1, 0, 1, gw - 1, gw, gw + 1};
    for (ptrdiff_t c = gw + 1; c <= (ptrdiff_t)cells - gw - 2; c++)
        for (int32_t d = 0; d < D; d++) {
            uint8_t o = 0;
            for (int k = 0; k < 9; k++) o |= t1[(size_t)(c + nb[k]) * D + d];
            t2[(size_t)c * D + d] = o;
        }
    const int32_t DN = p->disp_max + 2;
    memset(grid, 0, sizeof(int32_t) * cells * DN);
    for (int32_t x = 0; x < gw; x++)
        for (int32_t y = 0; y < gh; y++) {
            int32_t* cell = grid + ((size_t)y * gw + x) * DN;
            int32_t n = 0;
            for (int32_t d = 0; d <= p->disp_max; d++)
                if (t2[((size_t)y * gw + x) * D + d]) cell[1 + n++] = d;
            cell[0] = n;
        }
}

// ---------------------------------------------------------------------------
// E11  Elas::findMatch + updatePosteriorMinimum   elas.cpp:784-955
// ---------------------------------------------------------------------------
static void find_match(const svh_elas_params* p, int32_t u, int32_t v, float pa, float pb, float pc,
                       const int32_t* grid, int32_t gw, const uint8_t* desc1, const uint8_t* desc2,
                       const int32_t* P, int32_t plane_radius, bool valid, bool right_image,
                       int32_t w, int32_t h, float* D) {
    const int32_t disp_num = p->disp_max + 1;
    const int32_t window = 2;
    size_t d_addr = p->subsampling ? (size_t)(v / 2) * (w / 2) + u / 2 : (size_t)v * w + u;
    if (u < window || u >= w - window) return;
    const size_t row = (size_t)16 * w;
    const size_t line = row * (size_t)std::max(std::min(v, h - 3), 2);
    const uint8_t* own_line = (right_image ? desc2 : desc1) + line;
    const uint8_t* oth_line = (right_image ? desc1 : desc2) + line;
    const uint8_t* own_px = own_line + (size_t)16 * u;
    if (texture16(own_px) < p->match_texture) return;

    int32_t d_plane = (int32_t)(pa * (float)u + pb * (float)v + pc);
    int32_t d_plane_min = std::max(d_plane - plane_radius, 0);
    int32_t d_plane_max = std::min(d_plane + plane_radius, disp_num - 1);

    int32_t gx = (int32_t)floor((float)u / (float)p->grid_size);
    int32_t gy = (int32_t)floor((float)v / (float)p->grid_size);
    const int32_t* cell = grid + ((size_t)gy * gw + gx) * (p->disp_max + 2);
    int32_t num_grid = cell[0];
    const int32_t* d_grid = cell + 1;

    int32_t min_val = 10000, min_d = -1;
    for (int32_t i = 0; i < num_grid; i++) {
        int32_t dc = d_grid[i];
        if (dc < d_plane_min || dc > d_plane_max) {
            int32_t uw = right_image ? u + dc : u - dc;
            if (uw < window || uw >= w - window) continue;
            int32_t val = sad16(own_px, oth_line + (size_t)16 * uw);
            if (val < min_val) {
                min_val = val;
                min_d = dc;
            }
        }
    }
    for (int32_t dc = d_plane_min; dc <= d_plane_max; dc++) {
        int32_t uw = right_image ? u + dc : u - dc;
        if (uw < window || uw >= w - window) continue;
        int32_t val = sad16(own_px, oth_line + (size_t)16 * uw) + (valid ? P[iabs(dc - d_plane)] : 0);
        if (val < min_val) {
            min_val = val;
            min_d = dc;
        }
    }
    D[d_addr] = min_d >= 0 ? (float)min_d : -1.f;
}

// E10  Elas::computeDisparity   elas.cpp:960-1118
void orc_dense(const svh_elas_params* p, const int32_t* sup, const int32_t* tri,
               const float* planes, int32_t ntri, const int32_t* grid, const uint8_t* desc1,
               const uint8_t* desc2, int32_t w, int32_t h, int32_t right, float* D) {
    const int32_t disp_num = p->disp_max + 1;
    int32_t gw, gh;
    orc_grid_dims(p, w, h, &gw, &gh);
    const size_t dn = p->subsampling ? (size_t)(w / 2) * (h / 2) : (size_t)w * h;
    for (size_t i = 0; i < dn; i++) D[i] = -10.f;

    // prior table, float arithmetic with the float overloads of exp/log (elas.cpp:984-992)
    const float two_sigma_squared = 2 * p->sigma * p->sigma;
    std::vector<int32_t> P(disp_num);
    for (int32_t dd = 0; dd < disp_num; dd++) {
        float tmp = -logf(p->gamma + expf(-dd * dd / two_sigma_squared)) + logf(p->gamma);
        P[dd] = (int32_t)(tmp / p->beta);
    }
    int32_t plane_radius = (int32_t)std::max((float)ceilf(p->sigma * p->sradius), 2.0f);

    const bool rimg = right != 0;
    for (int32_t i = 0; i < ntri; i++) {
        const float* pl = planes + 6 * i;
        float pa, pb, pc, pd;
        if (!rimg) { pa = pl[0]; pb = pl[1]; pc = pl[2]; pd = pl[3]; }
        else       { pa = pl[3]; pb = pl[4]; pc = pl[5]; pd = pl[0]; }
        float tu[3], tv[3];
        for (int k = 0; k < 3; k++) {
            const int32_t* s = sup + 3 * tri[3 * i + k];
            tu[k] = rimg ? (float)(s[0] - s[2]) : (float)s[0];
            tv[k] = (float)s[1];
        }
        for (uint32_t j = 0; j < 3; j++)
            for (uint32_t k = 0; k < j; k++)
                if (tu[k] > tu[j]) {
                    std::swap(tu[j], tu[k]);
                    std::swap(tv[j], tv[k]);
                }
        float Au = tu[0], Av = tv[0], Bu = tu[1], Bv = tv[1], Cu = tu[2], Cv = tv[2];
        float ABa = 0, ACa = 0, BCa = 0;
        if ((int32_t)Au != (int32_t)Bu) ABa = (Av - Bv) / (Au - Bu);
        if ((int32_t)Au != (int32_t)Cu) ACa = (Av - Cv) / (Au - Cu);
        if ((int32_t)Bu != (int32_t)Cu) BCa = (Bv - Cv) / (Bu - Cu);
        float ABb = Av - ABa * Au;
        float ACb = Av - ACa * Au;
        float BCb = Bv - BCa * Bu;
        bool valid = fabs(pa) < 0.7 && fabs(pd) < 0.7;

        for (int part = 0; part < 2; part++) {
            float lo = part == 0 ? Au : Bu, hi = part == 0 ? Bu : Cu;
            float ea = part == 0 ? ABa : BCa, eb = part == 0 ? ABb : BCb;
            if ((int32_t)lo == (int32_t)hi) continue;
            for (int32_t u = std::max((int32_t)lo, 0); u < std::min((int32_t)hi, w); u++) {
                if (p->subsampling && u % 2 != 0) continue;
                int32_t v1 = f2u2i(ACa * (float)u + ACb);
                int32_t v2 = f2u2i(ea * (float)u + eb);
                for (int32_t v = std::min(v1, v2); v < std::max(v1, v2); v++) {
                    if (p->subsampling && v % 2 != 0) continue;
                    find_match(p, u, v, pa, pb, pc, grid, gw, desc1, desc2, P.data(), plane_radius,
                               valid, rimg, w, h, D);
                }
            }
        }
    }
}

// E12  Elas::leftRightConsistencyCheck   elas.cpp:1122-1204
void orc_lr_check(const svh_elas_params* p, float* D1, float* D2, int32_t dw, int32_t dh) {
    std::vector<float> c1(D1, D1 + (size_t)dw * dh), c2(D2, D2 + (size_t)dw * dh);
    for (int32_t u = 0; u < dw; u++)
        for (int32_t v = 0; v < dh; v++) {
            size_t a = (size_t)v * dw + u;
            float d1 = c1[a], d2 = c2[a];
            float uw1, uw2;
            if (p->subsampling) { uw1 = (float)u - d1 / 2; uw2 = (float)u + d2 / 2; }
            else                { uw1 = (float)u - d1;     uw2 = (float)u + d2; }
            if (d1 >= 0 && uw1 >= 0 && uw1 < dw) {
                size_t aw = (size_t)v * dw + (int32_t)uw1;
                if (fabs(c2[aw] - d1) > p->lr_threshold) D1[a] = -10;
            } else D1[a] = -10;
            if (d2 >= 0 && uw2 >= 0 && uw2 < dw) {
                size_t aw = (size_t)v * dw + (int32_t)uw2;
                if (fabs(c1[aw] - d2) > p->lr_threshold) D2[a] = -10;
            } else D2[a] = -10;
        }
}

// E13  Elas::removeSmallSegments   elas.cpp:1208-1326
void orc_remove_small_segments(const svh_elas_params* p, float* D, int32_t dw, int32_t dh) {
    int32_t min_size = p->speckle_size;
    if (p->subsampling) min_size = (int32_t)(sqrt((float)p->speckle_size) * 2);
    const size_t n = (size_t)dw * dh;
    std::vector<uint8_t> done(n, 0);
    std::vector<int32_t> list(n);
    for (int32_t u = 0; u < dw; u++)
        for (int32_t v = 0; v < dh; v++) {
            size_t start = (size_t)v * dw + u;
            if (done[start]) continue;
            size_t count = 1, curr = 0;
            list[0] = (int32_t)start;
            while (curr < count) {
                int32_t a = list[curr];
                int32_t cu = a % dw, cv = a / dw;
                const int32_t nu[4] = {cu - 1, cu + 1, cu, cu};
                const int32_t nv[4] = {cv, cv, cv - 1, cv + 1};
                for (int i = 0; i < 4; i++)
                    if (nu[i] >= 0 && nv[i] >= 0 && nu[i] < dw && nv[i] < dh) {
                        size_t an = (size_t)nv[i] * dw + nu[i];
                        if (!done[an] && D[an] >= 0)
                            if (fabs(D[a] - D[an]) <= p->speckle_sim_threshold) {
                                list[count++] = (int32_t)an;
                                done[an] = 1;
                            }
                    }
                curr++;
                done[a] = 1;
            }
            if ((int64_t)count < min_size)
                for (size_t i = 0; i < count; i++) D[list[i]] = -10;
        }
}

// E14  Elas::gapInterpolation   elas.cpp:1330-1530
void orc_gap_interpolation(const svh_elas_params* p, float* D, int32_t dw, int32_t dh) {
    int32_t gap = p->ipol_gap_width;
    if (p->subsampling) gap = p->ipol_gap_width / 2 + 1;
    const float discon = 3.0f;
    for (int32_t v = 0; v < dh; v++) {
        int32_t count = 0;
        for (int32_t u = 0; u < dw; u++) {
            if (D[(size_t)v * dw + u] >= 0) {
                if (count >= 1 && count <= gap) {
                    int32_t first = u - count, last = u - 1;
                    if (first > 0 && last < dw - 1) {
                        float d1 = D[(size_t)v * dw + first - 1], d2 = D[(size_t)v * dw + last + 1];
                        float di = fabs(d1 - d2) < discon ? (d1 + d2) / 2 : std::min(d1, d2);
                        for (int32_t uc = first; uc <= last; uc++) D[(size_t)v * dw + uc] = di;
                    }
                }
                count = 0;
            } else count++;
        }
        if (p->add_corners) {
            for (int32_t u = 0; u < dw; u++)
                if (D[(size_t)v * dw + u] >= 0) {
                    for (int32_t u2 = std::max(u - gap, 0); u2 < u; u2++) D[(size_t)v * dw + u2] = D[(size_t)v * dw + u];
                    break;
                }
            for (int32_t u = dw - 1; u >= 0; u--)
                if (D[(size_t)v * dw + u] >= 0) {
                    for (int32_t u2 = u; u2 <= std::min(u + gap, dw - 1); u2++) D[(size_t)v * dw + u2] = D[(size_t)v * dw + u];
                    break;
                }
        }
    }
    for (int32_t u = 0; u < dw; u++) {
        int32_t count = 0;
        for (int32_t v = 0; v < dh; v++) {
            if (D[(size_t)v * dw + u] >= 0) {
                if (count >= 1 && count <= gap) {
                    int32_t first = v - count, last = v - 1;
                    if (first > 0 && last < dh - 1) {
                        float d1 = D[(size_t)(first - 1) * dw + u], d2 = D[(size_t)(last + 1) * dw + u];
                        float di = fabs(d1 - d2) < discon ? (d1 + d2) / 2 : std::min(d1, d2);
                        for (int32_t vc = first; vc <= last; vc++) D[(size_t)vc * dw + u] = di;
                    }
                }
                count = 0;
            } else count++;
        }
        if (p->add_corners) {
            for (int32_t v = 0; v < dh; v++)
                if (D[(size_t)v * dw + u] >= 0) {
                    for (int32_t v2 = std::max(v - gap, 0); v2 < v; v2++) D[(size_t)v2 * dw + u] = D[(size_t)v * dw + u];
                    break;
                }
            for (int32_t v = dh - 1; v >= 0; v--)
                if (D[(size_t)v * dw + u] >= 0) {
                    for (int32_t v2 = v; v2 <= std::min(v + gap, dh - 1); v2++) D[(size_t)v2 * dw + u] = D[(size_t)v * dw + u];
                    break;
                }
        }
    }
}

// ---------------------------------------------------------------------------
// E15  Elas::adaptiveMean   elas.cpp:1535-1754
//   The "abs mask" is _mm_set1_ps(0x7FFFFFFF): an int->float conversion whose
//   bit pattern is 0x4F000000 (elas.cpp:1571).  weight = max(0, 4 - (bits(val -
//   val_c) & 0x4F000000)); lanes are ring slots (position % taps); the 8-tap
//   branch adds slot j and j+4 first, then sums ((l0+l1)+l2)+l3.
// ---------------------------------------------------------------------------
static inline float am_weight(float val, float centre) {
    float diff = val - centre;
    uint32_t b;
    memcpy(&b, &diff, 4);
    b &= 0x4F000000u;
    float m;
    memcpy(&m, &b, 4);
    float wgt = 4.0f - m;
    return wgt > 0.0f ? wgt : 0.0f;  // _mm_max_ps(0, w)
}

static inline bool am_filter(const float* ring, int taps, float centre, float* out) {
    float wl[4], fl[4];
    for (int j = 0; j < 4; j++) {
        float w0 = am_weight(ring[j], centre);
        float f0 = ring[j] * w0;
        if (taps == 8) {
            float w1 = am_weight(ring[j + 4], centre);
            float f1 = ring[j + 4] * w1;
            wl[j] = w0 + w1;
            fl[j] = f0 + f1;
        } else {
            wl[j] = w0;
            fl[j] = f0;
        }
    }
    float ws = wl[0] + wl[1] + wl[2] + wl[3];
    float fs = fl[0] + fl[1] + fl[2] + fl[3];
    if (ws > 0) {
        float d = fs / ws;
        if (d >= 0) {
            *out = d;
            return true;
        }
    }
    return false;
}

void orc_adaptive_mean(const svh_elas_params* p, float* D, int32_t dw, int32_t dh) {
    const size_t n = (size_t)dw * dh;
    std::vector<float> copy(D, D + n);
    // D_tmp is malloc'ed, set to -10 where D is invalid and otherwise only written by the
    // horizontal pass (elas.cpp:1548-1560, 1683-1701).  It is a fresh multi-MB block (zero
    // pages) and the pinned reference (oracle/_ref, ref_init(1)) zero-fills its allocations, so
    // the never-written valid pixels -- the 3 border rows / columns, reachable only with
    // add_corners -- read as 0 in the vertical pass.
    std::vector<float> tmp(n, 0.0f);
    for (size_t i = 0; i < n; i++)
        if (D[i] < 0) {
            copy[i] = -10;
            tmp[i] = -10;
        }
    const int taps = p->subsampling ? 4 : 8;
    const int lead = taps - 1;          // 3 or 7: first index at which the ring is full
    const int back = p->subsampling ? 1 : 3;  // centre = u - back
    float ring[8];
    for (int32_t v = 3; v < dh - 3; v++) {
        for (int32_t u = 0; u < lead; u++) ring[u] = copy[(size_t)v * dw + u];
        for (int32_t u = lead; u < dw; u++) {
            float centre = copy[(size_t)v * dw + (u - back)];
            ring[u % taps] = copy[(size_t)v * dw + u];
            float d;
            if (am_filter(ring, taps, centre, &d)) tmp[(size_t)v * dw + (u - back)] = d;
        }
    }
    for (int32_t u = 3; u < dw - 3; u++) {
        for (int32_t v = 0; v < lead; v++) ring[v] = tmp[(size_t)v * dw + u];
        for (int32_t v = lead; v < dh; v++) {
            float centre = tmp[(size_t)(v - back) * dw + u];
            ring[v % taps] = tmp[(size_t)v * dw + u];
            float d;
            if (am_filter(ring, taps, centre, &d)) D[(size_t)(v - back) * dw + u] = d;
        }
    }
}

// E16  Elas::median   elas.cpp:1758-1838 (separable 7-tap, insertion sorted)
void orc_median(const svh_elas_params* p, float* D, int32_t dw, int32_t dh) {
    (void)p;
    const int32_t ws = 3;
    std::vector<float> T((size_t)dw * dh, 0.f);
    float vals[7];
    auto med = [&](const float* src, ptrdiff_t stride) {
        int j = 0;
        for (int k = -ws; k <= ws; k++) {
            float t = src[k * stride];
            int i = j - 1;
            while (i >= 0 && vals[i] > t) {
                vals[i + 1] = vals[i];
                i--;
            }
            vals[i + 1] = t;
            j++;
        }
        return vals[ws];
    };
    for (int32_t u = ws; u < dw - ws; u++)
        for (int32_t v = ws; v < dh - ws; v++) {
            size_t a = (size_t)v * dw + u;
            T[a] = D[a] >= 0 ? med(D + a, 1) : D[a];
        }
    for (int32_t u = ws; u < dw - ws; u++)
        for (int32_t v = ws; v < dh - ws; v++) {
            size_t a = (size_t)v * dw + u;
            if (D[a] >= 0) D[a] = med(T.data() + a, dw);
        }
}

// ---------------------------------------------------------------------------
// Elas::process   elas.cpp:32-170
// ---------------------------------------------------------------------------
struct orc_run {
    int32_t status;
    std::vector<uint8_t> desc1, desc2;
    std::vector<int16_t> dcan_raw;
    std::vector<int32_t> support, tri1, tri2, grid1, grid2;
    std::vector<float> planes1, planes2;
    std::vector<float> d1_raw, d2_raw, d1_lr, d2_lr, d1_seg, d2_seg, d1_gap, d2_gap, d1, d2;
};

orc_run* orc_elas_run(const svh_elas_params* p, const uint8_t* I1_, const uint8_t* I2_,
                      const int32_t* dims, orc_triangulate_fn tri_fn) {
    orc_run* r = new orc_run();
    const int32_t W = dims[0], H = dims[1];
    const int32_t bpl = W + 15 - (W - 1) % 16;  // elas.cpp:37
    std::vector<uint8_t> I1((size_t)bpl * H, 0), I2((size_t)bpl * H, 0);
    for (int32_t v = 0; v < H; v++) {
        memcpy(&I1[(size_t)v * bpl], I1_ + (size_t)v * dims[2], W);
        memcpy(&I2[(size_t)v * bpl], I2_ + (size_t)v * dims[2], W);
    }
    r->desc1.resize((size_t)16 * W * H);
    r->desc2.resize((size_t)16 * W * H);
    orc_descriptor(I1.data(), W, H, bpl, p->subsampling, r->desc1.data());
    orc_descriptor(I2.data(), W, H, bpl, p->subsampling, r->desc2.data());

    int32_t wc, hc;
    orc_dcan_dims(p, W, H, &wc, &hc);
    r->dcan_raw.resize((size_t)wc * hc);
    orc_support_candidates(p, r->desc1.data(), r->desc2.data(), W, H, r->dcan_raw.data());
    std::vector<int16_t> dcan = r->dcan_raw;
    r->support.resize((size_t)3 * (wc * hc + 6));
    int32_t ns = orc_support_filter(p, dcan.data(), W, H, r->support.data(), wc * hc + 6);
    r->support.resize((size_t)3 * ns);
    if (ns < 3) {
        printf("ERROR: Need at least 3 support points!\n");
        r->status = 1;
        return r;
    }
    r->status = 0;
    // Delaunay on (u,v) and (u-d,v)  (elas.cpp:534-600) -- supplied triangulator
    for (int side = 0; side < 2; side++) {
        std::vector<float> pts((size_t)2 * ns);
        for (int32_t i = 0; i < ns; i++) {
            pts[2 * i] = (float)(side ? r->support[3 * i] - r->support[3 * i + 2] : r->support[3 * i]);
            pts[2 * i + 1] = (float)r->support[3 * i + 1];
        }
        std::vector<int32_t>& t = side ? r->tri2 : r->tri1;
        t.resize((size_t)3 * (2 * ns + 16));
        int32_t nt = tri_fn(pts.data(), ns, t.data(), 2 * ns + 16);
        if (nt < 0) nt = 0;
        t.resize((size_t)3 * nt);
        std::vector<float>& pl = side ? r->planes2 : r->planes1;
        pl.resize((size_t)6 * nt);
        orc_planes(r->support.data(), t.data(), nt, pl.data());
    }
    int32_t gw, gh;
    orc_grid_dims(p, W, H, &gw, &gh);
    r->grid1.resize((size_t)gw * gh * (p->disp_max + 2));
    r->grid2.resize((size_t)gw * gh * (p->disp_max + 2));
    orc_grid(p, r->support.data(), ns, W, H, 0, r->grid1.data());
    orc_grid(p, r->support.data(), ns, W, H, 1, r->grid2.data());

    const int32_t DW = p->subsampling ? W / 2 : W, DH = p->subsampling ? H / 2 : H;
    const size_t DN = (size_t)DW * DH;
    r->d1_raw.resize(DN);
    r->d2_raw.resize(DN);
    orc_dense(p, r->support.data(), r->tri1.data(), r->planes1.data(), (int32_t)(r->tri1.size() / 3),
              r->grid1.data(), r->desc1.data(), r->desc2.data(), W, H, 0, r->d1_raw.data());
    orc_dense(p, r->support.data(), r->tri2.data(), r->planes2.data(), (int32_t)(r->tri2.size() / 3),
              r->grid2.data(), r->desc1.data(), r->desc2.data(), W, H, 1, r->d2_raw.data());
    r->d1_lr = r->d1_raw;
    r->d2_lr = r->d2_raw;
    orc_lr_check(p, r->d1_lr.data(), r->d2_lr.data(), DW, DH);
    r->d1_seg = r->d1_lr;
    r->d2_seg = r->d2_lr;
    orc_remove_small_segments(p, r->d1_seg.data(), DW, DH);
    if (!p->postprocess_only_left) orc_remove_small_segments(p, r->d2_seg.data(), DW, DH);
    r->d1_gap = r->d1_seg;
    r->d2_gap = r->d2_seg;
    orc_gap_interpolation(p, r->d1_gap.data(), DW, DH);
    if (!p->postprocess_only_left) orc_gap_interpolation(p, r->d2_gap.data(), DW, DH);
    r->d1 = r->d1_gap;
    r->d2 = r->d2_gap;
    if (p->filter_adaptive_mean) {
        orc_adaptive_mean(p, r->d1.data(), DW, DH);
        if (!p->postprocess_only_left) orc_adaptive_mean(p, r->d2.data(), DW, DH);
    }
    if (p->filter_median) {
        orc_median(p, r->d1.data(), DW, DH);
        if (!p->postprocess_only_left) orc_median(p, r->d2.data(), DW, DH);
    }
    return r;
}

int32_t orc_elas_run_status(orc_run* r) { return r->status; }
void orc_elas_run_free(orc_run* r) { delete r; }

int64_t orc_elas_run_get(orc_run* r, int32_t stage, void* buf, int64_t cap) {
    const void* src = 0;
    int64_t n = 0;
#define VEC(v) src = (v).data(); n = (int64_t)((v).size() * sizeof((v)[0]));
    switch (stage) {
        case SVH_ELAS_DESC1: VEC(r->desc1) break;
        case SVH_ELAS_DESC2: VEC(r->desc2) break;
        case SVH_ELAS_DCAN_RAW: VEC(r->dcan_raw) break;
        case SVH_ELAS_SUPPORT: VEC(r->support) break;
        case SVH_ELAS_TRI1: VEC(r->tri1) break;
        case SVH_ELAS_TRI2: VEC(r->tri2) break;
        case SVH_ELAS_PLANES1: VEC(r->planes1) break;
        case SVH_ELAS_PLANES2: VEC(r->planes2) break;
        case SVH_ELAS_GRID1: VEC(r->grid1) break;
        case SVH_ELAS_GRID2: VEC(r->grid2) break;
        case SVH_ELAS_D1_RAW: VEC(r->d1_raw) break;
        case SVH_ELAS_D2_RAW: VEC(r->d2_raw) break;
        case SVH_ELAS_D1_LR: VEC(r->d1_lr) break;
        case SVH_ELAS_D2_LR: VEC(r->d2_lr) break;
        case SVH_ELAS_D1_SEG: VEC(r->d1_seg) break;
        case SVH_ELAS_D2_SEG: VEC(r->d2_seg) break;
        case SVH_ELAS_D1_GAP: VEC(r->d1_gap) break;
        case SVH_ELAS_D2_GAP: VEC(r->d2_gap) break;
        case SVH_ELAS_STAGE_COUNT + 0: VEC(r->d1) break;
        case SVH_ELAS_STAGE_COUNT + 1: VEC(r->d2) break;
        default: return -1;
    }
#undef VEC
    if (buf && cap >= n && n > 0) memcpy(buf, src, (size_t)n);
    return n;
}

int32_t orc_elas_process(const svh_elas_params* p, const uint8_t* I1, const uint8_t* I2, float* D1,
                         float* D2, const int32_t* dims, orc_triangulate_fn tri_fn) {
    orc_run* r = orc_elas_run(p, I1, I2, dims, tri_fn);
    int32_t st = r->status;
    if (st == 0) {
        memcpy(D1, r->d1.data(), r->d1.size() * sizeof(float));
        memcpy(D2, r->d2.data(), r->d2.size() * sizeof(float));
    }
    delete r;
    return st;
}

}  // extern "C"
