// Host-resident stages of the ELAS pipeline (SURVEY 8a, E4 second half .. E9).
//
// These stages are tiny (a 249x75 lattice, ~1 k support points, ~2 k triangles)
// and strictly serial in the reference, so they stay on the CPU between the two
// device phases -- but they are the whole host cost of a pair (~0.3 ms of one core:
// filters ~0.08 ms, two triangulations ~0.2 ms; SVH_HOST_PROF=1 prints the split),
// which is what bounds throughput when a rank has few cores.  They reproduce the
// reference's observable behaviour:
//   * the consistency / redundancy filters mutate D_can in place while scanning
//     u-major (elas.cpp:174-279), and row 0 / column 0 of D_can hold the valid
//     disparity 0 left by calloc (elas.cpp:464-479);
//   * the triangulation reproduces Triangle's output order (delaunay.cpp).
// Plane fits, triangle edge lines and the disparity grid are evaluated on the
// device (k_prior / k_grid_* in elas_kernels.hip) from the support points and
// triangle lists uploaded here.  Compiled with -ffp-contract=off.
#include <emmintrin.h>
#include <math.h>
#include <stddef.h>
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

// The three lattice filters in scalar form: any window size, threshold or value range.
static void lattice_filters_scalar(const svh_elas_params& p, int32_t Wc, int32_t Hc, int16_t* dc) {
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
}

// The same filters, 8 lattice cells per SSE2 operation.  Invalid cells become kGone (far
// from every disparity) and the lattice gets a kGone border, so "similar" is one compare of
// |a - b| and no window is clamped.
//   * consistency: the scan order (u-major, in place) is kept cell by cell; the vector work is
//     inside the window, whose rows are visited centre-out because only "fewer than
//     incon_min_support" matters and near rows settle that first;
//   * redundancy: the vertical pass only couples cells of one column and the horizontal pass
//     only cells of one row, so 8 columns (rows) advance together, each still in order.
// Returns false (nothing done) when the parameters or values do not fit the 16-bit lanes.
namespace {
constexpr int16_t kGone = -16000;   // |kGone - d| stays inside int16 and above every threshold
constexpr int32_t kPad = 8;   // >= redundancy distance 5, >= window half width, one vector

inline __m128i abs_diff16(__m128i a, __m128i b) {
    const __m128i d = _mm_sub_epi16(a, b);
    return _mm_max_epi16(d, _mm_sub_epi16(_mm_setzero_si128(), d));
}

// one redundancy pass over a padded plane: `line` advances along the in-place direction,
// lanes are 8 neighbouring independent lines (pitch = elements between successive steps).
// |x - d| <= 1  <=>  (uint16)(x - d + 1) <= 2; the smallest such key over the five cells on a
// side decides, and unsigned order is signed order after adding 0x8000.
void redundancy_pass(int16_t* plane, int32_t pitch, int32_t lines, int32_t steps) {
    const __m128i gone = _mm_set1_epi16(kGone);
    const __m128i bias = _mm_set1_epi16((int16_t)(0x8000 - 1)), hit = _mm_set1_epi16(-32765);
    for (int32_t l0 = 0; l0 < lines; l0 += 8) {
        int16_t* base = plane + l0;
        for (int32_t t = 0; t < steps; t++) {
            int16_t* at = base + (size_t)t * pitch;
            const __m128i dv = _mm_loadu_si128(reinterpret_cast<const __m128i*>(at));
            const __m128i c = _mm_add_epi16(dv, bias);   // x - c = x - d + 1 + 0x8000 (mod 2^16)
            __m128i before = _mm_set1_epi16(32767), after = before;
            for (int32_t j = 1; j <= 5; j++) {
                const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(at - (ptrdiff_t)j * pitch));
                const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(at + (ptrdiff_t)j * pitch));
                before = _mm_min_epi16(before, _mm_sub_epi16(b, c));
                after = _mm_min_epi16(after, _mm_sub_epi16(a, c));
            }
            // kGone lanes compare equal to their kGone neighbours; rewriting kGone is harmless
            const __m128i kill = _mm_cmplt_epi16(_mm_max_epi16(before, after), hit);
            const __m128i out = _mm_or_si128(_mm_and_si128(kill, gone), _mm_andnot_si128(kill, dv));
            _mm_storeu_si128(reinterpret_cast<__m128i*>(at), out);
        }
    }
}

// dst[c * dpitch + r] = src[r * spitch + c] for an 8 x 8 block of int16
inline void transpose8x8(const int16_t* src, int32_t spitch, int16_t* dst, int32_t dpitch) {
    __m128i r[8], t[8];
    for (int i = 0; i < 8; i++) r[i] = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + (size_t)i * spitch));
    for (int i = 0; i < 4; i++) {
        t[2 * i] = _mm_unpacklo_epi16(r[2 * i], r[2 * i + 1]);
        t[2 * i + 1] = _mm_unpackhi_epi16(r[2 * i], r[2 * i + 1]);
    }
    r[0] = _mm_unpacklo_epi32(t[0], t[2]); r[1] = _mm_unpackhi_epi32(t[0], t[2]);
    r[2] = _mm_unpacklo_epi32(t[1], t[3]); r[3] = _mm_unpackhi_epi32(t[1], t[3]);
    r[4] = _mm_unpacklo_epi32(t[4], t[6]); r[5] = _mm_unpackhi_epi32(t[4], t[6]);
    r[6] = _mm_unpacklo_epi32(t[5], t[7]); r[7] = _mm_unpackhi_epi32(t[5], t[7]);
    t[0] = _mm_unpacklo_epi64(r[0], r[4]); t[1] = _mm_unpackhi_epi64(r[0], r[4]);
    t[2] = _mm_unpacklo_epi64(r[1], r[5]); t[3] = _mm_unpackhi_epi64(r[1], r[5]);
    t[4] = _mm_unpacklo_epi64(r[2], r[6]); t[5] = _mm_unpackhi_epi64(r[2], r[6]);
    t[6] = _mm_unpacklo_epi64(r[3], r[7]); t[7] = _mm_unpackhi_epi64(r[3], r[7]);
    for (int i = 0; i < 8; i++) _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + (size_t)i * dpitch), t[i]);
}
}  // namespace

static bool lattice_filters_simd(const svh_elas_params& p, int32_t Wc, int32_t Hc, const int16_t* in,
                                 std::vector<int16_t>& rowmajor, std::vector<int16_t>& colmajor) {
    const int32_t ws = p.incon_window_size, thr = p.incon_threshold, need = p.incon_min_support;
    if (ws < 0 || ws > 7 || thr < 0 || thr > 8000) return false;
    // row-major padded plane: Rp[(vc + kPad) * Wp + uc + kPad]; lanes past Wc stay kGone
    const int32_t Wp = ((Wc + 7) & ~7) + 2 * kPad, Hp = Hc + 2 * kPad;
    rowmajor.assign((size_t)Wp * Hp, kGone);
    int32_t top = 0;
    for (int32_t vc = 0; vc < Hc; vc++) {
        int16_t* dst = rowmajor.data() + (size_t)(vc + kPad) * Wp + kPad;
        const int16_t* src = in + (size_t)vc * Wc;
        for (int32_t uc = 0; uc < Wc; uc++) {
            top = std::max<int32_t>(top, src[uc]);
            dst[uc] = src[uc] < 0 ? kGone : src[uc];
        }
    }
    if (top > 8000) return false;
    // --- consistency (elas.cpp:174-209)
    if (need > 0) {
        // lanes 0 .. 2*ws of the 16 loaded from (uc - ws) are the window row
        alignas(16) int16_t lane_on[16];
        for (int i = 0; i < 16; i++) lane_on[i] = i <= 2 * ws ? -1 : 0;
        const __m128i on_lo = _mm_load_si128(reinterpret_cast<const __m128i*>(lane_on));
        const __m128i on_hi = _mm_load_si128(reinterpret_cast<const __m128i*>(lane_on + 8));
        const __m128i lim = _mm_set1_epi16((int16_t)(thr + 1));
        for (int32_t uc = 0; uc < Wc; uc++) {
            int16_t* col = rowmajor.data() + (size_t)kPad * Wp + kPad + uc;
            for (int32_t vc = 0; vc < Hc; vc++) {
                const int16_t dv = col[(size_t)vc * Wp];
                if (dv == kGone) continue;
                const __m128i dvv = _mm_set1_epi16(dv);
                // most cells are settled by the 8 lanes around them in their own row
                const __m128i own = _mm_loadu_si128(reinterpret_cast<const __m128i*>(col + (size_t)vc * Wp - ws));
                const __m128i own_ok = _mm_and_si128(_mm_cmplt_epi16(abs_diff16(own, dvv), lim), on_lo);
                if (__builtin_popcount((unsigned)_mm_movemask_epi8(own_ok)) >= 2 * need) continue;
                int32_t cnt = 0;
                for (int32_t k = 0; k <= 2 * ws && cnt < need; k++) {
                    const int32_t off = (k & 1) ? (k + 1) / 2 : -(k / 2);   // 0, +1, -1, +2, -2, ...
                    const int16_t* row = col + (ptrdiff_t)(vc + off) * Wp - ws;
                    const __m128i lo = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row));
                    const __m128i hi = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + 8));
                    const __m128i ok_lo = _mm_and_si128(_mm_cmplt_epi16(abs_diff16(lo, dvv), lim), on_lo);
                    const __m128i ok_hi = _mm_and_si128(_mm_cmplt_epi16(abs_diff16(hi, dvv), lim), on_hi);
                    cnt += __builtin_popcount((unsigned)_mm_movemask_epi8(_mm_packs_epi16(ok_lo, ok_hi)));
                }
                if (cnt < need) col[(size_t)vc * Wp] = kGone;
            }
        }
    }
    // --- redundancy, vertical (elas.cpp:213-279 with the arguments of elas.cpp:501)
    redundancy_pass(rowmajor.data() + (size_t)kPad * Wp + kPad, Wp, Wc, Hc);
    // --- transpose: Cp[(uc + kPad) * Hq + vc + kPad]
    const int32_t Hq = ((Hc + 7) & ~7) + 2 * kPad, Wq = Wc + 2 * kPad;
    colmajor.assign((size_t)Hq * Wq, kGone);
    // 8 x 8 blocks; the source rows / columns past Hc / Wc are border cells (kGone), and the
    // destination has room for them: Wq - kPad >= Wc rounded up to 8 (kPad = 8)
    for (int32_t vc = 0; vc < Hc; vc += 8)
        for (int32_t uc = 0; uc < Wc; uc += 8)
            transpose8x8(rowmajor.data() + (size_t)(vc + kPad) * Wp + kPad + uc, Wp,
                         colmajor.data() + (size_t)(uc + kPad) * Hq + kPad + vc, Hq);
    // --- redundancy, horizontal (elas.cpp:502)
    redundancy_pass(colmajor.data() + (size_t)kPad * Hq + kPad, Hq, Hc, Wc);
    return true;
}

// dc: the candidate lattice [Hc][Wc]; with write_back it receives the filtered lattice (the
// reference's D_can after elas.cpp:502), otherwise it is only read.
void support_from_candidates(const svh_elas_params& p, const Dims& d, int16_t* dc,
                             std::vector<int32_t>& support, bool write_back) {
    const int32_t Wc = d.Wc, Hc = d.Hc;
    static thread_local std::vector<int16_t> rowmajor, colmajor;
    support.clear();
    if (lattice_filters_simd(p, Wc, Hc, dc, rowmajor, colmajor)) {
        // lattice -> list, u-major from 1 (elas.cpp:505-517): contiguous in the transposed plane
        const int32_t Hq = ((Hc + 7) & ~7) + 2 * kPad;
        for (int32_t uc = 1; uc < Wc; uc++) {
            const int16_t* col = colmajor.data() + (size_t)(uc + kPad) * Hq + kPad;
            for (int32_t v0 = 0; v0 < Hc; v0 += 8) {
                const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(col + v0));
                unsigned live = ~(unsigned)_mm_movemask_epi8(_mm_cmpeq_epi16(x, _mm_set1_epi16(kGone))) & 0xAAAAu;
                if (v0 == 0) live &= ~3u;   // row 0 is not part of the list
                for (; live; live &= live - 1) {
                    const int32_t vc = v0 + (__builtin_ctz(live) >> 1);   // < Hc: the rest is border
                    support.push_back(uc * d.step);
                    support.push_back(vc * d.step);
                    support.push_back(col[vc]);
                }
            }
        }
        if (write_back)
            for (int32_t uc = 0; uc < Wc; uc++) {
                const int16_t* col = colmajor.data() + (size_t)(uc + kPad) * Hq + kPad;
                for (int32_t vc = 0; vc < Hc; vc++) dc[vc * Wc + uc] = col[vc] == kGone ? -1 : col[vc];
            }
    } else {
        std::vector<int16_t> work;
        int16_t* w = dc;
        if (!write_back) {
            work.assign(dc, dc + (size_t)Wc * Hc);
            w = work.data();
        }
        lattice_filters_scalar(p, Wc, Hc, w);
        for (int32_t uc = 1; uc < Wc; uc++)
            for (int32_t vc = 1; vc < Hc; vc++) {
                int16_t dv = w[vc * Wc + uc];
                if (dv >= 0) {
                    support.push_back(uc * d.step);
                    support.push_back(vc * d.step);
                    support.push_back(dv);
                }
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
    // (two sides x two halves; four halves per side measured no faster: 119-133 us against 167-172 serial)
    const int par_side = 1;
    auto one = [&](int side) {
        std::vector<float> pts((size_t)2 * n);
        for (int32_t i = 0; i < n; i++) {
            pts[2 * i] = (float)(side ? hp.support[3 * i] - hp.support[3 * i + 2] : hp.support[3 * i]);
            pts[2 * i + 1] = (float)hp.support[3 * i + 1];
        }
        std::vector<int32_t>& tri = hp.tri[side];
        tri.resize((size_t)3 * (2 * n + 8));
        const int32_t nt = delaunay(pts.data(), n, tri.data(), 2 * n + 8, parallel ? par_side : 0);
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
