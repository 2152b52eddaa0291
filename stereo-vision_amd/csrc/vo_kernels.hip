// VisualOdometryStereo::estimateMotion on the device
//   libviso2/src/viso_stereo.cpp:72-228 (RANSAC + iterated Gauss-Newton),
//   :232-255 getInlier, :259-323 updateParameters, :341-485 residuals / Jacobian.
//
// k_vo_ransac   one wave per RANSAC hypothesis (the 200 hypotheses are independent):
//               Gauss-Newton on its 3 sampled matches, then the inlier vote over all N.
// k_vo_refine   one workgroup: first hypothesis with the most inliers, its inlier list
//               in index order, final Gauss-Newton over the inliers.
//
// Arithmetic is fp64 in the reference's operation order: the normal equations are
// summed over the Jacobian rows in ascending order by one lane per matrix entry, no
// FMA contraction (library flag -ffp-contract=off plus explicit *_rn intrinsics in
// the sums).  sin/cos are the device libm's; they can differ from glibc's in the
// last bit, which is the only source of (<= 1e-12) deviations from the CPU result.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svh.h"
#include "batch_rec.h"
#include "vo_internal.h"

namespace svh {

namespace {

struct Rot {   // R = Rx*Ry*Rz and its partial derivatives (viso_stereo.cpp:352-377)
    double r00, r01, r02, r10, r11, r12, r20, r21, r22;
    double ax10, ax11, ax12, ax20, ax21, ax22;
    double ay00, ay01, ay02, ay10, ay11, ay12, ay20, ay21, ay22;
    double az00, az01, az10, az11, az20, az21;
};

__device__ __forceinline__ Rot make_rot(const double* tr) {
    // (sincos: one argument reduction per angle instead of two -- six fp64 sine / cosine evaluations are a large part
    // of a Gauss-Newton iteration of one hypothesis)
    double sx, cx, sy, cy, sz, cz;
    sincos(tr[0], &sx, &cx);
    sincos(tr[1], &sy, &cy);
    sincos(tr[2], &sz, &cz);
    Rot R;
    R.r00 = +cy * cz;                R.r01 = -cy * sz;                R.r02 = +sy;
    R.r10 = +sx * sy * cz + cx * sz; R.r11 = -sx * sy * sz + cx * cz; R.r12 = -sx * cy;
    R.r20 = -cx * sy * cz + sx * sz; R.r21 = +cx * sy * sz + sx * cz; R.r22 = +cx * cy;
    R.ax10 = +cx * sy * cz - sx * sz; R.ax11 = -cx * sy * sz - sx * cz; R.ax12 = -cx * cy;
    R.ax20 = +sx * sy * cz + cx * sz; R.ax21 = -sx * sy * sz + cx * cz; R.ax22 = -sx * cy;
    R.ay00 = -sy * cz;      R.ay01 = +sy * sz;      R.ay02 = +cy;
    R.ay10 = +sx * cy * cz; R.ay11 = -sx * cy * sz; R.ay12 = +sx * sy;
    R.ay20 = -cx * cy * cz; R.ay21 = +cx * cy * sz; R.ay22 = -cx * sy;
    R.az00 = -cy * sz;                R.az01 = -cy * cz;
    R.az10 = -sx * sy * sz + cx * cz; R.az11 = -sx * sy * cz - cx * sz;
    R.az20 = +cx * sy * sz + sx * cz; R.az21 = +cx * sy * cz - sx * sz;
    return R;
}

struct Point3 { double X, Y, Z; };

// back-projection of the previous-frame match (viso_stereo.cpp:113-131)
__device__ __forceinline__ Point3 back_project(const svh_p_match& m, const VoCalib& c) {
    const float df = m.u1p - m.u2p;
    const double d = (double)(df > 0.0001f ? df : 0.0001f);
    Point3 p;
    p.X = ((double)m.u1p - c.cu) * c.base / d;
    p.Y = ((double)m.v1p - c.cv) * c.base / d;
    p.Z = c.f * c.base / d;
    return p;
}

// prediction of one match under (R, t): p[0..3] = u1c, v1c, u2c, v2c; also X1c.., weight
struct Pred { double X1c, Y1c, Z1c, X2c, p[4]; };
__device__ __forceinline__ Pred predict(const Rot& R, const double* tr, const Point3& P, const VoCalib& c) {
    Pred q;
    q.X1c = R.r00 * P.X + R.r01 * P.Y + R.r02 * P.Z + tr[3];
    q.Y1c = R.r10 * P.X + R.r11 * P.Y + R.r12 * P.Z + tr[4];
    q.Z1c = R.r20 * P.X + R.r21 * P.Y + R.r22 * P.Z + tr[5];
    q.X2c = q.X1c - c.base;
    q.p[0] = c.f * q.X1c / q.Z1c + c.cu;
    q.p[1] = c.f * q.Y1c / q.Z1c + c.cv;
    q.p[2] = c.f * q.X2c / q.Z1c + c.cu;
    q.p[3] = q.p[1];
    return q;
}

// column j of the four Jacobian rows of one match (viso_stereo.cpp:431-466)
__device__ __forceinline__ void jacobian_col(const Rot& R, const Point3& P, const Pred& q, double weight,
                                             const VoCalib& c, int j, double* J0, double* J1, double* J2) {
    double dX = 0, dY = 0, dZ = 0;
    switch (j) {
        case 0: dY = R.ax10 * P.X + R.ax11 * P.Y + R.ax12 * P.Z;
                dZ = R.ax20 * P.X + R.ax21 * P.Y + R.ax22 * P.Z; break;
        case 1: dX = R.ay00 * P.X + R.ay01 * P.Y + R.ay02 * P.Z;
                dY = R.ay10 * P.X + R.ay11 * P.Y + R.ay12 * P.Z;
                dZ = R.ay20 * P.X + R.ay21 * P.Y + R.ay22 * P.Z; break;
        case 2: dX = R.az00 * P.X + R.az01 * P.Y;
                dY = R.az10 * P.X + R.az11 * P.Y;
                dZ = R.az20 * P.X + R.az21 * P.Y; break;
        case 3: dX = 1; break;
        case 4: dY = 1; break;
        default: dZ = 1; break;
    }
    const double zz = q.Z1c * q.Z1c;
    *J0 = weight * c.f * (dX * q.Z1c - q.X1c * dZ) / zz;
    *J1 = weight * c.f * (dY * q.Z1c - q.Y1c * dZ) / zz;
    *J2 = weight * c.f * (dX * q.Z1c - q.X2c * dZ) / zz;
}

__device__ __forceinline__ double obs_weight(double u1c, const VoCalib& c) {
    return c.reweighting ? 1.0 / (fabs(u1c - c.cu) / fabs(c.cu) + 0.05) : 1.0;
}

// Matrix::solve (6x6, one right-hand side)   libviso2/src/matrix.cpp:648-760
// Gauss-Jordan with full pivoting, run by ONE WAVE: lane e < 42 owns element (e/7, e%7) of the
// augmented matrix [A | B].  The reference scans rows then columns and takes ">=": the LAST
// maximum wins, i.e. the largest (|a|, scan position) pair; rows are swapped, the pivot row is
// scaled by 1/pivot (after the pivot itself was set to 1) and eliminated from the others with
// the same expressions, so every element sees the reference's operation sequence.
// All 64 lanes must call this.  Returns false (uniformly) for a pivot below 1e-20.
__device__ __forceinline__ bool wave_solve6(double& a, int lane) {
    const int r = lane / 7, cidx = lane - 7 * r;
    const bool elem = lane < 42;
    uint32_t used = 0;   // ipiv as a bit set
    for (int it = 0; it < 6; it++) {
        double v = -1.0;
        int pos = -1;
        if (elem && cidx < 6 && !((used >> r) & 1) && !((used >> cidx) & 1)) {
            v = fabs(a);
            pos = r * 6 + cidx;
        }
        // largest (|a|, position) pair of the wave.  Round 6: four DPP steps inside each row of 16 lanes (pairs, quads,
        // half-row mirror, row mirror: every lane then holds its row's winner) and the four row winners combined from
        // scalar reads, instead of six dependent ds_bpermute round trips per pivot -- the solve was a chain of ~55
        // LDS-crossbar latencies, two thirds of a Gauss-Newton iteration of one hypothesis.  The winner does not
        // depend on the order in which pairs are compared.
        auto take = [&](double ov, int op) {
            if (ov > v || (ov == v && op > pos)) {
                v = ov;
                pos = op;
            }
        };
#define SVH_DPP_STEP(CTRL)                                                                                      \
        {                                                                                                       \
            const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);            \
            const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);            \
            const int op = __builtin_amdgcn_update_dpp(0, pos, CTRL, 0xF, 0xF, false);                          \
            take(__hiloint2double(hi, lo), op);                                                                 \
        }
        SVH_DPP_STEP(0xB1)    // quad_perm [1,0,3,2]
        SVH_DPP_STEP(0x4E)    // quad_perm [2,3,0,1]
        SVH_DPP_STEP(0x141)   // row_half_mirror
        SVH_DPP_STEP(0x140)   // row_mirror
#undef SVH_DPP_STEP
        {
            double rv[4];
            int rp[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(v), 16 * q);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 16 * q);
                rv[q] = __hiloint2double(hi, lo);
                rp[q] = __builtin_amdgcn_readlane(pos, 16 * q);
            }
            v = rv[0];
            pos = rp[0];
#pragma unroll
            for (int q = 1; q < 4; q++) take(rv[q], rp[q]);
        }
        pos = __builtin_amdgcn_readfirstlane(pos);   // (the same in every lane: scalar from here on)
        const int irow = pos / 6, icol = pos - 6 * irow;
        used |= 1u << icol;
        if (irow != icol) {   // swap rows irow and icol (uniform branch)
            const int src = r == irow ? icol * 7 + cidx : (r == icol ? irow * 7 + cidx : lane);
            a = __shfl(a, elem ? src : lane, 64);
        }
        const double pivot = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), icol * 7 + icol),
                                              __builtin_amdgcn_readlane(__double2loint(a), icol * 7 + icol));   // (a uniform lane)
        if (fabs(pivot) < 1e-20) return false;
        const double pivinv = __ddiv_rn(1.0, pivot);
        if (elem && r == icol) a = __dmul_rn(cidx == icol ? 1.0 : a, pivinv);
        const double rowval = __shfl(a, icol * 7 + (elem ? cidx : 0), 64);   // A[icol][l] after scaling
        const double dum = __shfl(a, (elem ? r : 0) * 7 + icol, 64);         // A[ll][icol] before zeroing
        if (elem && r != icol) a = __dsub_rn(cidx == icol ? 0.0 : a, __dmul_rn(rowval, dum));
    }
    return true;
}

enum { VO_UPDATED = 0, VO_FAILED = 1, VO_CONVERGED = 2 };

// One Gauss-Newton step by one wave (viso_stereo.cpp:281-322).  J is [rows][6] row major and
// res [rows] (LDS or global).  Lane e < 42 sums its entry of [JtJ | Jt r] over the rows in
// ascending order (the reference's loop order; loads are batched, the add chain is not
// reordered), the wave solves, lanes 0..5 update tr (LDS).  Returns the status uniformly.
// kCols: J is stored column by column, J[col * stride + row] (stride a multiple of 8, the rows up to the next multiple
// of 8 exist and are ignored): a lane's eight operands of a turn are consecutive, two per 16-byte LDS read -- half the
// LDS instructions of the row-major form, which issued 2 000 of them per lane and iteration next to the 1 000 additions
// that have to be a chain.  The order of the additions is the same.
template <bool kCols = false>
__device__ __forceinline__ int wave_gn_step(const double* J, const double* res, int rows, int lane,
                                            double* s_tr, double eps, int stride = 0) {
    const int m = lane / 7, n = lane - 7 * m;
    double acc = 0.0;
    if (lane < 42) {
        const double* xc = kCols ? J + (size_t)m * stride : J;
        const double* yc = kCols ? (n < 6 ? J + (size_t)n * stride : res) : J;
        int i0 = 0;
        if (kCols) {
            // whole turns of 8 rows without the per-row test, the next turn's operands requested before this turn's
            // chain of additions (the chain is what has to be serial; nothing else should wait for it)
            const int full = rows & ~7;
            double x[8], y[8], nx[8], ny[8];
            if (full > 0) {
#pragma unroll
                for (int q = 0; q < 8; q++) { x[q] = xc[q]; y[q] = yc[q]; }
            }
            for (; i0 < full; i0 += 8) {
                // (past the last turn this reads the padding rows: they exist, see the layout, and are dropped)
#pragma unroll
                for (int q = 0; q < 8; q++) { nx[q] = xc[i0 + 8 + q]; ny[q] = yc[i0 + 8 + q]; }
#pragma unroll
                for (int q = 0; q < 8; q++) acc = __dadd_rn(acc, __dmul_rn(x[q], y[q]));
#pragma unroll
                for (int q = 0; q < 8; q++) { x[q] = nx[q]; y[q] = ny[q]; }
            }
        }
        for (; i0 < rows; i0 += 8) {
            double x[8], y[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (kCols) {
                    x[q] = xc[i0 + q];
                    y[q] = yc[i0 + q];
                    continue;
                }
                const int i = i0 + q < rows ? i0 + q : rows - 1;
                x[q] = J[i * 6 + m];
                y[q] = n < 6 ? J[i * 6 + n] : res[i];
            }
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (i0 + q < rows) acc = __dadd_rn(acc, __dmul_rn(x[q], y[q]));
        }
    }
    if (!wave_solve6(acc, lane)) return VO_FAILED;
    // solution = column 6
    bool big = false;
    if (lane < 42 && n == 6) {
        s_tr[m] = __dadd_rn(s_tr[m], acc);   // step_size = 1
        big = fabs(acc) > eps;
    }
    return __ballot(big) ? VO_UPDATED : VO_CONVERGED;
}

}  // namespace

__device__ __forceinline__ void d_vo_ransac(const svh_p_match* __restrict__ pm, int N,
                                            const int32_t* __restrict__ samples, VoCalib c,
                                            double* __restrict__ hyp_tr,       // [iters][6]
                                            int32_t* __restrict__ hyp_count,   // [iters], -1 = failed
                                            uint8_t* __restrict__ hyp_flags,   // [iters][N]
                                            int k) {
    __shared__ double s_tr[6], s_J[12 * 6], s_res[12];
    const int lane = threadIdx.x;
    if (lane < 6) s_tr[lane] = 0.0;
    const int pt = lane / 6, col = lane - 6 * pt;      // lanes 0..17: (sampled match, parameter)
    svh_p_match m = pm[0];
    Point3 P = {0, 0, 0};
    if (lane < 18) {
        m = pm[samples[3 * k + pt]];
        P = back_project(m, c);
    }
    __syncthreads();
    int status = VO_UPDATED, iter = 0;
    while (status == VO_UPDATED) {
        double tr[6];
#pragma unroll
        for (int i = 0; i < 6; i++) tr[i] = s_tr[i];
        __syncthreads();   // everyone has read s_tr before the step updates it
        if (lane < 18) {
            const Rot R = make_rot(tr);
            const Pred q = predict(R, tr, P, c);
            const double w = obs_weight((double)m.u1c, c);
            double j0, j1, j2;
            jacobian_col(R, P, q, w, c, col, &j0, &j1, &j2);
            s_J[(4 * pt + 0) * 6 + col] = j0;
            s_J[(4 * pt + 1) * 6 + col] = j1;
            s_J[(4 * pt + 2) * 6 + col] = j2;
            s_J[(4 * pt + 3) * 6 + col] = j1;
            if (col == 0) {
                s_res[4 * pt + 0] = w * ((double)m.u1c - q.p[0]);
                s_res[4 * pt + 1] = w * ((double)m.v1c - q.p[1]);
                s_res[4 * pt + 2] = w * ((double)m.u2c - q.p[2]);
                s_res[4 * pt + 3] = w * ((double)m.v2c - q.p[3]);
            }
        }
        __syncthreads();
        status = wave_gn_step(s_J, s_res, 12, lane, s_tr, 1e-6);
        __syncthreads();
        if (iter++ > 20 || status == VO_CONVERGED) break;
    }
    // inlier vote over all matches (getInlier, viso_stereo.cpp:232-255)
    int count = -1;
    if (status != VO_FAILED) {
        double tr[6];
#pragma unroll
        for (int i = 0; i < 6; i++) tr[i] = s_tr[i];
        const Rot R = make_rot(tr);
        const double thr = c.inlier_threshold * c.inlier_threshold;
        count = 0;
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            bool in = false;
            if (i < N) {
                const svh_p_match mi = pm[i];
                const Pred q = predict(R, tr, back_project(mi, c), c);
                const double e0 = (double)mi.u1c - q.p[0], e1 = (double)mi.v1c - q.p[1];
                const double e2 = (double)mi.u2c - q.p[2], e3 = (double)mi.v2c - q.p[3];
                const double ss = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(e0, e0), __dmul_rn(e1, e1)),
                                                      __dmul_rn(e2, e2)), __dmul_rn(e3, e3));
                in = ss < thr;
                hyp_flags[(size_t)k * N + i] = in ? 1 : 0;
            }
            count += __popcll(__ballot(in));
        }
    }
    if (lane < 6) hyp_tr[6 * k + lane] = s_tr[lane];
    if (lane == 0) hyp_count[k] = count;
}

// Gauss-Newton refinement on the inlier set (viso_stereo.cpp:122-150 on the winning hypothesis):
// all 256 threads fill the Jacobian / residual rows, wave 0 forms and solves the normal equations.
template <bool kCols>
__device__ __forceinline__ int vo_refine_loop(double* J, double* res, const svh_p_match* __restrict__ pm,
                                              const int32_t* __restrict__ inliers, int nin, const VoCalib& c,
                                              double* s_tr, int* s_status, int t, int stride) {
    int status = VO_UPDATED, iter = 0;
    while (status == VO_UPDATED) {
        double tr[6];
#pragma unroll
        for (int i = 0; i < 6; i++) tr[i] = s_tr[i];
        const Rot R = make_rot(tr);
        for (int a = t; a < nin; a += 256) {
            const svh_p_match m = pm[inliers[a]];
            const Point3 P = back_project(m, c);
            const Pred q = predict(R, tr, P, c);
            const double w = obs_weight((double)m.u1c, c);
#pragma unroll
            for (int col = 0; col < 6; col++) {
                double j0, j1, j2;
                jacobian_col(R, P, q, w, c, col, &j0, &j1, &j2);
                if (kCols) {
                    double* jc = J + (size_t)col * stride + 4 * a;
                    jc[0] = j0; jc[1] = j1; jc[2] = j2; jc[3] = j1;
                } else {
                    J[(size_t)(4 * a + 0) * 6 + col] = j0;
                    J[(size_t)(4 * a + 1) * 6 + col] = j1;
                    J[(size_t)(4 * a + 2) * 6 + col] = j2;
                    J[(size_t)(4 * a + 3) * 6 + col] = j1;
                }
            }
            res[4 * a + 0] = w * ((double)m.u1c - q.p[0]);
            res[4 * a + 1] = w * ((double)m.v1c - q.p[1]);
            res[4 * a + 2] = w * ((double)m.u2c - q.p[2]);
            res[4 * a + 3] = w * ((double)m.v2c - q.p[3]);
        }
        __syncthreads();
        if (t < 64) {   // wave 0 forms and solves the normal equations
            const int st = wave_gn_step<kCols>(J, res, 4 * nin, t, s_tr, 1e-8, stride);
            if (t == 0) *s_status = st;
        }
        __syncthreads();
        status = *s_status;
        if (iter++ > 100 || status == VO_CONVERGED) break;
    }
    return status == VO_CONVERGED;
}

// One workgroup of 256.  The Jacobian / residual rows of the inliers live in LDS when they fit
// (lds_rows >= 4 * inliers; the launcher sizes the dynamic LDS from N), else in global scratch.
__device__ __forceinline__ void d_vo_refine(const svh_p_match* __restrict__ pm, int N, int iters,
                                            VoCalib c, const double* __restrict__ hyp_tr,
                                            const int32_t* __restrict__ hyp_count,
                                            const uint8_t* __restrict__ hyp_flags,
                                            double* __restrict__ Jg, double* __restrict__ resg,
                                            int lds_rows, VoResult* __restrict__ out,
                                            int32_t* __restrict__ out_inliers) {
    extern __shared__ double s_rows[];   // J [6][lds_rows] (column by column) then [lds_rows] residuals
    __shared__ double s_tr[6];
    __shared__ int s_best, s_status, s_scan[256];
    const int t = threadIdx.x;
    if (t == 0) {
        int best = -1, bc = 0;   // "more inliers than the current set" starting from the empty set
        for (int k = 0; k < iters; k++)
            if (hyp_count[k] > bc) {
                bc = hyp_count[k];
                best = k;
            }
        s_best = best;
    }
    __syncthreads();
    const int best = s_best;
    // inlier list of the winner, ascending index: ordered compaction, chunk per thread
    int nin = 0;
    if (best >= 0) {
        const uint8_t* fl = hyp_flags + (size_t)best * N;
        const int chunk = (N + 255) / 256;
        const int lo = min(t * chunk, N), hi = min(lo + chunk, N);
        int mine = 0;
        for (int i = lo; i < hi; i++) mine += fl[i];
        // (scan inside each wave by shuffles, the four wave totals through LDS: two barriers instead of sixteen)
        const int lane = t & 63, wave = t >> 6;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        int pos = incl - mine;
        for (int w = 0; w < 4; w++) {
            pos += w < wave ? s_scan[w] : 0;
            nin += s_scan[w];
        }
        for (int i = lo; i < hi; i++)
            if (fl[i]) out_inliers[pos++] = i;
        if (t < 6) s_tr[t] = hyp_tr[6 * best + t];
    }
    __syncthreads();
    // two calls, so that each sees where J / res live (LDS: ds_read / ds_write; global scratch
    // otherwise) -- one generic pointer would turn every access into a flat load
    int success = 0;
    if (nin >= 6) {
        if (4 * nin <= lds_rows)   // (LDS: column by column, lds_rows is a multiple of 8)
            success = vo_refine_loop<true>(s_rows, s_rows + (size_t)6 * lds_rows, pm, out_inliers, nin, c, s_tr, &s_status, t, lds_rows);
        else
            success = vo_refine_loop<false>(Jg, resg, pm, out_inliers, nin, c, s_tr, &s_status, t, 0);
    }
    if (t == 0) {
        out->success = success;
        out->n_inliers = nin;
        out->best = best;
        for (int i = 0; i < 6; i++) out->tr[i] = best >= 0 ? s_tr[i] : 0.0;
    }
}

// plain and batched forms (batch_rec.h: job blockIdx.z of a table in device memory)
// (pointers read from the job table are told to be global memory: global_load instead of flat_load, see gptr in
// matcher_kernels.hip)
template <class T>
__device__ __forceinline__ T* vgptr(T* p) {
    __attribute__((address_space(1))) T* q = (__attribute__((address_space(1))) T*)p;
    asm volatile("" : "+v"(q));
    return (T*)q;
}
struct VoRansacJob { const svh_p_match* pm; int N; const int32_t* samples; VoCalib c; double* hyp_tr; int32_t* hyp_count; uint8_t* hyp_flags; int iters; };
__global__ __launch_bounds__(64) void k_vo_ransac(VoRansacJob a) {
    d_vo_ransac(a.pm, a.N, a.samples, a.c, a.hyp_tr, a.hyp_count, a.hyp_flags, (int)blockIdx.x);
}
__global__ __launch_bounds__(64) void k_vo_ransac_b(const VoRansacJob* J) {
    const VoRansacJob& a = J[blockIdx.z];
    if ((int)blockIdx.x >= a.iters) return;
    d_vo_ransac(vgptr(a.pm), a.N, vgptr(a.samples), a.c, vgptr(a.hyp_tr), vgptr(a.hyp_count), vgptr(a.hyp_flags), (int)blockIdx.x);
}
struct VoRefineJob {
    const svh_p_match* pm; int N, iters; VoCalib c; const double* hyp_tr; const int32_t* hyp_count; const uint8_t* hyp_flags;
    double *Jg, *resg; int lds_rows; VoResult* out; int32_t* out_inliers;
};
__global__ __launch_bounds__(256) void k_vo_refine(VoRefineJob a) {
    d_vo_refine(a.pm, a.N, a.iters, a.c, a.hyp_tr, a.hyp_count, a.hyp_flags, a.Jg, a.resg, a.lds_rows, a.out, a.out_inliers);
}
__global__ __launch_bounds__(256) void k_vo_refine_b(const VoRefineJob* J) {
    const VoRefineJob& a = J[blockIdx.z];
    d_vo_refine(vgptr(a.pm), a.N, a.iters, a.c, vgptr(a.hyp_tr), vgptr(a.hyp_count), vgptr(a.hyp_flags), vgptr(a.Jg), vgptr(a.resg),
                a.lds_rows, vgptr(a.out), vgptr(a.out_inliers));
}
struct VoUploadJob { const uint4* host; uint4* dev; size_t n16; };
__global__ __launch_bounds__(256) void k_vo_upload(VoUploadJob a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n16) a.dev[i] = a.host[i];
}
__global__ __launch_bounds__(256) void k_vo_upload_b(const VoUploadJob* J) {
    const VoUploadJob a = J[blockIdx.z];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n16) vgptr(a.dev)[i] = vgptr(a.host)[i];
}
static void b_vo_ransac(const void* jobs, int njobs, unsigned gx, unsigned gy, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL(k_vo_ransac_b, dim3(gx, gy, (unsigned)njobs), dim3(64), lds, s, reinterpret_cast<const VoRansacJob*>(jobs));
}
static void b_vo_refine(const void* jobs, int njobs, unsigned gx, unsigned gy, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL(k_vo_refine_b, dim3(gx, gy, (unsigned)njobs), dim3(256), lds, s, reinterpret_cast<const VoRefineJob*>(jobs));
}
static void b_vo_upload(const void* jobs, int njobs, unsigned gx, unsigned gy, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL(k_vo_upload_b, dim3(gx, gy, (unsigned)njobs), dim3(256), lds, s, reinterpret_cast<const VoUploadJob*>(jobs));
}

void vlaunch_upload(void* stream, const uint8_t* pinned, uint8_t* dev, size_t bytes) {
    const size_t n16 = bytes / 16;
    const VoUploadJob a = {reinterpret_cast<const uint4*>(pinned), reinterpret_cast<uint4*>(dev), n16};
    const unsigned gx = (unsigned)((n16 + 255) / 256);
    if (t_rec) return t_rec->add(b_vo_upload, a, gx);
    hipLaunchKernelGGL(k_vo_upload, dim3(gx), dim3(256), 0, (hipStream_t)stream, a);
}

void vlaunch_estimate(void* stream, const svh_p_match* pm, int N, const int32_t* samples, int iters,
                      const VoCalib& c, double* hyp_tr, int32_t* hyp_count, uint8_t* hyp_flags, double* Jg,
                      double* resg, VoResult* out, int32_t* out_inliers) {
    hipStream_t s = (hipStream_t)stream;
    // dynamic LDS for the refinement rows: 4 rows per match, 7 doubles per row, up to 144 KB
    static bool attr_once = ((void)hipFuncSetAttribute((const void*)k_vo_refine,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       144 * 1024),
                             (void)hipFuncSetAttribute((const void*)k_vo_refine_b,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       144 * 1024), true);
    (void)attr_once;
    int lds_rows = ((4 * N + 7) & ~7) + 8;   // (+ 8: the sums read one turn ahead)
    if ((size_t)lds_rows * 7 * sizeof(double) > 144 * 1024) lds_rows = 0;
    const VoRansacJob ar = {pm, N, samples, c, hyp_tr, hyp_count, hyp_flags, iters};
    const VoRefineJob af = {pm, N, iters, c, hyp_tr, hyp_count, hyp_flags, Jg, resg, lds_rows, out, out_inliers};
    const size_t lds = (size_t)lds_rows * 7 * sizeof(double);
    if (t_rec) {
        if (iters > 0) t_rec->add(b_vo_ransac, ar, (unsigned)iters);
        return t_rec->add(b_vo_refine, af, 1, 1, lds);
    }
    if (iters > 0) hipLaunchKernelGGL(k_vo_ransac, dim3(iters), dim3(64), 0, s, ar);
    hipLaunchKernelGGL(k_vo_refine, dim3(1), dim3(256), lds, s, af);
}

}  // namespace svh
