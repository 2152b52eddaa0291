// HIP kernels (gfx950) for the stages between the two matching phases of ELAS -- the part that
// round 1 ran on the host and that bounded throughput per host core:
//
//   k_lattice    E5/E6 + list  removeInconsistentSupportPoints, removeRedundantSupportPoints,
//                              the support list and addCornerSupportPoints
//                              (libelas/src/elas.cpp:174-318, 495-523)
//   k_delaunay   E7            Delaunay triangulation of the support points in left and right
//                              image coordinates with the OUTPUT ORDER of Triangle 1.6 "zQB"
//                              (libelas/src/triangle.cpp:5446-6230, 7800-7860; elas.cpp:534-600)
//   k_stage_pack               packed support / triangle lists + the group header the phase-B
//                              kernels read (what the host used to upload)
//
// With these the whole of Elas::process runs on the device without a host round trip in the
// middle; the host only enqueues.  Everything here is exact integer work.
//
// k_lattice.  The consistency filter of the reference scans the lattice u-major IN PLACE: a cell
// is dropped when fewer than incon_min_support cells of its window are (still) valid and similar,
// so an earlier drop lowers the count of a later cell.  That recurrence is well-founded (a cell
// depends on earlier cells only), hence it has exactly one solution, and the solution is the least
// fixed point of the monotone operator "drop x if (similar valid cells of the window) minus
// (similar cells EARLIER than x that are dropped) < need".  The kernel iterates that operator in
// place from "nothing dropped" until nothing changes: every intermediate state is a subset of the
// true drop set, so any evaluation order (512 lanes at once) converges to the reference's result.
// The two redundancy passes couple cells of one column (row) only: one lane walks one column (row)
// in the reference's order.
//
// k_delaunay.  Same algorithm as csrc/delaunay.cpp (Guibas-Stolfi divide and conquer, Dwyer's
// alternating cuts, Triangle's tie rules and record creation order), re-expressed for one
// workgroup per triangulation:
//   * ranks instead of sorts: support points are distinct integer points of the image, so the
//     rank of a point in (x,y) order is (points in lower columns) + (points of its column below
//     it): two histograms, a scan, and a count inside the column;
//   * the alternating-cut order by level-synchronous stable partitions (prefix counts);
//   * the recursion bottom-up by depth: all subproblems of one depth are independent, one lane
//     each; a subproblem of n vertices owns exactly 2n-2 triangle records, so every lane knows its
//     record range in advance and the records come out in the sequential creation order;
//   * a record carries the coordinates of its corners next to their indices, so an orientation /
//     in-circle test needs no second dependent load; records live in global memory (L2-resident,
//     ~100 KB per triangulation) -- the walk is latency-bound pointer chasing, one wave of work
//     next to the streaming kernels of other lanes, and takes no LDS away from them;
//   * coincident points (possible only for parameter combinations with candidate_stepsize <=
//     2*lr_threshold) make the survivor depend on the reference's pivot stream: the kernel flags
//     the pair and the engine reruns that group through the host path.
#include <hip/hip_runtime.h>

#include <mutex>

#include <algorithm>

#include "svh_internal.h"
#include "dt_core.h"

namespace svh {

namespace {

constexpr int16_t kInv = -32768;   // lattice cell that never held a candidate

// exclusive prefix sum over the block (kT threads); *total = sum of all.  s_tmp: kT/64 + 1 ints
template <int kT>
__device__ __forceinline__ int block_excl_scan(int v, int* s_tmp, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();   // s_tmp may still be read from a previous call
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kT / 64; w++) {
        const int t = s_tmp[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

struct LatticeParams {
    int W, H, Wc, Hc, step, ws, thr, need, add_corners, sup_cap;
};

// ---------------------------------------------------------------------------
// removeInconsistentSupportPoints (elas.cpp:174-209), event driven.  Cell states in val[]:
// >= 0 valid, kInv never valid, otherwise ~d = dropped (was d).
//   1. c0(x) = similar valid cells in the window of x on the untouched lattice (what the
//      reference would count if nothing had been dropped before it reaches x);
//   2. cells with c0 < need are dropped; every drop of a cell y lowers the count of the similar
//      cells x of its window that the scan reaches AFTER y (the reference no longer sees y
//      there), and a count that falls below `need` is the next drop.
// A count only ever loses cells that the reference's scan has dropped as well, so no cell is
// dropped early, and at the end every consequence has been propagated: the reference's lattice.
// Counters are bytes (a window of up to 15 x 15 cells), four to a word, lowered with one atomic.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lattice_consistency_events(int16_t* val, uint32_t* cntw, int32_t* wl, int nc,
                                                           const LatticeParams& P, int* s_n) {
    // cntw arrives filled: c0 of every cell, counted on the untouched lattice by k_lattice_count
    const int tid = threadIdx.x, Wc = P.Wc, Hc = P.Hc;
    int32_t* alive = wl;
    int32_t* fresh[2] = {wl + nc, wl + 2 * (size_t)nc};
    if (tid < 3) s_n[tid] = 0;
    __syncthreads();
    // (one LDS atomic per wave, not per lane: lanes of a wave append behind a common base)
    for (int i0 = 0; i0 < nc; i0 += 512) {
        const int i = i0 + tid;
        const bool on = i < nc && val[i] >= 0;
        const uint64_t m = __builtin_amdgcn_ballot_w64(on);
        int base = 0;
        if ((tid & 63) == 0 && m) base = atomicAdd(&s_n[0], __builtin_popcountll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (on) alive[base + __builtin_popcountll(m & ((1ull << (tid & 63)) - 1))] = i;
    }
    __syncthreads();
    const int nalive = s_n[0];
    for (int k = tid; k < nalive; k += 512) {
        const int at = alive[k];
        const int c0 = (cntw[at >> 2] >> (8 * (at & 3))) & 255;
        if (c0 < P.need) {
            val[at] = (int16_t)~val[at];
            fresh[0][atomicAdd(&s_n[1], 1)] = at;
        }
    }
    __syncthreads();
    for (int pr = 0;; pr ^= 1) {
        const int n = s_n[1 + pr];
        if (n == 0) break;
        __syncthreads();
        if (tid == 0) s_n[1 + (pr ^ 1)] = 0;
        __syncthreads();
        for (int k = tid; k < n; k += 512) {
            const int at = fresh[pr][k];
            const int vy = at / Wc, uy = at - vy * Wc, dy = ~(int)val[at];
            const int uhi = min(uy + P.ws, Wc - 1);
            const int vlo = max(vy - P.ws, 0), vhi = min(vy + P.ws, Hc - 1);
            for (int u2 = uy; u2 <= uhi; u2++)
                for (int v2 = (u2 == uy ? vy + 1 : vlo); v2 <= vhi; v2++) {
                    const int x = v2 * Wc + u2, dx = val[x];
                    const int df = dx > dy ? dx - dy : dy - dx;
                    if (dx < 0 || df > P.thr) continue;
                    const uint32_t old = atomicSub(&cntw[x >> 2], 1u << (8 * (x & 3)));
                    if ((int)((old >> (8 * (x & 3))) & 255) == P.need) {   // this drop takes it below
                        val[x] = (int16_t)~dx;
                        fresh[pr ^ 1][atomicAdd(&s_n[1 + (pr ^ 1)], 1)] = x;
                    }
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < nc; i += 512)
        if (val[i] < 0) val[i] = kInv;
    __syncthreads();
}

// The same filter for windows of more than 255 cells: sweeps of the monotone operator "drop x if
// (similar valid cells of the window) minus (similar cells scanned before x that are dropped) <
// need", in place, until nothing changes (least fixed point = the reference's result).
__device__ __forceinline__ void lattice_consistency_sweeps(int16_t* val, int nc, const LatticeParams& P) {
    const int tid = threadIdx.x, Wc = P.Wc, Hc = P.Hc;
    for (;;) {
        int changed = 0;
        for (int i = tid; i < nc; i += 512) {
            const int uc = i / Hc, vc = i - uc * Hc;   // u-major: a sweep carries drops forward
            const int at = vc * Wc + uc;
            const int dv = val[at];
            if (dv < 0) continue;
            const int ulo = max(uc - P.ws, 0), uhi = min(uc + P.ws, Wc - 1);
            const int vlo = max(vc - P.ws, 0), vhi = min(vc + P.ws, Hc - 1);
            int cnt = 0;
            for (int v2 = vlo; v2 <= vhi; v2++) {
                const int16_t* row = val + v2 * Wc;
                for (int u2 = ulo; u2 <= uhi; u2++) {
                    const int x = row[u2];
                    if (x == kInv) continue;
                    const int dx = x >= 0 ? x : ~x;
                    const int df = dx > dv ? dx - dv : dv - dx;
                    const bool earlier = u2 < uc || (u2 == uc && v2 < vc);
                    cnt += (df <= P.thr) && !(x < 0 && earlier);
                }
            }
            if (cnt < P.need) {
                val[at] = (int16_t)~dv;
                changed = 1;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    for (int i = tid; i < nc; i += 512)
        if (val[i] < 0) val[i] = kInv;
    __syncthreads();
}

// ---------------------------------------------------------------------------
// Step 1 of the consistency filter for the whole group, on as many workgroups as the lattice has
// tiles: c0(x) = similar valid cells in the (2 ws + 1)^2 window of x on the UNTOUCHED lattice.  This is
// the bulk of the filter's work (121 cells per candidate with the reference's window) and the only
// part without a dependence between cells; k_lattice, one workgroup per pair, starts from these
// counts.  One byte per cell (windows of at most 255 cells), the layout of k_lattice's packed words.
// ---------------------------------------------------------------------------
constexpr int LCX = 64, LCY = 16, LCH = 7;   // tile and the largest halo (window 15 x 15)
__global__ __launch_bounds__(256) void k_lattice_count(StageDev S, LatticeParams P) {
    __shared__ int16_t s_t[LCY + 2 * LCH][LCX + 2 * LCH];
    const int pair = blockIdx.z, nc = P.Wc * P.Hc;
    const int16_t* raw = S.dcan + (size_t)pair * nc;
    uint8_t* cnt8 = reinterpret_cast<uint8_t*>(S.cntw + (size_t)pair * ((nc + 3) / 4));
    const int u0 = blockIdx.x * LCX, v0 = blockIdx.y * LCY;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int ws = P.ws, tw = LCX + 2 * ws, th = LCY + 2 * ws;
    {
        constexpr int NE = ((LCY + 2 * LCH) * (LCX + 2 * LCH) + 255) / 256;
        int16_t r[NE];
#pragma unroll
        for (int k = 0; k < NE; k++) {   // all loads of a thread in flight before the first store
            const int i = tid + 256 * k, y = i / tw, x = i - y * tw;
            const int v = v0 - ws + y, u = u0 - ws + x;
            r[k] = (i < th * tw && v >= 0 && v < P.Hc && u >= 0 && u < P.Wc) ? raw[v * P.Wc + u] : (int16_t)-1;
        }
#pragma unroll
        for (int k = 0; k < NE; k++) {
            const int i = tid + 256 * k, y = i / tw, x = i - y * tw;
            if (i < th * tw) s_t[y][x] = r[k] < 0 ? kInv : r[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LCY / 4; k++) {
        const int ty = threadIdx.y + 4 * k, tx = threadIdx.x;
        const int v = v0 + ty, u = u0 + tx;
        if (v >= P.Hc || u >= P.Wc) continue;
        const int dv = s_t[ty + ws][tx + ws];
        int cnt = 0;
        if (dv >= 0)
            for (int y = 0; y <= 2 * ws; y++)
                for (int x = 0; x <= 2 * ws; x++) {
                    const int c = s_t[ty + y][tx + x];      // cells outside the lattice hold kInv
                    const int df = c > dv ? c - dv : dv - c;
                    cnt += (c >= 0) & (df <= P.thr);
                }
        cnt8[v * P.Wc + u] = (uint8_t)cnt;
    }
}

// ---------------------------------------------------------------------------
// E5 / E6 / list / corners.  One block per pair.  kLds: lattice and counters fit LDS (3 bytes per
// cell); otherwise the lattice is filtered in place in global memory (very large images).
// ---------------------------------------------------------------------------
template <bool kLds>
__global__ __launch_bounds__(512) void k_lattice(StageDev S, LatticeParams P) {
    extern __shared__ int16_t s_dyn[];
    __shared__ int s_scan[512 / 64 + 1];
    __shared__ int s_n[4];
    __shared__ unsigned long long s_best[4];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int Wc = P.Wc, Hc = P.Hc, nc = Wc * Hc;
    int16_t* raw = S.dcan + (size_t)pair * nc;
    int16_t* val = kLds ? s_dyn : raw;
    const int ncw = (nc + 3) / 4;
    uint32_t* cntw = kLds ? reinterpret_cast<uint32_t*>(s_dyn + ((nc + 1) & ~1)) : S.cntw + (size_t)pair * ncw;
#define STAMP(k) do { if (pair == 0 && tid == 0) S.counts->dbg[k] = wall_clock64(); } while (0)
    STAMP(0);
    for (int i = tid; i < nc; i += 512) {
        const int16_t r = raw[i];
        val[i] = r < 0 ? kInv : r;
    }
    __syncthreads();
    STAMP(1);
    if (P.need > 0) {
        if ((2 * P.ws + 1) * (2 * P.ws + 1) <= 255 && P.need <= 255) {
            if (kLds) {
                const uint32_t* cg = S.cntw + (size_t)pair * ncw;     // counts of k_lattice_count
                for (int i = tid; i < ncw; i += 512) cntw[i] = cg[i];
            }
            lattice_consistency_events(val, cntw, S.wl + (size_t)pair * 3 * nc, nc, P, s_n);
        }
        else {
            lattice_consistency_sweeps(val, nc, P);
        }
    }
    STAMP(2);
    // ---- removeRedundantSupportPoints (elas.cpp:213-279 as called at :501-502: distance 5,
    // threshold 1), vertical then horizontal.  A pass couples the cells of one column (row) only:
    // one lane walks one line in the reference's order with the five cells behind it (as the
    // pass left them) and the five ahead (untouched) in registers.
    for (int pass = 0; pass < 2; pass++) {
        const int lines = pass == 0 ? Wc : Hc, steps = pass == 0 ? Hc : Wc;
        const int ls = pass == 0 ? 1 : Wc, ss = pass == 0 ? Wc : 1;   // strides: line, step
        for (int l = tid; l < lines; l += 512) {
            int16_t* line = val + l * ls;
            int b[5], a[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                b[j] = kInv;
                a[j] = 1 + j < steps ? line[(1 + j) * ss] : kInv;
            }
            int cur = line[0];
            for (int t = 0; t < steps; t++) {
                const int nxt = t + 6 < steps ? line[(t + 6) * ss] : kInv;
                int dv = cur;
                if (dv >= 0) {
                    // "valid and within 1 of dv" = (unsigned)(x - dv + 1) <= 2: an invalid cell is -32768
                    bool fb = false, fa = false;
                    const int t1 = 1 - dv;
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        fb |= (unsigned)(b[j] + t1) <= 2u;
                        fa |= (unsigned)(a[j] + t1) <= 2u;
                    }
                    if (fb && fa) {
                        dv = kInv;
                        line[t * ss] = kInv;
                    }
                }
#pragma unroll
                for (int j = 4; j > 0; j--) b[j] = b[j - 1];
                b[0] = dv;
                cur = a[0];
#pragma unroll
                for (int j = 0; j < 4; j++) a[j] = a[j + 1];
                a[4] = nxt;
            }
        }
        __syncthreads();
    }
    STAMP(3);
    // ---- list, u-major from (1,1) (elas.cpp:505-517)
    const int Hm = Hc - 1, tot = (Wc - 1) * Hm;
    const int chunk = (max(tot, 0) + 511) / 512;
    const int i0 = tid * chunk, i1 = min(tot, i0 + chunk);
    int mine = 0;
    for (int i = i0; i < i1; i++) {
        const int uc = 1 + i / Hm, vc = 1 + i - (uc - 1) * Hm;
        mine += val[vc * Wc + uc] >= 0;
    }
    int n = 0;
    int pos = block_excl_scan<512>(mine, s_scan, &n);
    int32_t* sup = S.sup_raw + (size_t)pair * 3 * P.sup_cap;
    const bool room = n + (P.add_corners ? 6 : 0) <= P.sup_cap;
    if (room)
        for (int i = i0; i < i1; i++) {
            const int uc = 1 + i / Hm, vc = 1 + i - (uc - 1) * Hm;
            const int dv = val[vc * Wc + uc];
            if (dv >= 0) {
                sup[3 * pos] = uc * P.step;
                sup[3 * pos + 1] = vc * P.step;
                sup[3 * pos + 2] = dv;
                pos++;
            }
        }
    // ---- addCornerSupportPoints (elas.cpp:283-318): disparity of the nearest support point
    // (first one on ties: smallest (distance, index) key)
    if (P.add_corners && room) {
        if (tid < 4) s_best[tid] = ~0ull;
        __syncthreads();   // also makes the list visible
        const int cu[4] = {0, 0, P.W - 1, P.W - 1};
        const int cv[4] = {0, P.H - 1, 0, P.H - 1};
        unsigned long long best[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        for (int j = tid; j < n; j += 512) {
            const int su = sup[3 * j], sv = sup[3 * j + 1];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const long long du = cu[c] - su, dw = cv[c] - sv;
                const unsigned long long dist = (unsigned long long)(du * du + dw * dw);
                if (dist < 10000000ull) {
                    const unsigned long long key = dist << 32 | (unsigned)j;
                    best[c] = key < best[c] ? key : best[c];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (best[c] != ~0ull) atomicMin(&s_best[c], best[c]);
        __syncthreads();
        if (tid == 0) {
            int cd[4];
            for (int c = 0; c < 4; c++)
                cd[c] = s_best[c] == ~0ull ? 0 : sup[3 * (int)(s_best[c] & 0xffffffffu) + 2];
            for (int c = 0; c < 4; c++) {
                sup[3 * (n + c)] = cu[c];
                sup[3 * (n + c) + 1] = cv[c];
                sup[3 * (n + c) + 2] = cd[c];
            }
            for (int c = 2; c < 4; c++) {   // the two right-image corners
                sup[3 * (n + 2 + c)] = cu[c] + cd[c];
                sup[3 * (n + 2 + c) + 1] = cv[c];
                sup[3 * (n + 2 + c) + 2] = cd[c];
            }
        }
        n += 6;
    }
    STAMP(4);
#undef STAMP
    if (tid == 0) {
        S.counts->nsup[pair] = room ? n : 0;
        S.counts->flags[pair] = room ? (n < 3 ? STG_FEW : 0) : STG_OVERFLOW;
    }
}

// ===========================================================================
// Delaunay
// ===========================================================================
// Triangle records, leaves and the merge of two halves: csrc/dt_core.h (shared with the CPU check
// tests/cxx/dt_core_check.cpp).  Round 5: records are fetched whole, their corners carry the coordinates.
using namespace dt;

struct DtParams {
    int lds_ints;                 // ints of dynamic LDS of a k_delaunay block
    int lds_cap;                  // points whose records fit the block's LDS (48 bytes per point)
    int xoff;                     // added to x: the left corner points lie at -d in the right image
    int W, H, sup_cap, rec_cap;   // W: columns the points may use (image width + disp_max: the two
                                  // right-image corner points of addCornerSupportPoints lie at W-1+d)
    int spread;                   // depths with at most this many nodes give consecutive nodes to different waves
    int uniform;                  // 1: depths with no more nodes than waves run one node per wave on all lanes (scalar walk)
};

// The recursion of the divide and conquer, bottom-up: all nodes of one depth are independent (one
// lane each), leaves at `depth` first, the root last.  FL / FR: hull handles (farleft, farright)
// of the nodes of a depth, by first vertex, two depths alternating.
template <int kT, class M>
__device__ __forceinline__ void dt_build(const M& mesh, int m, int depth, const int* order, const int* oxy,
                                         unsigned* FL, unsigned* FR, int sup_cap, int64_t* dbg, bool stamp, int spread, int uniform) {
    const int tid = threadIdx.x;
    // (block-uniform by construction; said explicitly, they come out of __syncthreads_or loops)
    m = __builtin_amdgcn_readfirstlane(m);
    depth = __builtin_amdgcn_readfirstlane(depth);
    if (tid == 0) mesh.make_rec(0);   // record 0 = outer space
    __syncthreads();
    for (int d = depth; d >= 0; d--) {
        unsigned* fl = FL + (size_t)(d & 1) * sup_cap;
        unsigned* fr = FR + (size_t)(d & 1) * sup_cap;
        const unsigned* cfl = FL + (size_t)((d + 1) & 1) * sup_cap;
        const unsigned* cfr = FR + (size_t)((d + 1) & 1) * sup_cap;
        const unsigned tasks = 1u << d;
        // Which lane takes which node.  A merge is ~250 instructions per seam step on ONE lane, and two merges on
        // lanes of the same wave branch apart at every step (flip or not, which side the new seam edge takes), so
        // they largely run one after the other.  Near the root, where a depth has only a few nodes, consecutive
        // nodes therefore go to DIFFERENT waves (lane l of wave w takes node l * waves + w): the waves sit on
        // different SIMDs and run side by side.  Deep in the tree (hundreds of short merges of similar shape) the
        // lanes of a wave share most of the code and the dense assignment is the faster one (round 3: spreading
        // EVERY depth of a KITTI-size set 634 -> 744 us; large sets with their records in L2 spread always).
        constexpr unsigned kWaves = kT / 64;
        const bool spr = spread < 0 || tasks <= (unsigned)spread;
        const unsigned j0 = spr ? (unsigned)(tid & 63) * kWaves + (unsigned)(tid >> 6) : (unsigned)tid;
        if (uniform > 0 && tasks <= kWaves * (unsigned)uniform) {
            // P.uniform = n > 0 (SVH_DT_UNIFORM; default: 1 for latency-bound calls, see launch_stage_device): a depth with at most n nodes per wave: wave w runs nodes w, w + waves, ..
            // one after the other on ALL its lanes with identical operands -- every address and every branch is
            // wave-uniform, so the seam walk compiles to scalar code (s_cbranch instead of exec-mask bookkeeping,
            // SALU arithmetic, v_readfirstlane behind every record read): 1.46 x faster per merge than one lane.
            // Measured (profiles/r05_delaunay_scalar_walk.txt): n = 1, 512 threads 672 -> 597 us per 64
            // triangulations (1024 threads 586), n = 2 618, n = 4 710 (two divergent lanes share half their
            // instructions, two scalar merges in a row share nothing) -- but the PIPELINE loses 1.2 % with it
            // (34.3 -> 33.9 k pairs/s, x 3): the scalar unit of a SIMD is shared with the matchers' waves, whose
            // loop control and waits then queue behind a wave that issues scalar instructions back to back.
            // Default: on for a group that has the device to itself, off otherwise (launch_stage_device).
            const unsigned jw = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
            // (all 64 lanes run it; with one lane active the same scalar code measured SLOWER: 640 vs 597 us)
            for (unsigned j = jw; j < tasks; j += kWaves) dt_node<true, M, true>(mesh, m, d, j, order, oxy, cfl, cfr, fl, fr);
        } else {
        for (unsigned j = j0; j < tasks; j += kT) dt_node<true>(mesh, m, d, j, order, oxy, cfl, cfr, fl, fr);
        }
        __syncthreads();
        if (stamp && tid == 0 && 11 + (depth - d) < 30) dbg[11 + (depth - d)] = wall_clock64();
    }
}

// ---------------------------------------------------------------------------
// One block of kT threads per (pair, side).  The dynamic LDS block is used three times over:
//   ranks      column / row histograms and cursors; behind them the packed coordinates and the two
//              bucket lists when they fit (the rank of a point walks its column and its row bucket:
//              dependent reads, ~10 per point -- from LDS instead of L2);
//   cut order  the two rank lists, the partition buffer and the prefix counts (16 bytes per point);
//   build      the triangle records (48 bytes per point: two 24-byte records) when they fit.
// kT = 256 for KITTI-size lattices; 1024 for large ones (1920x1080: 6-12 k points per side), where
// the chunked loops of the first two phases are 4x shorter per thread.
// ---------------------------------------------------------------------------
// kPhase 0: the whole triangulation in one launch.  Large point sets (1920x1080: 6-12 k points per side, records in
// L2) run it as TWO launches since round 5: kPhase 1 = ranks + cut order (the part that wants the CU's whole LDS,
// ~230 us) and kPhase 2 = the bottom-up build (~1.3 ms of pointer walking that needs no LDS at all) -- as one kernel
// a triangulation held 159 KB of LDS, i.e. a whole CU, for its entire duration.
template <int kT, int kPhase>
__global__ __launch_bounds__(kT) void k_delaunay(StageDev S, DtParams P) {
    // (16-byte aligned: MeshL reads whole 8-byte records through this base -- dt_core.h -- whatever static LDS precedes it)
    extern __shared__ __attribute__((aligned(16))) int s_hist[];   // [W + 1] column starts, [H + 1] row starts, then two cursors copies
    __shared__ int s_scan[kT / 64 + 1];
    __shared__ int s_m;               // points left after coincident ones were dropped (-1: stack overflow)
    const int slot = blockIdx.x, pair = slot >> 1, side = slot & 1, tid = threadIdx.x;
    int m = S.counts->nsup[pair];
    const int m_all = m;              // k_stage_pack walks the records of m_all points
    if (kPhase == 1 && tid == 0) S.counts->dt_m[slot] = 0;     // (every early exit below: nothing to build)
    if (m < 3 || (S.counts->flags[pair] & STG_OVERFLOW)) {
        if (tid == 0) S.counts->ntri[slot] = 0;
        return;
    }
    const size_t so = (size_t)slot * P.sup_cap;
    int* dmap = S.dmap + so;
    bool remap = false;               // point p of the triangulation is support point dmap[p]
    const int32_t* sup = S.sup_raw + (size_t)pair * 3 * P.sup_cap;
    int* pxy = S.pxy + so;
    int* bx = S.buck + so;
    int* by = S.buck2 + so;
    int* byx = S.byx + so;
    unsigned* lx = S.lx + so;
    unsigned* ly = S.ly + so;
    unsigned* tmp = S.tmp + so;
    unsigned* Pc = S.P + so;
    int* order = S.order + so;
    int* oxy = S.oxy + so;
    int* colstart = s_hist;                  // W + 1
    int* rowstart = s_hist + P.W + 1;        // H + 1
    int* curx = rowstart + P.H + 1;          // W
    int* cury = curx + P.W;                  // H
    const int nh = 2 * (P.W + P.H) + 2;
#define STAMP(k) do { if (slot == 0 && tid == 0) S.counts->dbg[k] = wall_clock64(); } while (0)
    int depth = 0;
    if (kPhase == 2) {
        // the ordering launch left order / oxy / pxy / dmap in memory and these three numbers
        m = S.counts->dt_m[slot];
        if (m < 3) return;             // (it has set ntri and the records of a slot without triangles)
        const int dd = S.counts->dt_depth[slot];
        depth = dd & 0xffff;
        remap = (dd & 0x10000) != 0;
    } else {
    STAMP(8);
    // ---- ranks in (x,y) and (y,x) order.  Second trip only when coincident points were found and dropped.
    for (;;) {
    for (int i = tid; i < nh; i += kT) s_hist[i] = 0;
    __syncthreads();
    const bool ldsR = nh + 3 * m <= P.lds_ints;          // coordinates + bucket lists next to the histograms
    int* lpxy = ldsR ? s_hist + nh : pxy;
    int* lbx = ldsR ? s_hist + nh + m : bx;
    int* lby = ldsR ? s_hist + nh + 2 * m : by;
    int bad = 0;
    for (int p = tid; p < m; p += kT) {
        int x, y;
        if (!remap) {
            x = (side ? sup[3 * p] - sup[3 * p + 2] : sup[3 * p]) + P.xoff;
            y = sup[3 * p + 1];
        } else {
            x = pxy[p] & 0xffff;
            y = pxy[p] >> 16;
        }
        if (x < 0 || x >= P.W || y < 0 || y >= P.H) {
            bad = 1;
            continue;
        }
        pxy[p] = x | y << 16;
        if (ldsR) lpxy[p] = x | y << 16;
        atomicAdd(&colstart[x + 1], 1);
        atomicAdd(&rowstart[y + 1], 1);
    }
    if (__syncthreads_or(bad)) {
        if (tid == 0) {
            atomicOr(&S.counts->flags[pair], STG_OVERFLOW);
            S.counts->ntri[slot] = 0;
        }
        return;
    }
    // inclusive scans of the two histograms (entry k+1 holds the count of column k)
    for (int which = 0; which < 2; which++) {
        int* h = which ? rowstart : colstart;
        const int len = (which ? P.H : P.W) + 1;
        const int chunk = (len + kT - 1) / kT, a = tid * chunk, b = min(len, a + chunk);
        int sum = 0;
        for (int i = a; i < b; i++) sum += h[i];
        int tot;
        int run = block_excl_scan<kT>(sum, s_scan, &tot);
        for (int i = a; i < b; i++) {
            run += h[i];
            h[i] = run;
        }
    }
    __syncthreads();
    for (int i = tid; i < P.W; i += kT) curx[i] = colstart[i];
    for (int i = tid; i < P.H; i += kT) cury[i] = rowstart[i];
    __syncthreads();
    for (int p = tid; p < m; p += kT) {
        const int xy = lpxy[p], x = xy & 0xffff, y = xy >> 16;
        lbx[atomicAdd(&curx[x], 1)] = p;
        lby[atomicAdd(&cury[y], 1)] = p;
    }
    __syncthreads();
    int dup = 0;
    for (int p = tid; p < m; p += kT) {
        const int xy = lpxy[p], x = xy & 0xffff, y = xy >> 16;
        int xr = colstart[x], yr = rowstart[y];
        for (int k = colstart[x]; k < colstart[x + 1]; k++) {
            const int q = lbx[k], qy = lpxy[q] >> 16;
            xr += qy < y;
            dup |= (qy == y) & (q != p);
        }
        for (int k = rowstart[y]; k < rowstart[y + 1]; k++) yr += (lpxy[lby[k]] & 0xffff) < x;
        const unsigned e = (unsigned)xr | (unsigned)yr << 16;
        lx[xr] = e;
        ly[yr] = e;
        byx[xr] = p;
    }
    if (!__syncthreads_or(dup)) break;
    // ---- Coincident points (possible with candidate_stepsize <= 2 * lr_threshold: two support points
    // of a row that land on the same right-image pixel).  Triangle sorts the vertices with a quicksort
    // whose pivots come from its own generator (vertexsort / randomnation, triangle.cpp:5446-5476,
    // 4045-4049; the mesh's seed starts at 1) and then keeps the FIRST vertex of every run of equal
    // ones (triangle.cpp:6179-6196): which of two coincident points survives is a property of that
    // pivot stream.  One lane replays it -- same generator, same Hoare partition, same order of the
    // recursive calls (a node, its left part, its right part) -- on packed keys x << 14 | y with the
    // point index riding along, drops the repeats, and the rank phase runs again on the survivors
    // (distinct points: everything after this is independent of the pivots).
    for (int p = tid; p < m; p += kT) {
        const int xy = pxy[p];
        tmp[p] = (unsigned)((xy & 0xffff) << 14 | (xy >> 16));
        dmap[p] = p;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned* key = tmp;
        unsigned* stk = Pc;                     // (start, count) pairs
        const int stk_cap = P.sup_cap / 2;
        unsigned seed = 1;
        int top = 0;
        bool ok = true;
        stk[0] = 0; stk[1] = (unsigned)m; top = 1;
        while (top > 0 && ok) {
            top--;
            const int s0 = (int)stk[2 * top], n = (int)stk[2 * top + 1];
            unsigned* a = key + s0;
            int* ai = dmap + s0;
            if (n == 2) {
                if (a[0] > a[1]) {
                    const unsigned t = a[0]; a[0] = a[1]; a[1] = t;
                    const int ti = ai[0]; ai[0] = ai[1]; ai[1] = ti;
                }
                continue;
            }
            seed = (seed * 1366u + 150889u) % 714025u;
            const unsigned pv = a[seed / (714025u / (unsigned)n + 1u)];
            int l = -1, r = n;
            while (l < r) {
                do { l++; } while (l <= r && a[l] < pv);
                do { r--; } while (l <= r && a[r] > pv);
                if (l < r) {
                    const unsigned t = a[l]; a[l] = a[r]; a[r] = t;
                    const int ti = ai[l]; ai[l] = ai[r]; ai[r] = ti;
                }
            }
            // children, pushed right first so that the left one is sorted next (the reference recurses
            // into the left part before the right part; the generator advances once per partition)
            if (top + 2 > stk_cap) { ok = false; break; }
            if (r < n - 2) { stk[2 * top] = (unsigned)(s0 + r + 1); stk[2 * top + 1] = (unsigned)(n - r - 1); top++; }
            if (l > 1) { stk[2 * top] = (unsigned)s0; stk[2 * top + 1] = (unsigned)l; top++; }
        }
        int j = 0;
        for (int i = 1; i < m; i++)
            if (key[i] != key[j]) {
                j++;
                key[j] = key[i];
                dmap[j] = dmap[i];
            }
        s_m = ok ? j + 1 : -1;
    }
    __syncthreads();
    if (s_m < 0 || remap) {
        // (a stack deeper than the scratch holds, or repeats among distinct points: cannot happen; the
        // host path would decide)
        if (tid == 0) {
            atomicOr(&S.counts->flags[pair], STG_DUP);
            S.counts->ntri[slot] = 0;
        }
        return;
    }
    m = s_m;
    for (int p = tid; p < m; p += kT) pxy[p] = (int)(tmp[p] >> 14) | (int)(tmp[p] & 0x3fffu) << 16;
    remap = true;
    __syncthreads();
    if (m < 3) {
        // fewer than three distinct vertices: no triangle
        for (int t = 1 + tid; t < 2 * m_all - 1; t += kT)
            *reinterpret_cast<int4*>(S.ids + (size_t)slot * 4 * P.rec_cap + 4 * (size_t)t) = make_int4(-1, -1, -1, 0);
        if (tid == 0) S.counts->ntri[slot] = 0;
        return;
    }
    }

    STAMP(9);
    // ---- alternating-cut order (triangle.cpp:5582-5604): level-synchronous stable partitions.
    // lx is sorted by x rank, ly by y rank, inside every segment; a cut by axis a takes the lower
    // half of the a-list and splits the other list the same way, stably.
    const int chunk = (m + kT - 1) / kT, c0 = tid * chunk, c1 = min(m, c0 + chunk);
    if (4 * m <= P.lds_ints) {
        // the two lists, the partition buffer and the prefix counts move to LDS (the histograms and
        // the bucket lists are dead; lx / ly were written to memory so that nothing aliased them)
        unsigned* l0 = reinterpret_cast<unsigned*>(s_hist);
        for (int i = c0; i < c1; i++) {
            l0[i] = lx[i];
            l0[m + i] = ly[i];
        }
        lx = l0; ly = l0 + m; tmp = l0 + 2 * m; Pc = l0 + 3 * m;
        __syncthreads();
    }
    for (;; depth++) {
        const int axis = depth & 1;
        unsigned* src = axis == 0 ? ly : lx;
        const unsigned* oth = axis == 0 ? lx : ly;
        const int sh = axis == 0 ? 0 : 16;      // key of the cut: x rank (low half) or y rank
        int cnt = 0, split = 0;
        for (int i = c0; i < c1; i++) {
            int s, n;
            dt_segment(m, depth, i, &s, &n);
            if (n <= 3) continue;
            split = 1;
            const unsigned pivot = (oth[s + (n >> 1)] >> sh) & 0xffffu;
            cnt += ((src[i] >> sh) & 0xffffu) < pivot;
        }
        if (!__syncthreads_or(split)) break;
        int tot;
        int run = block_excl_scan<kT>(cnt, s_scan, &tot);
        for (int i = c0; i < c1; i++) {
            int s, n;
            dt_segment(m, depth, i, &s, &n);
            unsigned low = 0;
            if (n > 3) {
                const unsigned pivot = (oth[s + (n >> 1)] >> sh) & 0xffffu;
                low = ((src[i] >> sh) & 0xffffu) < pivot;
            }
            Pc[i] = (unsigned)run | low << 31;
            run += (int)low;
        }
        __syncthreads();
        for (int i = c0; i < c1; i++) {
            int s, n;
            dt_segment(m, depth, i, &s, &n);
            int to = i;
            if (n > 3) {
                const unsigned pi = Pc[i];
                const int before = (int)(pi & 0x7fffffffu) - (int)(Pc[s] & 0x7fffffffu);   // lows in [s, i)
                to = (pi >> 31) ? s + before : s + (n >> 1) + (i - s - before);
            }
            tmp[to] = src[i];
        }
        __syncthreads();
        // rotate: the partitioned copy becomes the list
        if (axis == 0) { unsigned* t = ly; ly = tmp; tmp = t; }
        else           { unsigned* t = lx; lx = tmp; tmp = t; }
    }
    STAMP(10);
    for (int i = c0; i < c1; i++) {
        const int p = byx[lx[i] & 0xffffu];
        order[i] = p;
        oxy[i] = pxy[p];
    }
    if (kPhase == 1) {
        if (tid == 0) {
            S.counts->dt_m[slot] = m;
            S.counts->dt_depth[slot] = depth | (remap ? 0x10000 : 0);
        }
        return;
    }
    }   // (kPhase != 2)
    // ---- divide and conquer, bottom-up by depth (`depth` is where every node is a leaf), in LDS
    // when the records of this triangulation fit the block's allocation
    MeshG mg;
    mg.ids = S.ids + (size_t)slot * 4 * P.rec_cap;
    mg.xys = S.xys + (size_t)slot * 4 * P.rec_cap;
    mg.nbr = S.nbr + (size_t)slot * 4 * P.rec_cap;
    unsigned* FL = S.fl + (size_t)slot * 2 * P.sup_cap;
    unsigned* FR = S.fr + (size_t)slot * 2 * P.sup_cap;
    __syncthreads();   // the histograms are dead: the LDS block is reused for the records
    if (m <= P.lds_cap) {
        MeshL ml;
        ml.base = reinterpret_cast<unsigned char*>(s_hist);      // 24 bytes per record, 2m + 2 records
        dt_build<kT>(ml, m, depth, order, oxy, FL, FR, P.sup_cap, S.counts->dbg, slot == 0, P.spread, P.uniform);
        // corner indices out for k_stage_pack (records 1 .. 2m-2)
        // (after coincident points were dropped the triangulation's point p is support point dmap[p])
        auto sid = [&](int a) { return a < 0 ? -1 : (remap ? dmap[a] : a); };
        for (int t = 1 + tid; t < 2 * m - 1; t += kT) {
            const Rec rc = ml.load((unsigned)t * 4u);
            *reinterpret_cast<int4*>(mg.ids + 4 * (size_t)t) = make_int4(sid(rc.id0), sid(rc.id1), sid(rc.id2), 0);
        }
        __syncthreads();
    } else {
        dt_build<kT>(mg, m, depth, order, oxy, FL, FR, P.sup_cap, S.counts->dbg, slot == 0, P.spread, P.uniform);
        if (remap) {
            for (int t = 1 + tid; t < 2 * m - 1; t += kT) {
                int4 v = *reinterpret_cast<const int4*>(mg.ids + 4 * (size_t)t);
                v.x = v.x >= 0 ? dmap[v.x] : -1;
                v.y = v.y >= 0 ? dmap[v.y] : -1;
                v.z = v.z >= 0 ? dmap[v.z] : -1;
                *reinterpret_cast<int4*>(mg.ids + 4 * (size_t)t) = v;
            }
            __syncthreads();
        }
    }
    if (remap) {
        // fewer points, fewer records: what lies behind them is a leftover of an earlier group
        for (int t = 2 * m - 1 + tid; t < 2 * m_all - 1; t += kT)
            *reinterpret_cast<int4*>(mg.ids + 4 * (size_t)t) = make_int4(-1, -1, -1, 0);
        __syncthreads();
    }
    const MeshG& mesh = mg;
    // ---- surviving records = those without the ghost corner (the hull fan dies,
    // triangle.cpp:7800-7860); k_stage_pack writes them out in creation order
    const int nrec = 2 * m - 1;   // records 1 .. 2m-2
    int live = 0;
    for (int t = 1 + tid; t < nrec; t += kT) {
        const int4 v = *reinterpret_cast<const int4*>(mesh.ids + 4 * (size_t)t);
        live += (v.x >= 0) & (v.y >= 0) & (v.z >= 0);
    }
    int tot;
    block_excl_scan<kT>(live, s_scan, &tot);
    if (tid == 0) S.counts->ntri[slot] = tot;
    STAMP(30);
#undef STAMP
}

// ---------------------------------------------------------------------------
// Packed lists + group header (what the host used to build and upload).  One block per slot.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stage_pack(StageDev S, DtParams P, int g, GroupHdr* hdr, int32_t* support,
                                                    int32_t* tri) {
    __shared__ int s_scan[256 / 64 + 1];
    const int slot = blockIdx.x, pair = slot >> 1, side = slot & 1, tid = threadIdx.x;
    // offsets from the counts of all slots (<= 32 values)
    int sup_off = 0, tri_off = 0, total_sup = 0, total_tri = 0;
    bool active_me = false;
    for (int j = 0; j < g; j++) {
        const bool act = S.counts->nsup[j] >= 3 && !(S.counts->flags[j] & (STG_DUP | STG_OVERFLOW));
        const int ns = act ? S.counts->nsup[j] : 0;
        if (j < pair) sup_off += ns;
        if (j == pair) active_me = act;
        total_sup += ns;
        for (int k = 0; k < 2; k++) {
            const int nt = act ? S.counts->ntri[2 * j + k] : 0;
            if (2 * j + k < slot) tri_off += nt;
            total_tri += nt;
        }
    }
    if (slot == 0 && tid == 0) {
        hdr->npairs = g;
        int so = 0, te = 0;
        for (int j = 0; j < g; j++) {
            const bool act = S.counts->nsup[j] >= 3 && !(S.counts->flags[j] & (STG_DUP | STG_OVERFLOW));
            hdr->active[j] = act ? 1 : 0;
            hdr->sup_off[j] = so;
            so += act ? S.counts->nsup[j] : 0;
            for (int k = 0; k < 2; k++) {
                te += act ? S.counts->ntri[2 * j + k] : 0;
                hdr->tri_end[2 * j + k] = te;
            }
        }
        hdr->sup_off[g] = so;
        hdr->total_sup = total_sup;
        hdr->total_tri = total_tri;
    }
    if (!active_me) return;
    const int m = S.counts->nsup[pair];
    if (side == 0) {
        const int32_t* src = S.sup_raw + (size_t)pair * 3 * P.sup_cap;
        int32_t* dst = support + 3 * (size_t)sup_off;
        for (int i = tid; i < 3 * m; i += 256) dst[i] = src[i];
    }
    // ordered compaction of the surviving records: corners (org, dest, apex) of orientation 0
    const int* ids = S.ids + (size_t)slot * 4 * P.rec_cap;
    const int nrec = 2 * m - 1;
    const int chunk = (nrec + 255) / 256, a = max(1, tid * chunk), b = min(nrec, (tid + 1) * chunk);
    int live = 0;
    for (int t = a; t < b; t++) {
        const int4 v = *reinterpret_cast<const int4*>(ids + 4 * (size_t)t);
        live += (v.x >= 0) & (v.y >= 0) & (v.z >= 0);
    }
    int tot;
    int pos = block_excl_scan<256>(live, s_scan, &tot);
    int32_t* out = tri + 3 * (size_t)tri_off;
    for (int t = a; t < b; t++) {
        const int4 v = *reinterpret_cast<const int4*>(ids + 4 * (size_t)t);
        if ((v.x >= 0) & (v.y >= 0) & (v.z >= 0)) {
            out[3 * pos] = v.y;
            out[3 * pos + 1] = v.z;
            out[3 * pos + 2] = v.x;
            pos++;
        }
    }
}

struct Timed {
    Profiler* p;
    Timed(const LaunchCtx& cx, const char* name) : p(cx.prof) {
        if (p) p->begin(name);
    }
    ~Timed() {
        if (p) p->end();
    }
};

}  // namespace

// columns the points of a triangulation may use: x + disp_max in [0, W + 2 disp_max] (corner points
// of addCornerSupportPoints: -d in the right image, W-1+d in the left one)
static int dt_columns(const svh_elas_params& p, const Dims& d) { return d.W + 2 * std::max(p.disp_max, 0) + 1; }
// large lattices (1920x1080: 83 k cells, 6-12 k support points per side) get the whole LDS of a CU: the
// rank lists of the cut-order phase take 16 bytes per point
static bool dt_large(const Dims& d) { return (size_t)d.Wc * d.Hc / 8 * 16 > 63 * 1024; }
// More than 64 KB of dynamic LDS needs an opt-in per kernel and device: asked once per device; a refusal (a part
// with 64 KB of LDS per workgroup) sends large lattices through the 256-thread / 63 KB form and keeps the small
// form at 63 KB (records of larger point sets then live in L2).
// large point sets: ordering and build as two launches (SVH_DT_SPLIT=0: one launch, round 4's form)
static bool dt_split() {
    static const bool on = !(svh::env("SVH_DT_SPLIT") && atoi(svh::env("SVH_DT_SPLIT")) == 0);
    return on;
}
static int dt_small_threads() {
    static const int t = svh::env("SVH_DT_THREADS") ? atoi(svh::env("SVH_DT_THREADS")) : 512;
    return t == 512 || t == 1024 ? t : 256;
}
static bool dt_lds_optin(bool big, size_t bytes) {
    static std::mutex mu;
    static int state[2][64];   // 0 unknown, 1 granted, 2 refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (state[big][dev] == 0) {
        const int t = big ? 1024 : dt_small_threads();
        const void* fn = t == 1024 ? (big && dt_split() ? (const void*)k_delaunay<1024, 1> : (const void*)k_delaunay<1024, 0>)
                                   : (t == 512 ? (const void*)k_delaunay<512, 0> : (const void*)k_delaunay<256, 0>);
        // (the small form on 1024 threads shares the kernel with the large one: the larger request stands)
        const bool ok = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)(t == 1024 ? 159 * 1024 : bytes)) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        state[big][dev] = ok ? 1 : 2;
    }
    return state[big][dev] == 1;
}
// KITTI-size lattices: 96 KB = 2 046 points per side with their records in LDS at 48 bytes per point (the crops
// have <= 1 761); round 4 kept 28-byte records in 54 KB so that a CU hosting a triangulation still fitted two
// blocks of k_match_list beside it -- with whole-record reads the build is short enough to pay for the one block
// (SVH_DT_LDS_KB overrides: A/B in profiles/r05_delaunay.txt)
static size_t dt_small_kb() {
    static const size_t kb = svh::env("SVH_DT_LDS_KB") ? (size_t)atoi(svh::env("SVH_DT_LDS_KB")) : 96;
    return std::min<size_t>(159, std::max<size_t>(kb, 8));
}
static size_t dt_lds_bytes(const svh_elas_params& p, const Dims& d, bool big) {
    const size_t hist = 4 * (size_t)(2 * (dt_columns(p, d) + d.H) + 2);
    if (big) return std::max<size_t>(hist, 159 * 1024);
    size_t want = dt_small_kb() * 1024;
    if (want > 63 * 1024 && !dt_lds_optin(false, dt_small_kb() * 1024)) want = 63 * 1024;
    return std::max<size_t>(hist, want);
}

bool stage_device_ok(const svh_elas_params& p, const Dims& d) {
    // LDS of k_delaunay: 2 (columns + rows) + 2 ints; coordinates must stay below 2^14 for the
    // 64-bit in-circle determinant, ranks are packed in 16 bits
    const int wx = dt_columns(p, d);
    return 4 * (size_t)(2 * (wx + d.H) + 2) <= 63 * 1024 && wx < (1 << 14) && d.H < (1 << 14);
}

// automatic mode takes the device stage where it is the faster one.  Lattices that k_lattice holds in
// LDS (up to ~20 k cells: KITTI-size images): always for batches.  Large lattices (1920x1080: 83 k cells
// filtered in L2, 6-12 k support points per side with 32-bit records in L2): a group takes ~2 ms through
// k_lattice + k_delaunay against ~1.2 ms on two host threads, so the host stage wins a latency-bound
// batch of 8 pairs (3.2 k vs 1.7 k pairs/s, with 7 host cores) and the device stage a deep one (64
// pairs: 4.3-4.7 k vs 3.4 k pairs/s, with 0.1 host cores); svh_elas_set_stage(1) forces it where host
// cores are scarce (8 ranks on a 16-core quota).
bool stage_device_preferred(const svh_elas_params& p, const Dims& d, bool deep_batch) {
    const size_t nc = (size_t)d.Wc * d.Hc;
    if (!stage_device_ok(p, d)) return false;
    return deep_batch || 2 * ((nc + 1) & ~(size_t)1) + 4 * ((nc + 3) / 4) <= 62 * 1024;
}

void launch_stage_device(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g, const StageDev& S,
                         GroupHdr* hdr, int32_t* support, int32_t* tri) {
    hipStream_t s = (hipStream_t)cx.stream;
    LatticeParams L;
    L.W = d.W; L.H = d.H; L.Wc = d.Wc; L.Hc = d.Hc; L.step = d.step;
    L.ws = p.incon_window_size; L.thr = p.incon_threshold; L.need = p.incon_min_support;
    L.add_corners = p.add_corners; L.sup_cap = S.sup_cap;
    const size_t nc = (size_t)d.Wc * d.Hc;
    const size_t lat_bytes = 2 * ((nc + 1) & ~(size_t)1) + 4 * ((nc + 3) / 4);   // cells + count bytes
    if (L.need > 0 && (2 * L.ws + 1) * (2 * L.ws + 1) <= 255 && L.need <= 255) {
        Timed t(cx, "k_lattice_count");
        hipLaunchKernelGGL(k_lattice_count, dim3((d.Wc + LCX - 1) / LCX, (d.Hc + LCY - 1) / LCY, g), dim3(64, 4), 0, s,
                           S, L);
    }
    {
        Timed t(cx, "k_lattice");
        if (lat_bytes <= 62 * 1024)
            hipLaunchKernelGGL(k_lattice<true>, dim3(g), dim3(512), lat_bytes, s, S, L);
        else
            hipLaunchKernelGGL(k_lattice<false>, dim3(g), dim3(512), 0, s, S, L);
    }
    DtParams D;
    D.W = dt_columns(p, d); D.H = d.H; D.sup_cap = S.sup_cap; D.rec_cap = S.rec_cap;
    D.xoff = std::max(p.disp_max, 0);
    const bool big = dt_large(d) && dt_lds_optin(true, 159 * 1024);
    static const int dt_spread = svh::env("SVH_DT_SPREAD") ? atoi(svh::env("SVH_DT_SPREAD")) : 64;
    D.spread = big ? -1 : dt_spread;      // (large sets: every depth spread)
    // scalar seam walk (see dt_build): on for a group that is the ONLY one of its call (single call, batch entry with
    // one group): the device is far from full and nobody competes for the scalar units (batch entry, 8 / 16 / 32 pairs
    // per call: 6.4 -> 6.7, 11.1 -> 11.8, 17.6 -> 18.6 k pairs/s).  Off where groups share the device -- a batch of six
    // groups on six workers lost 2 % (31.8 -> 31.1 k), deep batches 1.2 %, streams of 8-pair steps 1.8 % -- and for
    // large point sets (records in L2: 8-pair batches of 1920x1080 lost 5 % although the kernel alone got 6 %
    // faster).  SVH_DT_UNIFORM=0 / n: never / always with n nodes per wave
    static const int dt_uniform = svh::env("SVH_DT_UNIFORM") ? atoi(svh::env("SVH_DT_UNIFORM")) : -1;
    D.uniform = dt_uniform >= 0 ? dt_uniform : (cx.latency && !big ? 1 : 0);
    const size_t dt_lds = dt_lds_bytes(p, d, big);
    D.lds_ints = (int)(dt_lds / 4);
    D.lds_cap = (int)std::min<size_t>((dt_lds - 32) / 48 - 1, 8000);   // 16-bit handles: < 8191 points
    {
        Timed t(cx, "k_delaunay");
        if (big && dt_split()) {
            hipLaunchKernelGGL((k_delaunay<1024, 1>), dim3(2 * g), dim3(1024), dt_lds, s, S, D);
            DtParams B = D;
            B.lds_ints = 0;
            B.lds_cap = 0;            // records in L2: the build launch takes no dynamic LDS
            hipLaunchKernelGGL((k_delaunay<1024, 2>), dim3(2 * g), dim3(1024), 0, s, S, B);
        } else if (big) {
            hipLaunchKernelGGL((k_delaunay<1024, 0>), dim3(2 * g), dim3(1024), dt_lds, s, S, D);
        } else {
            if (dt_small_threads() == 1024) hipLaunchKernelGGL((k_delaunay<1024, 0>), dim3(2 * g), dim3(1024), dt_lds, s, S, D);
            else if (dt_small_threads() == 512) hipLaunchKernelGGL((k_delaunay<512, 0>), dim3(2 * g), dim3(512), dt_lds, s, S, D);
            else hipLaunchKernelGGL((k_delaunay<256, 0>), dim3(2 * g), dim3(256), dt_lds, s, S, D);
        }
    }
    {
        Timed t(cx, "k_stage_pack");
        hipLaunchKernelGGL(k_stage_pack, dim3(2 * g), dim3(256), 0, s, S, D, g, hdr, support, tri);
    }
}

}  // namespace svh
