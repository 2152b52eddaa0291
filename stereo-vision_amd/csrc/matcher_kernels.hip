// HIP kernels (gfx950) of the libviso2 Matcher path: feature extraction for
// Matcher::pushBack and the circular matcher / refinement of
// Matcher::matchFeatures.  Integer stencil, rank and SAD work; bit-exact with
// the reference, including the ORDER of the feature lists (match indices are
// positions in those lists):
//   * non-maximum suppression writes one slot per (block, class) in the
//     reference's scan order (u-block outer, v-block inner, classes f1min,
//     f1max, f2min, f2max) and an order-preserving compaction packs them;
//   * bin index lists are kept in ascending feature index, candidate bins are
//     walked u outer / v inner, strict "<" keeps the first minimum.
// Each kernel cites the reference lines whose result it reproduces.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "batch_rec.h"
#include "matcher_internal.h"

namespace svh {

namespace {

__device__ __forceinline__ int32_t sat_u8(int32_t x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

// ---------------------------------------------------------------------------
// M1  Matcher::createHalfResolutionImage   libviso2/src/matcher.cpp:760-776
// ---------------------------------------------------------------------------
__device__ __forceinline__ void d_half(const uint8_t* __restrict__ I, int bpl,
                                              uint8_t* __restrict__ out, int hw, int hh, int hbpl, unsigned bx, unsigned by) {
    const int x = bx * 64 + threadIdx.x;
    const int y = by * 4 + threadIdx.y;
    if (x >= hw || y >= hh) return;
    const uint8_t* r0 = I + (size_t)(2 * y) * bpl + 2 * x;
    const uint8_t* r1 = r0 + bpl;
    out[(size_t)y * hbpl + x] = (uint8_t)(((int)r0[0] + r0[1] + r1[0] + r1[1]) / 4);
}

// ---------------------------------------------------------------------------
// M2  filter::sobel5x5   libviso2/src/filter.cpp:474 (+306-361, 154-222, 93-152)
// M3  filter::blob5x5 / checkerboard5x5   filter.cpp:507-532, 492-497
// Separable form on packed 16-bit pairs (v_pk_* arithmetic; every intermediate fits 16 bits, so the
// integers are those of the 25-tap sums): a thread owns 4 adjacent pixels and walks FR rows down the
// image with the last five rows of its 8 input columns (x-2 .. x+5) in registers; per row it takes the
// five vertical sums of each column, then the horizontal combinations for its 4 pixels, and stores one
// 32-bit word per 8-bit plane (two for the 16-bit planes).  No LDS: the three aligned words a thread
// reads per row are its neighbours' words too (L1/L2 hits).
//   du = h[1 2 0 -2 -1] of v[1 4 6 4 1]      dv = h[1 4 6 4 1] of v[1 2 0 -2 -1]       (>>7, +128, clamp)
//   f1 = -box5x5 + 2 box3x3 + 7 centre       f2 = h[1 1 0 -1 -1] of v[1 1 0 -1 -1]
// Defined on rows 2..h-3, cols 2..w-3 (0 elsewhere), which covers everything the matcher ever reads.
// ---------------------------------------------------------------------------
constexpr int FX = 64, FR = 8;   // a block of 64x4 threads covers 256 columns x 32 rows
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 pk_bytes01(uint32_t w) { return __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, w, 0x0c010c00u)); }
__device__ __forceinline__ s16x2 pk_bytes23(uint32_t w) { return __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, w, 0x0c030c02u)); }
// (a.hi, b.lo): the pair one column to the right of a, given the next pair b
__device__ __forceinline__ s16x2 pk_mid(s16x2 a, s16x2 b) {
    return __builtin_bit_cast(s16x2, __builtin_amdgcn_alignbyte(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 2u));
}
// four values in two pairs -> their low bytes in one word
__device__ __forceinline__ uint32_t pk_to_bytes(s16x2 lo, s16x2 hi) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
}
__device__ __forceinline__ s16x2 pk_sobel_out(s16x2 v) {   // sat_u8((v >> 7) + 128)
    const s16x2 lo = {0, 0}, hi = {255, 255}, off = {128, 128};
    return __builtin_elementwise_min(__builtin_elementwise_max((v >> 7) + off, lo), hi);
}

template <bool kFeatures>
__device__ __forceinline__ void d_filters(const uint8_t* __restrict__ I, int w, int h, int bpl,
                                                 uint8_t* __restrict__ du, uint8_t* __restrict__ dv,
                                                 int16_t* __restrict__ f1, int16_t* __restrict__ f2, unsigned bx, unsigned by) {
    const int x = 4 * (int)(bx * FX + threadIdx.x);
    if (x >= bpl) return;
    const int y0 = (int)(by * 4 + threadIdx.y) * FR;
    if (y0 >= h) return;
    // byte mask of the columns 2..w-3 among this thread's four
    uint32_t xmask = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (x + i >= 2 && x + i < w - 2) xmask |= 0xFFu << (8 * i);
    const bool left = x >= 4, right = x + 4 < bpl;
    s16x2 win[5][4];
#pragma unroll
    for (int j = 0; j < FR + 4; j++) {
        // input row y0 - 2 + j (clamped: rows outside the image only feed outputs that are zeroed)
        const int yi = min(max(y0 - 2 + j, 0), h - 1);
        const uint32_t* row = reinterpret_cast<const uint32_t*>(I + (size_t)yi * bpl + x);
        const uint32_t w1 = row[0];
        const uint32_t w0 = left ? row[-1] : 0u, w2 = right ? row[1] : 0u;
        s16x2* r = win[j % 5];
        r[0] = pk_bytes23(w0);   // columns x-2, x-1
        r[1] = pk_bytes01(w1);   //         x,   x+1
        r[2] = pk_bytes23(w1);   //         x+2, x+3
        r[3] = pk_bytes01(w2);   //         x+4, x+5
        if (j < 4) continue;
        const int y = y0 + j - 4;
        if (y >= h) break;
        const s16x2 *p0 = win[(j + 1) % 5], *p1 = win[(j + 2) % 5], *p2 = win[(j + 3) % 5], *p3 = win[(j + 4) % 5], *p4 = win[j % 5];
        s16x2 S[4], T[4], V5[4], V3[4], VK[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const s16x2 a = p0[c] + p4[c], b = p1[c] + p3[c], d04 = p0[c] - p4[c], d13 = p1[c] - p3[c];
            const s16x2 four = {4, 4}, six = {6, 6}, two = {2, 2};
            S[c] = a + four * b + six * p2[c];       // vertical 1 4 6 4 1
            T[c] = d04 + two * d13;                  // vertical 1 2 0 -2 -1
            if (kFeatures) {
                V5[c] = a + b + p2[c];               // vertical 1 1 1 1 1
                V3[c] = b + p2[c];                   // vertical 0 1 1 1 0
                VK[c] = d04 + d13;                   // vertical 1 1 0 -1 -1
            }
        }
        const bool rowok = y >= 2 && y < h - 2;
        const uint32_t keep = rowok ? xmask : 0u;
        const s16x2 two = {2, 2}, four = {4, 4}, six = {6, 6};
        // outputs (x, x+1) sit on column pairs 1; (x+2, x+3) on pair 2.  m01 = columns (x-1, x), m12 = (x+1, x+2), ...
        const s16x2 Sm01 = pk_mid(S[0], S[1]), Sm12 = pk_mid(S[1], S[2]), Sm23 = pk_mid(S[2], S[3]);
        const s16x2 du_a = (S[0] - S[2]) + two * (Sm01 - Sm12);   // S[c-2] + 2 S[c-1] - 2 S[c+1] - S[c+2]
        const s16x2 du_b = (S[1] - S[3]) + two * (Sm12 - Sm23);
        const s16x2 Tm01 = pk_mid(T[0], T[1]), Tm12 = pk_mid(T[1], T[2]), Tm23 = pk_mid(T[2], T[3]);
        const s16x2 dv_a = (T[0] + T[2]) + four * (Tm01 + Tm12) + six * T[1];
        const s16x2 dv_b = (T[1] + T[3]) + four * (Tm12 + Tm23) + six * T[2];
        const size_t o = (size_t)y * bpl + x;
        *reinterpret_cast<uint32_t*>(du + o) = pk_to_bytes(pk_sobel_out(du_a), pk_sobel_out(du_b)) & keep;
        *reinterpret_cast<uint32_t*>(dv + o) = pk_to_bytes(pk_sobel_out(dv_a), pk_sobel_out(dv_b)) & keep;
        if (kFeatures) {
            const s16x2 Am01 = pk_mid(V5[0], V5[1]), Am12 = pk_mid(V5[1], V5[2]), Am23 = pk_mid(V5[2], V5[3]);
            const s16x2 Bm01 = pk_mid(V3[0], V3[1]), Bm12 = pk_mid(V3[1], V3[2]), Bm23 = pk_mid(V3[2], V3[3]);
            const s16x2 Km01 = pk_mid(VK[0], VK[1]), Km12 = pk_mid(VK[1], VK[2]), Km23 = pk_mid(VK[2], VK[3]);
            const s16x2 seven = {7, 7};
            const s16x2 box5_a = V5[0] + Am01 + V5[1] + Am12 + V5[2], box5_b = V5[1] + Am12 + V5[2] + Am23 + V5[3];
            const s16x2 box3_a = Bm01 + V3[1] + Bm12, box3_b = Bm12 + V3[2] + Bm23;
            const s16x2 f1_a = two * box3_a - box5_a + seven * p2[1], f1_b = two * box3_b - box5_b + seven * p2[2];
            const s16x2 f2_a = (VK[0] - VK[2]) + (Km01 - Km12), f2_b = (VK[1] - VK[3]) + (Km12 - Km23);
            // 16-bit planes: mask = the byte mask widened to halves
            const uint32_t k01 = ((keep & 0xFFu) ? 0xFFFFu : 0u) | ((keep & 0xFF00u) ? 0xFFFF0000u : 0u);
            const uint32_t k23 = ((keep & 0xFF0000u) ? 0xFFFFu : 0u) | ((keep & 0xFF000000u) ? 0xFFFF0000u : 0u);
            uint2 o1, o2;
            o1.x = __builtin_bit_cast(uint32_t, f1_a) & k01;
            o1.y = __builtin_bit_cast(uint32_t, f1_b) & k23;
            o2.x = __builtin_bit_cast(uint32_t, f2_a) & k01;
            o2.y = __builtin_bit_cast(uint32_t, f2_b) & k23;
            *reinterpret_cast<uint2*>(f1 + o) = o1;
            *reinterpret_cast<uint2*>(f2 + o) = o2;
        }
    }
}

// ---------------------------------------------------------------------------
// M4  Matcher::nonMaximumSuppression   matcher.cpp:395-530
// kG lanes per (n+1)x(n+1) block (16 for the dense n = 3 pass: one lane per
// pixel of the block; 64 for wider blocks).  The reference's scan keeps the
// FIRST smallest / largest value in (i outer, j inner) order; that is the lane
// minimum of the key  value<<16 | scan_index  (maximum of value<<16 | ~index).
// The neighbourhood test is an order-free "any pixel beats it" vote.
// Slot = (iblock*nj + jblock)*4 + class.
// ---------------------------------------------------------------------------
template <int kG>
__device__ __forceinline__ int group_min(int v) {
#pragma unroll
    for (int m = kG / 2; m >= 1; m >>= 1) {
        const int o = __shfl_xor(v, m, kG);
        v = o < v ? o : v;
    }
    return v;
}
template <int kG>
__device__ __forceinline__ int group_max(int v) {
#pragma unroll
    for (int m = kG / 2; m >= 1; m >>= 1) {
        const int o = __shfl_xor(v, m, kG);
        v = o > v ? o : v;
    }
    return v;
}
template <int kG>
__device__ __forceinline__ bool group_any(bool p) {
    const unsigned long long b = __ballot(p);
    if (kG == 64) return b != 0;
    const int sh = (int)(threadIdx.x & 63) / kG * kG;
    return ((b >> sh) & ((1ull << kG) - 1ull)) != 0;
}

template <int kG>
__device__ __forceinline__ void d_nms(const int16_t* __restrict__ f1,
                                             const int16_t* __restrict__ f2, int w, int h, int bpl,
                                             int n, int tau, int margin, int ni, int nj,
                                             int4* __restrict__ slots, int32_t* __restrict__ flags, unsigned bx) {
    const int b = (int)(bx * 256 + threadIdx.x) / kG;
    const int lane = (int)threadIdx.x % kG;
    if (b >= ni * nj) return;   // whole groups leave together
    const int ib = b / nj, jb = b - ib * nj;
    const int i = n + margin + ib * (n + 1), j = n + margin + jb * (n + 1);
    const int n1 = n + 1, np = n1 * n1;
    const int16_t* F[2] = {f1, f2};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        int kmin = 0x7FFFFFFF, kmax = (int)0x80000000;
        for (int l = lane; l < np; l += kG) {
            const int di = l / n1, dj = l - di * n1;
            const int c = F[k][(size_t)(j + dj) * bpl + i + di];
            const int lo = c * 65536 + l, hi = c * 65536 + (0xFFFF - l);
            kmin = lo < kmin ? lo : kmin;
            kmax = hi > kmax ? hi : kmax;
        }
        kmin = group_min<kG>(kmin);
        kmax = group_max<kG>(kmax);
#pragma unroll
        for (int mm = 0; mm < 2; mm++) {
            const bool is_min = mm == 0;
            const int key = is_min ? kmin : kmax;
            const int cv = key >> 16;
            const int l = is_min ? (key & 0xFFFF) : 0xFFFF - (key & 0xFFFF);
            const int ci = i + l / n1, cj = j + l % n1;
            const int i0 = ci - n, j0 = cj - n;
            const int ie = min(ci + n, w - 1 - margin), je = min(cj + n, h - 1 - margin);
            const int nwj = je - j0 + 1, total = (ie - i0 + 1) * nwj;
            bool bad = false;
            for (int q = lane; q < total; q += kG) {
                const int di = q / nwj, i2 = i0 + di, j2 = j0 + (q - di * nwj);
                const int c = F[k][(size_t)j2 * bpl + i2];
                const bool beats = is_min ? c < cv : c > cv;
                bad = bad || (beats && (i2 < i || i2 > i + n || j2 < j || j2 > j + n));
            }
            const bool ok = !group_any<kG>(bad) && (is_min ? cv <= -tau : cv >= tau);
            if (lane == 0) {
                const int slot = b * 4 + 2 * k + mm;
                flags[slot] = ok ? 1 : 0;
                if (ok) slots[slot] = make_int4(ci, cj, cv, 2 * k + mm);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Order-preserving compaction by one workgroup: thread t owns a contiguous
// chunk of slots, an LDS scan of the per-thread counts gives its output base.
// ---------------------------------------------------------------------------
// (round 6: a scan inside each wave by lane shuffles, then the 16 wave totals through LDS -- two barriers instead of
// the twenty of a 1024-wide Hillis-Steele scan, which were most of the 11-14 us these single-workgroup kernels took)
__device__ __forceinline__ int block_exclusive_scan_1024(int value, int* total) {
    __shared__ int s_wave[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int incl = value;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const int x = s_wave[w];
        before += w < wave ? x : 0;
        all += x;
    }
    __syncthreads();   // (s_wave may be written again by the caller's next scan)
    *total = all;
    return before + incl - value;
}

// ordered compaction of the surviving slots: order[k] = slot of the k-th feature
__device__ __forceinline__ void d_compact_slots(const int32_t* __restrict__ flags, int nslots,
                                                        int32_t* __restrict__ order,
                                                        int32_t* __restrict__ count, unsigned bx,
                                                        int32_t* __restrict__ count_host = nullptr) {
    const int t = threadIdx.x;
    const int chunk = ((nslots + 1023) / 1024 + 3) & ~3;   // multiple of 4: 16-byte flag loads
    const int lo = min(t * chunk, nslots), hi = min(lo + chunk, nslots);
    int mine = 0;
    // the flags of a chunk of up to 64 slots are kept as a bit mask: the write pass then reads nothing (it used to load
    // every flag again, one dependent load per slot and turn: 20 of the kernel's 24-35 us on a dense table)
    const bool masked = chunk <= 64;
    unsigned long long bits = 0;
    if (masked) {
#pragma unroll 4
        for (int q = 0; q < 16; q++) {
            const int s = lo + 4 * q;
            if (s >= hi) break;
            if (s + 4 <= hi) {
                const int4 f = *reinterpret_cast<const int4*>(flags + s);
                const unsigned nib = (f.x ? 1u : 0u) | (f.y ? 2u : 0u) | (f.z ? 4u : 0u) | (f.w ? 8u : 0u);
                bits |= (unsigned long long)nib << (4 * q);
            } else {
                for (int r = s; r < hi; r++) bits |= (unsigned long long)(flags[r] ? 1 : 0) << (r - lo);
            }
        }
        mine = __popcll(bits);
    } else {
        for (int s = lo; s < hi; s += 4) {
            if (s + 4 <= hi) {
                const int4 f = *reinterpret_cast<const int4*>(flags + s);
                mine += f.x + f.y + f.z + f.w;
            } else {
                for (int r = s; r < hi; r++) mine += flags[r];
            }
        }
    }
    int total;
    int base = block_exclusive_scan_1024(mine, &total);
    if (masked) {
        for (; bits; bits &= bits - 1) order[base++] = lo + __builtin_ctzll(bits);
    } else if (mine) {
        for (int s = lo; s < hi; s++)
            if (flags[s]) order[base++] = s;
    }
    if (t == 0) {
        *count = total;
        if (count_host) *count_host = total;   // (pinned host memory: the count needs no copy launch of its own)
    }
}

// M5  descriptor + record packing   matcher.cpp:534-579, 854-877
// one thread per (feature, descriptor word): 16 (du,dv) pairs around (u, v-1),
// rows -5,-3,-1,+1,+3,+5 relative to v
__device__ __forceinline__ void d_feature_records(const int4* __restrict__ slots,
                                                         const int32_t* __restrict__ order,
                                                         const int32_t* __restrict__ count,
                                                         const uint8_t* __restrict__ du,
                                                         const uint8_t* __restrict__ dv, int bpl,
                                                         int scale, int32_t* __restrict__ table, unsigned bx) {
    const int g = bx * 256 + threadIdx.x;
    const int feat = g >> 3, q = g & 7;
    if (feat >= *count) return;
    const int4 m = slots[order[feat]];
    int32_t* rec = table + (size_t)12 * feat;
    if (q < 4) rec[q] = q == 0 ? m.x * scale : q == 1 ? m.y * scale : q == 2 ? 0 : m.w;
    // word q packs (du,dv) at (row -r, dx) and (row +r, dx), rows relative to v
    const int dx[8] = {-3, -1, 3, 1, -1, 1, -5, 5};
    const int rr[8] = {1, 1, 1, 1, 5, 5, 3, 3};
    const ptrdiff_t base = (ptrdiff_t)m.y * bpl + m.x + dx[q];
    const ptrdiff_t a0 = base - (ptrdiff_t)rr[q] * bpl, b0 = base + (ptrdiff_t)rr[q] * bpl;
    const uint32_t w0 = du[a0], w1 = dv[a0], w2 = du[b0], w3 = dv[b0];
    rec[4 + q] = (int32_t)(w0 | (w1 << 8) | (w2 << 16) | (w3 << 24));
}

// ---------------------------------------------------------------------------
// M6  Matcher::createIndexVector   matcher.cpp:1036-1057
// CSR over class x v_bin x u_bin; every list ends up in ascending feature index
// (the reference's push_back order).  One workgroup per table.
// ---------------------------------------------------------------------------
// LDS build: histogram, scan, scatter and the per-bin ascending sort all stay on chip;
// used when 2*nb + 1 + n ints fit the LDS budget (launcher), else k_bin_index below.
// (table / count / off / ids: the four pointers of the workgroup's table.  The callers pick them out of their BinJobs
// -- a kernel argument in the single-object form, a row of the job table in the batched form -- so that the 32
// pointers are never copied: round 4's batched form indexed a by-value copy and spilled 264 bytes of scratch)
__device__ __forceinline__ void d_bin_index_lds(const int32_t* __restrict__ table, const int32_t* __restrict__ count,
                                                int32_t* __restrict__ off, int32_t* __restrict__ ids, int ub, int vb,
                                                int binsize) {
    extern __shared__ int s_bin[];
    // one workgroup per table: the four tables of a stereo frame build concurrently
    const int n = *count, nb = 4 * ub * vb, t = threadIdx.x;
    int* s_off = s_bin;            // nb + 1
    int* s_cur = s_bin + nb + 1;   // nb
    int* s_ids = s_cur + nb;       // n
    for (int b = t; b <= nb; b += 1024) s_off[b] = 0;
    for (int b = t; b < nb; b += 1024) s_cur[b] = 0;
    __syncthreads();
    auto bin_of = [&](int i) {
        const int32_t* r = table + (size_t)12 * i;
        int u_bin = (int)floorf(__fdiv_rn((float)r[0], (float)binsize));
        int v_bin = (int)floorf(__fdiv_rn((float)r[1], (float)binsize));
        u_bin = u_bin < ub - 1 ? u_bin : ub - 1;
        v_bin = v_bin < vb - 1 ? v_bin : vb - 1;
        return (r[3] * vb + v_bin) * ub + u_bin;
    };
    for (int i = t; i < n; i += 1024) atomicAdd(&s_off[bin_of(i) + 1], 1);
    __syncthreads();
    // inclusive scan of s_off[1..nb]: thread t owns a contiguous chunk
    {
        const int chunk = (nb + 1023) / 1024;
        const int lo = min(1 + t * chunk, nb + 1), hi = min(lo + chunk, nb + 1);
        int mine = 0;
        for (int b = lo; b < hi; b++) mine += s_off[b];
        int total;
        int run = block_exclusive_scan_1024(mine, &total);
        for (int b = lo; b < hi; b++) {
            run += s_off[b];
            s_off[b] = run;
        }
    }
    __syncthreads();
    // the atomics hand out the places of a bin in arrival order; the reference's lists are in index order
    // (matcher.cpp:1036-1057).  Usual sizes: an entry carries its bin in the high bits, and every ENTRY then counts the
    // smaller indices of its bin and goes straight to its place in the output -- all entries at once, where one thread
    // per bin used to insertion-sort its list (10 of the kernel's 20 us on a dense table).
    const bool ranked = nb < 2048 && n < (1 << 20);
    for (int i = t; i < n; i += 1024) {
        const int b = bin_of(i);
        s_ids[s_off[b] + atomicAdd(&s_cur[b], 1)] = ranked ? i | (b << 20) : i;
    }
    __syncthreads();
    if (ranked) {
        for (int p = t; p < n; p += 1024) {
            const int e = s_ids[p], b = e >> 20;
            const int lo = s_off[b], hi = s_off[b + 1];
            int rank = 0;
            for (int q = lo; q < hi; q++) rank += s_ids[q] < e;
            ids[lo + rank] = e & ((1 << 20) - 1);
        }
        for (int b = t; b <= nb; b += 1024) off[b] = s_off[b];
        return;
    }
    for (int b = t; b < nb; b += 1024) {   // short lists: insertion sort to ascending index
        const int lo = s_off[b], hi = s_off[b + 1];
        for (int a = lo + 1; a < hi; a++) {
            const int key = s_ids[a];
            int q = a - 1;
            while (q >= lo && s_ids[q] > key) {
                s_ids[q + 1] = s_ids[q];
                q--;
            }
            s_ids[q + 1] = key;
        }
    }
    __syncthreads();
    for (int b = t; b <= nb; b += 1024) off[b] = s_off[b];
    for (int i = t; i < n; i += 1024) ids[i] = s_ids[i];
}

__global__ __launch_bounds__(1024) void k_bin_index(const int32_t* __restrict__ table,
                                                    const int32_t* __restrict__ count, int ub, int vb,
                                                    int binsize, int32_t* __restrict__ off,
                                                    int32_t* __restrict__ ids,
                                                    int32_t* __restrict__ cursor) {
    const int n = *count, nb = 4 * ub * vb, t = threadIdx.x;
    for (int b = t; b <= nb; b += 1024) off[b] = 0;
    for (int b = t; b < nb; b += 1024) cursor[b] = 0;
    __syncthreads();
    auto bin_of = [&](int i) {
        const int32_t* r = table + (size_t)12 * i;
        int u_bin = (int)floorf(__fdiv_rn((float)r[0], (float)binsize));
        int v_bin = (int)floorf(__fdiv_rn((float)r[1], (float)binsize));
        u_bin = u_bin < ub - 1 ? u_bin : ub - 1;
        v_bin = v_bin < vb - 1 ? v_bin : vb - 1;
        return (r[3] * vb + v_bin) * ub + u_bin;
    };
    for (int i = t; i < n; i += 1024) atomicAdd(&off[bin_of(i) + 1], 1);
    __syncthreads();
    if (t == 0)
        for (int b = 0; b < nb; b++) off[b + 1] += off[b];
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const int b = bin_of(i);
        ids[off[b] + atomicAdd(&cursor[b], 1)] = i;
    }
    __syncthreads();
    for (int b = t; b < nb; b += 1024) {   // short lists: insertion sort to ascending index
        const int lo = off[b], hi = off[b + 1];
        for (int a = lo + 1; a < hi; a++) {
            const int key = ids[a];
            int q = a - 1;
            while (q >= lo && ids[q] > key) {
                ids[q + 1] = ids[q];
                q--;
            }
            ids[q + 1] = key;
        }
    }
}

// ---------------------------------------------------------------------------
// M7  Matcher::findMatch   matcher.cpp:1061-1157
// Sequential walk in the reference's order: u_bin outer, v_bin inner, list order,
// strict "<" on a double cost.  fp64 / fp32 expressions use the non-contracted
// IEEE intrinsics so windows and predicted-position costs round identically.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sad32(const int32_t* a, const int32_t* b) {
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s = __builtin_amdgcn_sad_u8((uint32_t)a[k], (uint32_t)b[k], s);
    return s;
}

// kQ lanes work on one query: bins are walked in the reference's order by the whole
// group, the lanes stride over a bin's list.  Every lane keeps its first minimum
// (strict "<" on its ascending subsequence); the group winner is the smallest
// (cost, traversal position) pair, i.e. the first minimum of the sequential walk.
constexpr int kQ = 16;

__device__ int find_match(const MatchParams& P, const FeatView& t1, int i1, const FeatView& t2,
                          const float* __restrict__ ranges, int stat_bin, int stage, bool flow,
                          bool use_prior, double u_, double v_, int lane) {
    int min_ind = 0, min_seq = 0x7FFFFFFF;
    double min_cost = 10000000;
    const int32_t* r1 = t1.rec + (size_t)12 * i1;
    const int u1 = r1[0], v1 = r1[1], c = r1[3];
    int32_t d1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) d1[k] = r1[4 + k];
    float u_min, u_max, v_min, v_max;
    if (use_prior) {
        const float* rg = ranges + (size_t)16 * stat_bin;   // u_min[4],u_max[4],v_min[4],v_max[4]
        u_min = __fadd_rn((float)u1, rg[0 + stage]);
        u_max = __fadd_rn((float)u1, rg[4 + stage]);
        v_min = __fadd_rn((float)v1, rg[8 + stage]);
        v_max = __fadd_rn((float)v1, rg[12 + stage]);
    } else {
        u_min = (float)(u1 - P.match_radius);
        u_max = (float)(u1 + P.match_radius);
        v_min = (float)(v1 - P.match_radius);
        v_max = (float)(v1 + P.match_radius);
    }
    if (!flow) {
        v_min = (float)(v1 - P.match_disp_tolerance);
        v_max = (float)(v1 + P.match_disp_tolerance);
    }
    const float bs = (float)P.binsize;
    auto bin = [&](float x, int nb) {
        int b = (int)floorf(__fdiv_rn(x, bs));
        b = b > 0 ? b : 0;
        return b < nb - 1 ? b : nb - 1;
    };
    const int ub0 = bin(u_min, P.ub), ub1 = bin(u_max, P.ub);
    const int vb0 = bin(v_min, P.vb), vb1 = bin(v_max, P.vb);
    const bool predicted = u_ >= 0 && v_ >= 0;
    int seq0 = 0;   // traversal position of the first entry of the current chunk of bins
    // Round 6: the walk used to take the bins one after the other, the kQ lanes sharing the entries of one bin, and
    // waited for a bin's offsets, then an id, then the candidate's position, then its descriptor -- four dependent
    // global loads per bin, 25 bins in a flow stage without a prior, mostly with one or two entries each.  Now every
    // lane fetches the offsets of ONE bin of a chunk of kQ, a scan over the lanes lays the entries of the chunk out in
    // the reference's traversal order (u_bin outer, v_bin inner, list order), and the lanes share that flat list:
    // entry p belongs to the bin whose range of positions holds p.  A candidate's id and its whole 48-byte record are
    // two round trips.  The traversal position p decides ties exactly as before.
    const int nv = vb1 - vb0 + 1, nbins = (ub1 - ub0 + 1) * nv;
    // (the dense pass -- windows from the prior statistics, a few bins of half a dozen entries each, 1 350 waves that fill
    // the device -- keeps the lanes on the entries of one bin at a time, the offsets of four bins requested together:
    // the flat list measured slower there, 0.065 -> 0.074 ms, and faster on the sparse pass, 0.048 -> 0.028 ms)
    if (use_prior) {
        for (int k0 = 0; k0 < nbins; k0 += 4) {
            int lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = k0 + j;
                lo[j] = hi[j] = 0;
                if (k < nbins) {
                    const int du = k / nv;
                    const int b = (c * P.vb + vb0 + (k - du * nv)) * P.ub + ub0 + du;
                    lo[j] = t2.off[b];
                    hi[j] = t2.off[b + 1];
                }
            }
            // a lane's first entry of each of the four bins: the four ids together, then the four records together --
            // three round trips for a batch whose bins hold at most kQ entries each (nearly all), where the bins used
            // to take their two trips one after the other.  Evaluated in bin order, so ties fall as before.
            int q4[4], i4[4];
            int4 ra4[4], rb4[4], rc4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                q4[j] = lo[j] + lane;
                i4[j] = q4[j] < hi[j] ? t2.ids[q4[j]] : -1;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (i4[j] >= 0) {
                    const int4* r4 = reinterpret_cast<const int4*>(t2.rec + (size_t)12 * i4[j]);
                    ra4[j] = r4[0]; rb4[j] = r4[1]; rc4[j] = r4[2];
                }
            }
            auto consider = [&](int i2, const int4& ra, const int4& rb, const int4& rc, int seq) {
                const float u2 = (float)ra.x, v2 = (float)ra.y;
                if (u2 >= u_min && u2 <= u_max && v2 >= v_min && v2 <= v_max) {
                    const int32_t d2[8] = {rb.x, rb.y, rb.z, rb.w, rc.x, rc.y, rc.z, rc.w};
                    double cost = (double)sad32(d1, d2);
                    if (predicted) {
                        const double du = __dsub_rn((double)ra.x, u_), dv = __dsub_rn((double)ra.y, v_);
                        const double dist = __dsqrt_rn(__dadd_rn(__dmul_rn(du, du), __dmul_rn(dv, dv)));
                        cost = __dadd_rn(cost, __dmul_rn(4.0, dist));
                    }
                    if (cost < min_cost) {
                        min_ind = i2;
                        min_cost = cost;
                        min_seq = seq;
                    }
                }
            };
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (i4[j] >= 0) consider(i4[j], ra4[j], rb4[j], rc4[j], seq0 + (q4[j] - lo[j]));
                for (int q = q4[j] + kQ; q < hi[j]; q += kQ) {   // (a bin with more than kQ entries)
                    const int i2 = t2.ids[q];
                    const int4* r4 = reinterpret_cast<const int4*>(t2.rec + (size_t)12 * i2);
                    const int4 ra = r4[0], rb = r4[1], rc = r4[2];
                    consider(i2, ra, rb, rc, seq0 + (q - lo[j]));
                }
                seq0 += hi[j] - lo[j];
            }
        }
    } else
    for (int k0 = 0; k0 < nbins; k0 += kQ) {
        const int k = k0 + lane;
        int lo = 0, cnt = 0;
        if (k < nbins) {
            const int du = k / nv;
            const int b = (c * P.vb + vb0 + (k - du * nv)) * P.ub + ub0 + du;
            lo = t2.off[b];
            cnt = t2.off[b + 1] - lo;
        }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < kQ; o <<= 1) {
            const int up = __shfl_up(incl, o, kQ);
            if (lane >= o) incl += up;
        }
        const int start = incl - cnt;
        const int total = __shfl(incl, kQ - 1, kQ);
        for (int p0 = 0; p0 < total; p0 += kQ) {
            const int p = p0 + lane;
            int q = -1;
            const int nj = min(kQ, nbins - k0);   // (a window with a prior has one to four bins: as many turns)
            for (int j = 0; j < nj; j++) {
                const int sj = __shfl(start, j, kQ), cj = __shfl(cnt, j, kQ), lj = __shfl(lo, j, kQ);
                if (p >= sj && p < sj + cj) q = lj + (p - sj);
            }
            if (q >= 0) {
                const int i2 = t2.ids[q];
                const int4* r4 = reinterpret_cast<const int4*>(t2.rec + (size_t)12 * i2);
                const int4 ra = r4[0], rb = r4[1], rc = r4[2];
                const float u2 = (float)ra.x, v2 = (float)ra.y;
                if (u2 >= u_min && u2 <= u_max && v2 >= v_min && v2 <= v_max) {
                    const int32_t d2[8] = {rb.x, rb.y, rb.z, rb.w, rc.x, rc.y, rc.z, rc.w};
                    double cost = (double)sad32(d1, d2);
                    if (predicted) {
                        const double du = __dsub_rn((double)ra.x, u_), dv = __dsub_rn((double)ra.y, v_);
                        const double dist = __dsqrt_rn(__dadd_rn(__dmul_rn(du, du), __dmul_rn(dv, dv)));
                        cost = __dadd_rn(cost, __dmul_rn(4.0, dist));
                    }
                    if (cost < min_cost) {
                        min_ind = i2;
                        min_cost = cost;
                        min_seq = seq0 + p;
                    }
                }
            }
        }
        seq0 += total;
    }
    // group winner: smallest cost, ties to the earlier traversal position
#pragma unroll
    for (int m = kQ / 2; m >= 1; m >>= 1) {
        const double oc = __shfl_xor(min_cost, m, kQ);
        const int os = __shfl_xor(min_seq, m, kQ);
        const int oi = __shfl_xor(min_ind, m, kQ);
        if (oc < min_cost || (oc == min_cost && os < min_seq)) {
            min_cost = oc;
            min_seq = os;
            min_ind = oi;
        }
    }
    return min_ind;
}

__device__ __forceinline__ int stat_bin_of(const MatchParams& P, int u, int v) {
    int u_bin = (int)floorf(__fdiv_rn((float)u, (float)P.binsize));
    int v_bin = (int)floorf(__fdiv_rn((float)v, (float)P.binsize));
    u_bin = u_bin < P.ub - 1 ? u_bin : P.ub - 1;
    v_bin = v_bin < P.vb - 1 ? v_bin : P.vb - 1;
    return v_bin * P.ub + u_bin;
}

__device__ __forceinline__ svh_p_match mk(float u1p, float v1p, int i1p, float u2p, float v2p, int i2p,
                                          float u1c, float v1c, int i1c, float u2c, float v2c, int i2c) {
    svh_p_match m;
    m.u1p = u1p; m.v1p = v1p; m.i1p = i1p; m.u2p = u2p; m.v2p = v2p; m.i2p = i2p;
    m.u1c = u1c; m.v1c = v1c; m.i1c = i1c; m.u2c = u2c; m.v2c = v2c; m.i2c = i2c;
    return m;
}

// M8  Matcher::matching   matcher.cpp:1161-1379 -- one thread per query feature.
// flags: 0 = no match, 1 = match.  For flow/stereo the "pixel not matched yet"
// rule (first query in index order wins) is resolved by k_match_dedupe.
__device__ __forceinline__ void d_match(MatchParams P, FeatView m1p, FeatView m2p, FeatView m1c,
                                               FeatView m2c, const float* __restrict__ ranges,
                                               int use_prior, svh_p_match* __restrict__ out,
                                               int32_t* __restrict__ flags,
                                               int32_t* __restrict__ pixel_owner, unsigned bx) {
    const int i = (int)(bx * 128 + threadIdx.x) / kQ;
    const int lane = (int)threadIdx.x % kQ;
    const FeatView& q = P.method == 2 ? m1p : m1c;
    if (i >= *q.count) return;   // whole groups leave together
    int ok = 0;
    svh_p_match m = mk(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const int32_t* r = q.rec + (size_t)12 * i;
    const int uq = r[0], vq = r[1];
    const int sb = stat_bin_of(P, uq, vq);
    if (P.method == 0) {
        const int i1p = find_match(P, m1c, i, m1p, ranges, sb, 0, true, use_prior, -1, -1, lane);
        const int i1c2 = find_match(P, m1p, i1p, m1c, ranges, sb, 1, true, use_prior, -1, -1, lane);
        if (i1c2 == i) {
            const int32_t* rp = m1p.rec + (size_t)12 * i1p;
            m = mk((float)rp[0], (float)rp[1], i1p, -1, -1, -1, (float)uq, (float)vq, i, -1, -1, -1);
            ok = 1;
        }
    } else if (P.method == 1) {
        const int i2c = find_match(P, m1c, i, m2c, ranges, sb, 0, false, use_prior, -1, -1, lane);
        const int i1c2 = find_match(P, m2c, i2c, m1c, ranges, sb, 1, false, use_prior, -1, -1, lane);
        if (i1c2 == i) {
            const int32_t* r2 = m2c.rec + (size_t)12 * i2c;
            if (uq >= r2[0]) {
                m = mk(-1, -1, -1, -1, -1, -1, (float)uq, (float)vq, i, (float)r2[0], (float)r2[1], i2c);
                ok = 1;
            }
        }
    } else {
        const int i2p = find_match(P, m1p, i, m2p, ranges, sb, 0, false, use_prior, -1, -1, lane);
        const int32_t* r2p = m2p.rec + (size_t)12 * i2p;
        const int u2p = r2p[0], v2p = r2p[1];
        double pu = -1, pv = -1, bu = -1, bv = -1;
        if (P.has_tr) {
            // predicted position in the current right image (matcher.cpp:1312-1327)
            double d = __dsub_rn((double)uq, (double)u2p);
            d = d > 1.0 ? d : 1.0;
            const double x1p = __ddiv_rn(__dmul_rn(__dsub_rn((double)uq, P.cu), P.base), d);
            const double y1p = __ddiv_rn(__dmul_rn(__dsub_rn((double)vq, P.cv), P.base), d);
            const double z1p = __ddiv_rn(__dmul_rn(P.f, P.base), d);
            const double* T = P.tr;
            const double x2c = __dsub_rn(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T[0], x1p), __dmul_rn(T[1], y1p)),
                                                             __dmul_rn(T[2], z1p)), T[3]), P.base);
            const double y2c = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T[4], x1p), __dmul_rn(T[5], y1p)),
                                                   __dmul_rn(T[6], z1p)), T[7]);
            const double z2c = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T[8], x1p), __dmul_rn(T[9], y1p)),
                                                   __dmul_rn(T[10], z1p)), T[11]);
            pu = __dadd_rn(__ddiv_rn(__dmul_rn(P.f, x2c), z2c), P.cu);
            pv = __dadd_rn(__ddiv_rn(__dmul_rn(P.f, y2c), z2c), P.cv);
            bu = (double)uq;
            bv = (double)vq;
        }
        const int i2c = find_match(P, m2p, i2p, m2c, ranges, sb, 1, true, use_prior, pu, pv, lane);
        const int i1c = find_match(P, m2c, i2c, m1c, ranges, sb, 2, false, use_prior, -1, -1, lane);
        const int i1p2 = find_match(P, m1c, i1c, m1p, ranges, sb, 3, true, use_prior, bu, bv, lane);
        if (i1p2 == i) {
            const int32_t* r2c = m2c.rec + (size_t)12 * i2c;
            const int32_t* r1c = m1c.rec + (size_t)12 * i1c;
            if (uq >= u2p && r1c[0] >= r2c[0]) {
                m = mk((float)uq, (float)vq, i, (float)u2p, (float)v2p, i2p, (float)r1c[0], (float)r1c[1], i1c,
                       (float)r2c[0], (float)r2c[1], i2c);
                ok = 1;
            }
        }
    }
    if (lane != 0) return;
    if (ok && P.method < 2) atomicMin(&pixel_owner[(size_t)vq * P.width + uq], i);
    flags[i] = ok;
    if (ok) out[i] = m;
}

// flow / stereo: keep a match only if its query is the first one on its pixel
__device__ __forceinline__ void d_match_dedupe(const int32_t* __restrict__ n, int width,
                                                      const svh_p_match* __restrict__ m,
                                                      int32_t* __restrict__ flags,
                                                      const int32_t* __restrict__ pixel_owner, unsigned bx) {
    const int i = bx * 256 + threadIdx.x;
    if (i >= *n || !flags[i]) return;
    const int u = (int)m[i].u1c, v = (int)m[i].v1c;
    if (pixel_owner[(size_t)v * width + u] != i) flags[i] = 0;
}

constexpr int kCompactLds = 8192;   // survivors whose slot numbers fit the LDS list of d_compact_matches
__device__ __forceinline__ void d_compact_matches(const svh_p_match* __restrict__ in,
                                                          const int32_t* __restrict__ flags,
                                                          const int32_t* __restrict__ nslots_ptr,
                                                          svh_p_match* __restrict__ out,
                                                          int32_t* __restrict__ count, unsigned bx,
                                                          int32_t* __restrict__ count_host = nullptr) {
    const int nslots = *nslots_ptr;
    const int t = threadIdx.x;
    const int chunk = (nslots + 1023) / 1024;
    const int lo = min(t * chunk, nslots), hi = min(lo + chunk, nslots);
    int mine = 0;
    const bool masked = chunk <= 32;   // (as in d_compact_slots: the flags of the chunk as a bit mask)
    unsigned bits = 0;
    if (masked) {
#pragma unroll 8
        for (int q = 0; q < 32; q++)
            if (lo + q < hi) bits |= (flags[lo + q] ? 1u : 0u) << q;
        mine = __popc(bits);
    } else {
        for (int s = lo; s < hi; s++) mine += flags[s];
    }
    int total;
    int base = block_exclusive_scan_1024(mine, &total);
    // the survivors' slot numbers go through LDS, then the whole workgroup moves the 48-byte records as 16-byte pieces
    // side by side (a thread copying its own records one after the other was 6 of the kernel's 10 us)
    __shared__ int s_src[kCompactLds];
    static_assert(sizeof(svh_p_match) == 48, "three 16-byte pieces per record");
    if (masked && total <= kCompactLds) {
        for (; bits; bits &= bits - 1) s_src[base++] = lo + __builtin_ctz(bits);
        __syncthreads();
        const uint4* src = reinterpret_cast<const uint4*>(in);
        uint4* dst = reinterpret_cast<uint4*>(out);
        for (int k = t; k < 3 * total; k += 1024) {
            const int m = k / 3, part = k - 3 * m;
            dst[k] = src[3 * s_src[m] + part];
        }
    } else if (masked) {
        for (; bits; bits &= bits - 1) out[base++] = in[lo + __builtin_ctz(bits)];
    } else {
        for (int s = lo; s < hi; s++)
            if (flags[s]) out[base++] = in[s];
    }
    if (t == 0) {
        *count = total;
        if (count_host) *count_host = total;   // (pinned host memory: no copy launch for the count)
    }
}

// ---------------------------------------------------------------------------
// M11  Matcher::relocateMinimum / refinement (refinement == 1)
//      matcher.cpp:1666-1711, 1715-1821; computeSmallDescriptor :583-611
// ---------------------------------------------------------------------------
__device__ __forceinline__ void small_desc(const uint8_t* du, const uint8_t* dv, int bpl, int u, int v,
                                           uint32_t d[4]) {
    const ptrdiff_t a2 = (ptrdiff_t)v * bpl + u, a1 = a2 - bpl, a0 = a1 - bpl, a3 = a2 + bpl, a4 = a3 + bpl;
    d[0] = du[a0] | (du[a1 - 2] << 8) | (du[a1] << 16) | ((uint32_t)du[a1 + 2] << 24);
    d[1] = du[a2 - 1] | (du[a2] << 8) | (du[a2] << 16) | ((uint32_t)du[a2 + 1] << 24);
    d[2] = du[a3 - 2] | (du[a3] << 8) | (du[a3 + 2] << 16) | ((uint32_t)du[a4] << 24);
    d[3] = dv[a1] | (dv[a2 - 1] << 8) | (dv[a2 + 1] << 16) | ((uint32_t)dv[a3] << 24);
}

__device__ void relocate(const SobelView& s1, const SobelView& s2, int margin, float u1, float v1,
                         float* u2, float* v2) {
    if (*u2 - 2 < margin || *u2 + 2 > s2.w - 1 - margin || *v2 - 2 < margin || *v2 + 2 > s2.h - 1 - margin)
        return;
    uint32_t ref[4], d[4];
    small_desc(s1.du, s1.dv, s1.bpl, (int)u1, (int)v1, ref);
    int best = 0, best_cost = 0;
    for (int k = 0; k < 25; k++) {
        small_desc(s2.du, s2.dv, s2.bpl, (int)*u2 + k % 5 - 2, (int)*v2 + k / 5 - 2, d);
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) c = __builtin_amdgcn_sad_u8(ref[q], d[q], c);
        if (k == 0 || (int)c < best_cost) {
            best = k;
            best_cost = (int)c;
        }
    }
    // u2 += (float)(min_ind%5) - 2.0  (double arithmetic, then back to float)
    *u2 = (float)((double)*u2 + ((double)(float)(best % 5) - 2.0));
    *v2 = (float)((double)*v2 + ((double)(float)(best / 5) - 2.0));
}

// relocateMinimum with 32 lanes per match: lane k < 25 evaluates search position k, the winner is
// the smallest (cost, k) pair == the reference's first minimum of its k = 0..24 scan
__device__ __forceinline__ void relocate_group(const SobelView& s1, const SobelView& s2, int margin, float u1,
                                               float v1, float* u2, float* v2, int lane) {
    // group-uniform exit (all 32 lanes hold the same match)
    if (*u2 - 2 < margin || *u2 + 2 > s2.w - 1 - margin || *v2 - 2 < margin || *v2 + 2 > s2.h - 1 - margin)
        return;
    int key = 0x7FFFFFFF;
    if (lane < 25) {
        uint32_t ref[4], d[4];
        small_desc(s1.du, s1.dv, s1.bpl, (int)u1, (int)v1, ref);
        small_desc(s2.du, s2.dv, s2.bpl, (int)*u2 + lane % 5 - 2, (int)*v2 + lane / 5 - 2, d);
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) c = __builtin_amdgcn_sad_u8(ref[q], d[q], c);
        key = (int)(c << 5) | lane;   // cost <= 4080
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        const int o = __shfl_xor(key, m, 32);
        key = o < key ? o : key;
    }
    const int best = key & 31;
    *u2 = (float)((double)*u2 + ((double)(float)(best % 5) - 2.0));
    *v2 = (float)((double)*v2 + ((double)(float)(best / 5) - 2.0));
}

__device__ __forceinline__ void d_refine_group(svh_p_match* __restrict__ m,
                                                      const int32_t* __restrict__ count, int method, int margin,
                                                      SobelView s1p, SobelView s2p, SobelView s1c,
                                                      SobelView s2c, unsigned bx) {
    const int i = (int)(bx * 256 + threadIdx.x) >> 5, lane = threadIdx.x & 31;
    if (i >= *count) return;   // whole groups leave together
    svh_p_match q = m[i];
    if (method == 0 || method == 2) relocate_group(s1c, s1p, margin, q.u1c, q.v1c, &q.u1p, &q.v1p, lane);
    if (method == 1 || method == 2) relocate_group(s1c, s2c, margin, q.u1c, q.v1c, &q.u2c, &q.v2c, lane);
    if (method == 2) relocate_group(s1c, s2p, margin, q.u1c, q.v1c, &q.u2p, &q.v2p, lane);
    if (lane == 0) m[i] = q;
}

// Matrix::solve, 6x6 with one right-hand side   libviso2/src/matrix.cpp:648-760
// The matrix of parabolicFitting is A^T A of a CONSTANT 9x6 design matrix (matcher.cpp:1725-1733), so
// the Gauss-Jordan elimination with full pivoting takes the same pivots and the same multipliers for
// every match: they are evaluated at compile time (IEEE double, one rounding per operation, no
// contraction -- the reference's operation sequence), and the kernel only replays what the elimination
// does to the right-hand side: swap, scale by 1/pivot, subtract multiples -- 36 double operations on
// registers instead of a 6x6 elimination with data-dependent indexing (which lived in scratch).
struct Solve6Prog {
    int irow[6], icol[6];
    double pivinv[6];
    double dum[6][6];
    bool ok;
};
constexpr double kFA[9][6] = {{1, 1, 1, -1, -1, 1}, {0, 1, 0, 0, -1, 1}, {1, 1, -1, 1, -1, 1},
                              {1, 0, 0, -1, 0, 1},  {0, 0, 0, 0, 0, 1},  {1, 0, 0, 1, 0, 1},
                              {1, 1, -1, -1, 1, 1}, {0, 1, 0, 0, 1, 1},  {1, 1, 1, 1, 1, 1}};
constexpr double cabs_(double x) { return x < 0 ? -x : x; }
constexpr Solve6Prog make_solve6_prog() {
    Solve6Prog P{};
    double A[36] = {};
    for (int r = 0; r < 6; r++)
        for (int e = 0; e < 6; e++) {
            double s = 0.0;
            for (int q = 0; q < 9; q++) s = s + kFA[q][r] * kFA[q][e];
            A[r * 6 + e] = s;
        }
    int ipiv[6] = {0, 0, 0, 0, 0, 0};
    P.ok = true;
    for (int it = 0; it < 6; it++) {
        double big = 0.0;
        int irow = 0, icol = 0;
        for (int j = 0; j < 6; j++)
            if (ipiv[j] != 1)
                for (int q = 0; q < 6; q++)
                    if (ipiv[q] == 0 && cabs_(A[j * 6 + q]) >= big) {
                        big = cabs_(A[j * 6 + q]);
                        irow = j;
                        icol = q;
                    }
        ++ipiv[icol];
        if (irow != icol)
            for (int l = 0; l < 6; l++) {
                const double t = A[irow * 6 + l];
                A[irow * 6 + l] = A[icol * 6 + l];
                A[icol * 6 + l] = t;
            }
        P.irow[it] = irow;
        P.icol[it] = icol;
        if (cabs_(A[icol * 6 + icol]) < 1e-20) P.ok = false;
        const double pivinv = 1.0 / A[icol * 6 + icol];
        P.pivinv[it] = pivinv;
        A[icol * 6 + icol] = 1.0;
        for (int l = 0; l < 6; l++) A[icol * 6 + l] = A[icol * 6 + l] * pivinv;
        for (int ll = 0; ll < 6; ll++) {
            P.dum[it][ll] = 0.0;
            if (ll != icol) {
                const double dum = A[ll * 6 + icol];
                P.dum[it][ll] = dum;
                A[ll * 6 + icol] = 0.0;
                for (int l = 0; l < 6; l++) A[ll * 6 + l] = A[ll * 6 + l] - A[icol * 6 + l] * dum;
            }
        }
    }
    return P;
}
constexpr Solve6Prog kSolve6 = make_solve6_prog();
static_assert(kSolve6.ok, "A^T A of the quadratic fit is regular");

__device__ __forceinline__ void solve6_rhs(double (&B)[6]) {
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int irow = kSolve6.irow[it], icol = kSolve6.icol[it];
        if (irow != icol) {
            const double t = B[irow];
            B[irow] = B[icol];
            B[icol] = t;
        }
        B[icol] = __dmul_rn(B[icol], kSolve6.pivinv[it]);
#pragma unroll
        for (int ll = 0; ll < 6; ll++)
            if (ll != icol) B[ll] = __dsub_rn(B[ll], __dmul_rn(B[icol], kSolve6.dum[it][ll]));
    }
}

// M11'  Matcher::parabolicFitting   matcher.cpp:1574-1662 (refinement == 2); false drops the match
__device__ bool parabolic(const SobelView& s1, const SobelView& s2, int margin, float u1, float v1,
                          float* u2, float* v2) {
    if (*u2 - 3 < margin || *u2 + 3 > s2.w - 1 - margin || *v2 - 3 < margin || *v2 + 3 > s2.h - 1 - margin)
        return false;
    uint32_t ref[4], d[4];
    small_desc(s1.du, s1.dv, s1.bpl, (int)u1, (int)v1, ref);
    auto cost_at = [&](int q7u, int q7v) {
        small_desc(s2.du, s2.dv, s2.bpl, (int)*u2 + q7u - 3, (int)*v2 + q7v - 3, d);
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) c = __builtin_amdgcn_sad_u8(ref[w], d[w], c);
        return (int)c;
    };
    int min_ind = 0, min_cost = 0;
    for (int q = 0; q < 49; q++) {
        const int c = cost_at(q % 7, q / 7);
        if (q == 0 || c < min_cost) {
            min_ind = q;
            min_cost = c;
        }
    }
    const int du = min_ind % 7, dv = min_ind / 7;
    if (du == 0 || du == 6 || dv == 0 || dv == 6) return false;
    // the 3x3 costs around the minimum (evaluated again: nine small descriptors instead of a
    // 49-entry array with a data-dependent index), design matrix of the fit: matcher.cpp:1725-1733
    double c9[9], b[6];
#pragma unroll
    for (int i = -1; i <= 1; i++)
#pragma unroll
        for (int j = -1; j <= 1; j++) c9[(i + 1) * 3 + (j + 1)] = (double)cost_at(du + j, dv + i);
#pragma unroll
    for (int r = 0; r < 6; r++) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 9; q++) acc = __dadd_rn(acc, __dmul_rn(kFA[q][r], c9[q]));
        b[r] = acc;
    }
    solve6_rhs(b);
    const float divisor = (float)__dsub_rn(__dmul_rn(b[2], b[2]), __dmul_rn(__dmul_rn(4.0, b[0]), b[1]));
    if ((double)fabsf(divisor) < 1e-8 || fabs(b[2]) < 1e-8) return false;
    const float ddv = (float)__ddiv_rn(
        __dsub_rn(__dmul_rn(__dmul_rn(2.0, b[0]), b[4]), __dmul_rn(b[2], b[3])), (double)divisor);
    const float ddu = (float)__ddiv_rn(
        -__dadd_rn(b[4], __dmul_rn(__dmul_rn(2.0, b[1]), (double)ddv)), b[2]);
    if ((double)fabsf(ddu) >= 1.0 || (double)fabsf(ddv) >= 1.0) return false;
    *u2 = (float)__dadd_rn((double)*u2, __dadd_rn(__dsub_rn((double)(float)du, 3.0), (double)ddu));
    *v2 = (float)__dadd_rn((double)*v2, __dadd_rn(__dsub_rn((double)(float)dv, 3.0), (double)ddv));
    return true;
}

template <bool kParabolic>
__device__ __forceinline__ void d_refine(svh_p_match* __restrict__ m,
                                                const int32_t* __restrict__ count, int method, int margin,
                                                SobelView s1p, SobelView s2p, SobelView s1c,
                                                SobelView s2c, int32_t* __restrict__ flags, unsigned bx) {
    const int i = bx * 128 + threadIdx.x;
    if (i >= *count) return;
    svh_p_match q = m[i];
    bool ok = true;
    if (kParabolic) {
        if (method == 0 || method == 2) ok = parabolic(s1c, s1p, margin, q.u1c, q.v1c, &q.u1p, &q.v1p);
        if (ok && (method == 1 || method == 2)) ok = parabolic(s1c, s2c, margin, q.u1c, q.v1c, &q.u2c, &q.v2c);
        if (ok && method == 2) ok = parabolic(s1c, s2p, margin, q.u1c, q.v1c, &q.u2p, &q.v2p);
        flags[i] = ok ? 1 : 0;
    } else {
        if (method == 0 || method == 2) relocate(s1c, s1p, margin, q.u1c, q.v1c, &q.u1p, &q.v1p);
        if (method == 1 || method == 2) relocate(s1c, s2c, margin, q.u1c, q.v1c, &q.u2c, &q.v2c);
        if (method == 2) relocate(s1c, s2p, margin, q.u1c, q.v1c, &q.u2p, &q.v2p);
    }
    if (ok) m[i] = q;
}


// ---------------------------------------------------------------------------
// Kernels: every body above is a __device__ function of (arguments, block index); each gets a plain
// __global__ form (one object: arguments by value) and a BATCHED form (K objects in lockstep, batch_rec.h:
// arguments of job blockIdx.z read from a job table in device memory, grid = the largest job's).
// ---------------------------------------------------------------------------
// Pointers that arrive through a job table in memory have lost their address space: hipcc then reads and writes
// through FLAT instructions (round 4: 913 of them in the lockstep kernels, 514 in k_refine_parabolic_b alone).  Every
// buffer of a job is device (or device-mapped pinned host) memory: a round trip through address space 1 tells the
// compiler so and the kernels use global_load / global_store like their single-object forms.
template <class T>
__device__ __forceinline__ T* gptr(T* p) {
    // (the empty asm keeps the address-space-1 value opaque: a plain generic -> global -> generic cast pair is folded
    // away before the compiler's address-space inference sees it; held in a vector register pair)
    __attribute__((address_space(1))) T* q = (__attribute__((address_space(1))) T*)p;
    asm volatile("" : "+v"(q));
    return (T*)q;
}
__device__ __forceinline__ FeatView gview(const FeatView& v) {
    FeatView o;
    o.rec = gptr(v.rec); o.count = gptr(v.count); o.off = gptr(v.off); o.ids = gptr(v.ids);
    return o;
}
__device__ __forceinline__ SobelView gview(const SobelView& v) {
    SobelView o = v;
    o.du = gptr(v.du); o.dv = gptr(v.dv);
    return o;
}

struct HalfJob { const uint8_t* I; int bpl; uint8_t* out; int hw, hh, hbpl; };
__global__ __launch_bounds__(256) void k_half(HalfJob a) { d_half(a.I, a.bpl, a.out, a.hw, a.hh, a.hbpl, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(256) void k_half_b(const HalfJob* J) {
    const HalfJob a = J[blockIdx.z];
    d_half(gptr(a.I), a.bpl, gptr(a.out), a.hw, a.hh, a.hbpl, blockIdx.x, blockIdx.y);
}

struct FiltersJob { const uint8_t* I; int w, h, bpl; uint8_t *du, *dv; int16_t *f1, *f2; };
template <bool kFeatures>
__global__ __launch_bounds__(256) void k_filters(FiltersJob a) {
    d_filters<kFeatures>(a.I, a.w, a.h, a.bpl, a.du, a.dv, a.f1, a.f2, blockIdx.x, blockIdx.y);
}
template <bool kFeatures>
__global__ __launch_bounds__(256) void k_filters_b(const FiltersJob* J) {
    const FiltersJob a = J[blockIdx.z];
    d_filters<kFeatures>(gptr(a.I), a.w, a.h, a.bpl, gptr(a.du), gptr(a.dv), gptr(a.f1), gptr(a.f2), blockIdx.x, blockIdx.y);
}

// the half-resolution image and the full-resolution Sobel planes read the same uploaded image and nothing of each
// other: one launch (single-object path), workgroups [0, gh) halve, the rest filter
__global__ __launch_bounds__(256) void k_half_filters(HalfJob a, FiltersJob b, int ghx, int gh, int gfx) {
    const int id = (int)blockIdx.x;
    if (id < gh) {
        d_half(a.I, a.bpl, a.out, a.hw, a.hh, a.hbpl, (unsigned)(id % ghx), (unsigned)(id / ghx));
    } else {
        const int k = id - gh;
        d_filters<false>(b.I, b.w, b.h, b.bpl, b.du, b.dv, b.f1, b.f2, (unsigned)(k % gfx), (unsigned)(k / gfx));
    }
}

struct NmsJob { const int16_t *f1, *f2; int w, h, bpl, n, tau, margin, ni, nj; int4* slots; int32_t* flags; };
template <int kG>
__global__ __launch_bounds__(256) void k_nms(NmsJob a) {
    d_nms<kG>(a.f1, a.f2, a.w, a.h, a.bpl, a.n, a.tau, a.margin, a.ni, a.nj, a.slots, a.flags, blockIdx.x);
}
template <int kG>
__global__ __launch_bounds__(256) void k_nms_b(const NmsJob* J) {
    const NmsJob a = J[blockIdx.z];
    d_nms<kG>(gptr(a.f1), gptr(a.f2), a.w, a.h, a.bpl, a.n, a.tau, a.margin, a.ni, a.nj, gptr(a.slots), gptr(a.flags), blockIdx.x);
}

// the sparse and the dense table of one camera image by the same three launches (single-object path: every launch is
// on the frame's critical path, and the two tables have nothing to wait for in each other): workgroups [0, ga) belong
// to job a, the rest to job b
__global__ __launch_bounds__(256) void k_nms2(NmsJob a, NmsJob b, int ga, int small_a, int small_b) {
    const bool first = (int)blockIdx.x < ga;
    const NmsJob& j = first ? a : b;
    const unsigned bx = first ? blockIdx.x : blockIdx.x - ga;
    if (first ? small_a : small_b)
        d_nms<16>(j.f1, j.f2, j.w, j.h, j.bpl, j.n, j.tau, j.margin, j.ni, j.nj, j.slots, j.flags, bx);
    else
        d_nms<64>(j.f1, j.f2, j.w, j.h, j.bpl, j.n, j.tau, j.margin, j.ni, j.nj, j.slots, j.flags, bx);
}
struct CompactSlotsJob { const int32_t* flags; int nslots; int32_t *order, *count; };
__global__ __launch_bounds__(1024) void k_compact_slots(CompactSlotsJob a) { d_compact_slots(a.flags, a.nslots, a.order, a.count, 0u); }
__global__ __launch_bounds__(1024) void k_compact_slots_b(const CompactSlotsJob* J) {
    const CompactSlotsJob a = J[blockIdx.z];
    d_compact_slots(gptr(a.flags), a.nslots, gptr(a.order), gptr(a.count), 0u);
}

__global__ __launch_bounds__(1024) void k_compact_slots2(CompactSlotsJob a, CompactSlotsJob b, int32_t* host_a, int32_t* host_b) {
    const CompactSlotsJob& j = blockIdx.x == 0 ? a : b;
    d_compact_slots(j.flags, j.nslots, j.order, j.count, 0u, blockIdx.x == 0 ? host_a : host_b);
}
struct FeatureRecordsJob { const int4* slots; const int32_t *order, *count; const uint8_t *du, *dv; int bpl, scale; int32_t* table; };
__global__ __launch_bounds__(256) void k_feature_records(FeatureRecordsJob a) {
    d_feature_records(a.slots, a.order, a.count, a.du, a.dv, a.bpl, a.scale, a.table, blockIdx.x);
}
__global__ __launch_bounds__(256) void k_feature_records2(FeatureRecordsJob a, FeatureRecordsJob b, int ga) {
    const bool first = (int)blockIdx.x < ga;
    const FeatureRecordsJob& j = first ? a : b;
    d_feature_records(j.slots, j.order, j.count, j.du, j.dv, j.bpl, j.scale, j.table, first ? blockIdx.x : blockIdx.x - ga);
}
__global__ __launch_bounds__(256) void k_feature_records_b(const FeatureRecordsJob* J) {
    const FeatureRecordsJob a = J[blockIdx.z];
    d_feature_records(gptr(a.slots), gptr(a.order), gptr(a.count), gptr(a.du), gptr(a.dv), a.bpl, a.scale, gptr(a.table), blockIdx.x);
}

struct BinIndexJob { BinJobs J; int njobs, ub, vb, binsize; };
__global__ __launch_bounds__(1024) void k_bin_index_lds(BinJobs J, int ub, int vb, int binsize) {
    const unsigned bx = blockIdx.x;
    d_bin_index_lds(J.table[bx], J.count[bx], J.off[bx], J.ids[bx], ub, vb, binsize);
}
__global__ __launch_bounds__(1024) void k_bin_index_lds_b(const BinIndexJob* J) {
    const BinIndexJob& a = J[blockIdx.z];
    const unsigned bx = blockIdx.x;
    if ((int)bx >= a.njobs) return;
    d_bin_index_lds(gptr(a.J.table[bx]), gptr(a.J.count[bx]), gptr(a.J.off[bx]), gptr(a.J.ids[bx]), a.ub, a.vb, a.binsize);
}

struct MatchJob {
    MatchParams P;
    FeatView m1p, m2p, m1c, m2c;
    const float* ranges;
    int use_prior;
    svh_p_match* out;
    int32_t *flags, *pixel_owner;
};
__global__ __launch_bounds__(128) void k_match(MatchJob a) {
    d_match(a.P, a.m1p, a.m2p, a.m1c, a.m2c, a.ranges, a.use_prior, a.out, a.flags, a.pixel_owner, blockIdx.x);
}
__global__ __launch_bounds__(128) void k_match_b(const MatchJob* J) {
    const MatchJob& a = J[blockIdx.z];
    d_match(a.P, gview(a.m1p), gview(a.m2p), gview(a.m1c), gview(a.m2c), gptr(a.ranges), a.use_prior, gptr(a.out), gptr(a.flags),
            gptr(a.pixel_owner), blockIdx.x);
}

struct DedupeJob { const int32_t* n; int width; const svh_p_match* m; int32_t* flags; const int32_t* pixel_owner; };
__global__ __launch_bounds__(256) void k_match_dedupe(DedupeJob a) { d_match_dedupe(a.n, a.width, a.m, a.flags, a.pixel_owner, blockIdx.x); }
__global__ __launch_bounds__(256) void k_match_dedupe_b(const DedupeJob* J) {
    const DedupeJob a = J[blockIdx.z];
    d_match_dedupe(gptr(a.n), a.width, gptr(a.m), gptr(a.flags), gptr(a.pixel_owner), blockIdx.x);
}

struct CompactMatchesJob { const svh_p_match* in; const int32_t *flags, *nslots; svh_p_match* out; int32_t* count; };
__global__ __launch_bounds__(1024) void k_compact_matches(CompactMatchesJob a, int32_t* count_host) { d_compact_matches(a.in, a.flags, a.nslots, a.out, a.count, 0u, count_host); }
__global__ __launch_bounds__(1024) void k_compact_matches_b(const CompactMatchesJob* J) {
    const CompactMatchesJob a = J[blockIdx.z];
    d_compact_matches(gptr(a.in), gptr(a.flags), gptr(a.nslots), gptr(a.out), gptr(a.count), 0u);
}

struct RefineJob { svh_p_match* m; const int32_t* count; int method, margin; SobelView s1p, s2p, s1c, s2c; int32_t* flags; };
__global__ __launch_bounds__(256) void k_refine_group(RefineJob a) {
    d_refine_group(a.m, a.count, a.method, a.margin, a.s1p, a.s2p, a.s1c, a.s2c, blockIdx.x);
}
__global__ __launch_bounds__(256) void k_refine_group_b(const RefineJob* J) {
    const RefineJob& a = J[blockIdx.z];
    d_refine_group(gptr(a.m), gptr(a.count), a.method, a.margin, gview(a.s1p), gview(a.s2p), gview(a.s1c), gview(a.s2c), blockIdx.x);
}
__global__ __launch_bounds__(128) void k_refine_parabolic(RefineJob a) {
    d_refine<true>(a.m, a.count, a.method, a.margin, a.s1p, a.s2p, a.s1c, a.s2c, a.flags, blockIdx.x);
}
__global__ __launch_bounds__(128) void k_refine_parabolic_b(const RefineJob* J) {
    const RefineJob& a = J[blockIdx.z];
    d_refine<true>(gptr(a.m), gptr(a.count), a.method, a.margin, gview(a.s1p), gview(a.s2p), gview(a.s1c), gview(a.s2c), gptr(a.flags),
                   blockIdx.x);
}

// ---------------------------------------------------------------------------
// Image upload: the rows were packed into PINNED host memory; the device reads
// them over PCIe with 16-byte loads and writes its HBM copy.  One kernel launch
// instead of a hipMemcpyAsync (whose submission alone costs ~90 us per image).
// The batched entries move every small transfer of a phase this way (feature counts and match lists to
// pinned host memory, search ranges to the device): words of 4 bytes, any direction the device can address.
// ---------------------------------------------------------------------------
struct UploadJob { const uint4* host; uint4* dev; size_t n16; };
__global__ __launch_bounds__(256) void k_upload(UploadJob a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n16) a.dev[i] = a.host[i];
}
__global__ __launch_bounds__(256) void k_upload_b(const UploadJob* J) {
    const UploadJob a = J[blockIdx.z];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n16) gptr(a.dev)[i] = gptr(a.host)[i];
}
struct Copy4Job { uint32_t* dst; const uint32_t* src; size_t n4; };
__global__ __launch_bounds__(256) void k_copy4(Copy4Job a) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * 256) a.dst[i] = a.src[i];
}
__global__ __launch_bounds__(256) void k_copy4_b(const Copy4Job* J) {
    const Copy4Job a = J[blockIdx.z];
    uint32_t* const dst = gptr(a.dst);
    const uint32_t* const src = gptr(a.src);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
struct Fill4Job { uint32_t* dst; uint32_t value; size_t n4; };
__global__ __launch_bounds__(256) void k_fill4_b(const Fill4Job* J) {
    const Fill4Job a = J[blockIdx.z];
    uint32_t* const dst = gptr(a.dst);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * 256) dst[i] = a.value;
}

// batched launch entries (BatchLaunchFn): jobs = device copy of the table, grid (gx, gy, njobs)
#define SVH_BATCH_FN(name, kernel, Job, threads)                                                                   \
    void name(const void* jobs, int njobs, unsigned gx, unsigned gy, size_t lds, hipStream_t s) {                  \
        hipLaunchKernelGGL(kernel, dim3(gx, gy, (unsigned)njobs), threads, lds, s, reinterpret_cast<const Job*>(jobs)); \
    }
SVH_BATCH_FN(b_half, k_half_b, HalfJob, dim3(64, 4))
SVH_BATCH_FN(b_filters0, k_filters_b<false>, FiltersJob, dim3(FX, 4))
SVH_BATCH_FN(b_filters1, k_filters_b<true>, FiltersJob, dim3(FX, 4))
SVH_BATCH_FN(b_nms16, k_nms_b<16>, NmsJob, dim3(256))
SVH_BATCH_FN(b_nms64, k_nms_b<64>, NmsJob, dim3(256))
SVH_BATCH_FN(b_compact_slots, k_compact_slots_b, CompactSlotsJob, dim3(1024))
SVH_BATCH_FN(b_feature_records, k_feature_records_b, FeatureRecordsJob, dim3(256))
SVH_BATCH_FN(b_bin_index, k_bin_index_lds_b, BinIndexJob, dim3(1024))
SVH_BATCH_FN(b_match, k_match_b, MatchJob, dim3(128))
SVH_BATCH_FN(b_dedupe, k_match_dedupe_b, DedupeJob, dim3(256))
SVH_BATCH_FN(b_compact_matches, k_compact_matches_b, CompactMatchesJob, dim3(1024))
SVH_BATCH_FN(b_refine_group, k_refine_group_b, RefineJob, dim3(256))
SVH_BATCH_FN(b_refine_parabolic, k_refine_parabolic_b, RefineJob, dim3(128))
SVH_BATCH_FN(b_upload, k_upload_b, UploadJob, dim3(256))
SVH_BATCH_FN(b_copy4, k_copy4_b, Copy4Job, dim3(256))
SVH_BATCH_FN(b_fill4, k_fill4_b, Fill4Job, dim3(256))

}  // namespace

// ---------------------------------------------------------------------------
// launchers: launch now, or -- while the calling thread records a batch (batch_rec.h) -- append a job
// ---------------------------------------------------------------------------
void mlaunch_upload(void* stream, const uint8_t* pinned, uint8_t* dev, size_t bytes) {
    const size_t n16 = bytes / 16;   // bpl is a multiple of 16
    const UploadJob a = {reinterpret_cast<const uint4*>(pinned), reinterpret_cast<uint4*>(dev), n16};
    const unsigned gx = (unsigned)((n16 + 255) / 256);
    if (t_rec) return t_rec->add(b_upload, a, gx);
    hipLaunchKernelGGL(k_upload, dim3(gx), dim3(256), 0, (hipStream_t)stream, a);
}

// small transfers between pinned host and device memory, either direction (bytes: a multiple of 4)
void mlaunch_copy(void* stream, void* dst, const void* src, size_t bytes, int kind) {
    const Copy4Job a = {static_cast<uint32_t*>(dst), static_cast<const uint32_t*>(src), bytes / 4};
    const unsigned gx = (unsigned)std::min<size_t>((bytes / 4 + 255) / 256, 64);
    if (t_rec) return t_rec->add(b_copy4, a, gx);
    // Small transfers go by a kernel as well: both ends are device-addressable (pinned host memory), and a copy
    // kernel in the stream's own queue spares the hand-over to a DMA engine and back (~15 us per copy; a frame has
    // five of them between kernels that wait for each other).
    static const bool by_kernel = !(svh::env("SVH_MATCHER_COPY_KERNEL") && atoi(svh::env("SVH_MATCHER_COPY_KERNEL")) == 0);
    if (by_kernel && bytes % 4 == 0 && bytes > 0 && bytes <= ((size_t)1 << 20)) {
        hipLaunchKernelGGL(k_copy4, dim3(gx), dim3(256), 0, (hipStream_t)stream, a);
        return;
    }
    (void)hipMemcpyAsync(dst, src, bytes, (hipMemcpyKind)kind, (hipStream_t)stream);
}
void mlaunch_fill(void* stream, void* dst, int byte_value, size_t bytes) {
    if (t_rec) {
        const uint32_t b = (uint32_t)(byte_value & 0xFF);
        const Fill4Job a = {static_cast<uint32_t*>(dst), b | b << 8 | b << 16 | b << 24, bytes / 4};
        return t_rec->add(b_fill4, a, (unsigned)std::min<size_t>((bytes / 4 + 255) / 256, 256));
    }
    (void)hipMemsetAsync(dst, byte_value, bytes, (hipStream_t)stream);
}

void mlaunch_half(void* stream, const uint8_t* I, int bpl, uint8_t* out, int hw, int hh, int hbpl) {
    const HalfJob a = {I, bpl, out, hw, hh, hbpl};
    const unsigned gx = (hw + 63) / 64, gy = (hh + 3) / 4;
    if (t_rec) return t_rec->add(b_half, a, gx, gy);
    hipLaunchKernelGGL(k_half, dim3(gx, gy), dim3(64, 4), 0, (hipStream_t)stream, a);
}

void mlaunch_filters(void* stream, const uint8_t* I, int w, int h, int bpl, uint8_t* du, uint8_t* dv,
                     int16_t* f1, int16_t* f2) {
    const FiltersJob a = {I, w, h, bpl, du, dv, f1, f2};
    const dim3 grid((bpl / 4 + FX - 1) / FX, (h + 4 * FR - 1) / (4 * FR)), block(FX, 4);
    if (t_rec) return t_rec->add(f1 ? b_filters1 : b_filters0, a, grid.x, grid.y);
    if (f1)
        hipLaunchKernelGGL(k_filters<true>, grid, block, 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_filters<false>, grid, block, 0, (hipStream_t)stream, a);
}

// k_half on I -> Ih and the Sobel planes of I (no feature images) by one launch
void mlaunch_half_filters(void* stream, const uint8_t* I, int w, int h, int bpl, uint8_t* Ih, int hw, int hh, int hbpl,
                          uint8_t* du_full, uint8_t* dv_full) {
    if (t_rec || FX != 64) {
        mlaunch_half(stream, I, bpl, Ih, hw, hh, hbpl);
        mlaunch_filters(stream, I, w, h, bpl, du_full, dv_full, nullptr, nullptr);
        return;
    }
    const HalfJob a = {I, bpl, Ih, hw, hh, hbpl};
    const FiltersJob b = {I, w, h, bpl, du_full, dv_full, nullptr, nullptr};
    const int ghx = (hw + 63) / 64, ghy = (hh + 3) / 4;
    const int gfx = (bpl / 4 + FX - 1) / FX, gfy = (h + 4 * FR - 1) / (4 * FR);
    hipLaunchKernelGGL(k_half_filters, dim3(ghx * ghy + gfx * gfy), dim3(64, 4), 0, (hipStream_t)stream, a, b, ghx,
                       ghx * ghy, gfx);
}

int mnms_blocks(int extent, int n, int margin) {
    int c = 0;
    for (int i = n + margin; i < extent - n - margin; i += n + 1) c++;
    return c;
}

void mlaunch_features(void* stream, const int16_t* f1, const int16_t* f2, const uint8_t* du,
                      const uint8_t* dv, int w, int h, int bpl, int n, int tau, int margin, int scale,
                      int4* slots, int32_t* flags, int32_t* order, int32_t* table, int32_t* count) {
    hipStream_t s = (hipStream_t)stream;
    const int ni = mnms_blocks(w, n, margin), nj = mnms_blocks(h, n, margin);
    const int nb = ni * nj;
    const NmsJob an = {f1, f2, w, h, bpl, n, tau, margin, ni, nj, slots, flags};
    const CompactSlotsJob ac = {flags, nb * 4, order, count};
    const FeatureRecordsJob af = {slots, order, count, du, dv, bpl, scale, table};
    const bool small = (n + 1) * (n + 1) <= 16;
    if (t_rec) {
        if (nb > 0) t_rec->add(small ? b_nms16 : b_nms64, an, small ? (nb + 15) / 16 : (nb + 3) / 4);
        t_rec->add(b_compact_slots, ac, 1);
        if (nb > 0) t_rec->add(b_feature_records, af, (nb * 4 * 8 + 255) / 256);
        return;
    }
    if (nb > 0) {
        if (small)
            hipLaunchKernelGGL(k_nms<16>, dim3((nb + 15) / 16), dim3(256), 0, s, an);
        else
            hipLaunchKernelGGL(k_nms<64>, dim3((nb + 3) / 4), dim3(256), 0, s, an);
    }
    hipLaunchKernelGGL(k_compact_slots, dim3(1), dim3(1024), 0, s, ac);
    if (nb > 0) hipLaunchKernelGGL(k_feature_records, dim3((nb * 4 * 8 + 255) / 256), dim3(256), 0, s, af);
}

// both tables of a camera image (n_a: the sparse table's NMS radius, n_b: the dense one's); scratch set per table;
// the two counts also go to host_counts[0..1] (pinned), written by the compaction itself
void mlaunch_features2(void* stream, const int16_t* f1, const int16_t* f2, const uint8_t* du, const uint8_t* dv, int w,
                       int h, int bpl, int tau, int margin, int scale, int n_a, int4* slots_a, int32_t* flags_a,
                       int32_t* order_a, int32_t* table_a, int32_t* count_a, int n_b, int4* slots_b, int32_t* flags_b,
                       int32_t* order_b, int32_t* table_b, int32_t* count_b, int32_t* host_counts) {
    hipStream_t s = (hipStream_t)stream;
    const int ni_a = mnms_blocks(w, n_a, margin), nj_a = mnms_blocks(h, n_a, margin), nb_a = ni_a * nj_a;
    const int ni_b = mnms_blocks(w, n_b, margin), nj_b = mnms_blocks(h, n_b, margin), nb_b = ni_b * nj_b;
    if (t_rec || nb_a <= 0 || nb_b <= 0) {   // (recorded batches and degenerate images: the two plain sequences)
        mlaunch_features(stream, f1, f2, du, dv, w, h, bpl, n_a, tau, margin, scale, slots_a, flags_a, order_a, table_a, count_a);
        mlaunch_features(stream, f1, f2, du, dv, w, h, bpl, n_b, tau, margin, scale, slots_b, flags_b, order_b, table_b, count_b);
        // (count_a and count_b are consecutive words of the view: one copy, as the callers of mlaunch_features do)
        mlaunch_copy(stream, host_counts, count_a, 2 * sizeof(int32_t), hipMemcpyDeviceToHost);
        return;
    }
    const NmsJob na = {f1, f2, w, h, bpl, n_a, tau, margin, ni_a, nj_a, slots_a, flags_a};
    const NmsJob nb = {f1, f2, w, h, bpl, n_b, tau, margin, ni_b, nj_b, slots_b, flags_b};
    const bool small_a = (n_a + 1) * (n_a + 1) <= 16, small_b = (n_b + 1) * (n_b + 1) <= 16;
    const int ga = small_a ? (nb_a + 15) / 16 : (nb_a + 3) / 4, gb = small_b ? (nb_b + 15) / 16 : (nb_b + 3) / 4;
    hipLaunchKernelGGL(k_nms2, dim3(ga + gb), dim3(256), 0, s, na, nb, ga, small_a ? 1 : 0, small_b ? 1 : 0);
    const CompactSlotsJob ca = {flags_a, nb_a * 4, order_a, count_a}, cb = {flags_b, nb_b * 4, order_b, count_b};
    hipLaunchKernelGGL(k_compact_slots2, dim3(2), dim3(1024), 0, s, ca, cb, host_counts, host_counts + 1);
    const FeatureRecordsJob fa = {slots_a, order_a, count_a, du, dv, bpl, scale, table_a};
    const FeatureRecordsJob fb = {slots_b, order_b, count_b, du, dv, bpl, scale, table_b};
    const int ra = (nb_a * 4 * 8 + 255) / 256, rb = (nb_b * 4 * 8 + 255) / 256;
    hipLaunchKernelGGL(k_feature_records2, dim3(ra + rb), dim3(256), 0, s, fa, fb, ra);
}

void mlaunch_bin_index(void* stream, const BinJobs& J, int njobs, int n_host_max, int ub, int vb, int binsize,
                       int32_t* cursor) {
    const int nb = 4 * ub * vb;
    const size_t lds = ((size_t)2 * nb + 1 + (size_t)std::max(n_host_max, 0)) * sizeof(int32_t);
    if (t_rec) {
        // (the batched form has the LDS build only, with the 160 KB opt-in: ~38 k features per table)
        BinIndexJob a;
        a.J = J; a.njobs = njobs; a.ub = ub; a.vb = vb; a.binsize = binsize;
        static bool attr_once = ((void)hipFuncSetAttribute((const void*)k_bin_index_lds_b,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024), true);
        (void)attr_once;
        if (lds > 156 * 1024) t_rec->broken = true;
        return t_rec->add(b_bin_index, a, 8, 1, lds);
    }
    if (lds <= 56 * 1024) {
        hipLaunchKernelGGL(k_bin_index_lds, dim3(njobs), dim3(1024), lds, (hipStream_t)stream, J, ub, vb, binsize);
    } else {
        for (int j = 0; j < njobs; j++)
            hipLaunchKernelGGL(k_bin_index, dim3(1), dim3(1024), 0, (hipStream_t)stream, J.table[j], J.count[j], ub,
                               vb, binsize, J.off[j], J.ids[j], cursor);
    }
}

void mlaunch_match(void* stream, const MatchParams& P, const FeatView& m1p, const FeatView& m2p,
                   const FeatView& m1c, const FeatView& m2c, int nquery_cap, const float* ranges,
                   int use_prior, svh_p_match* slots, int32_t* flags, int32_t* pixel_owner,
                   svh_p_match* out, int32_t* out_count, int32_t* out_count_host) {
    hipStream_t s = (hipStream_t)stream;
    const FeatView& q = P.method == 2 ? m1p : m1c;
    if (P.method < 2) mlaunch_fill(stream, pixel_owner, 0x7F, (size_t)P.width * P.height * sizeof(int32_t));
    MatchJob am;
    am.P = P; am.m1p = m1p; am.m2p = m2p; am.m1c = m1c; am.m2c = m2c; am.ranges = ranges; am.use_prior = use_prior;
    am.out = slots; am.flags = flags; am.pixel_owner = pixel_owner;
    const DedupeJob ad = {q.count, P.width, slots, flags, pixel_owner};
    const CompactMatchesJob ac = {slots, flags, q.count, out, out_count};
    if (t_rec) {
        if (nquery_cap > 0) {
            t_rec->add(b_match, am, (nquery_cap * kQ + 127) / 128);
            if (P.method < 2) t_rec->add(b_dedupe, ad, (nquery_cap + 255) / 256);
        }
        return t_rec->add(b_compact_matches, ac, 1);
    }
    if (nquery_cap > 0) {
        hipLaunchKernelGGL(k_match, dim3((nquery_cap * kQ + 127) / 128), dim3(128), 0, s, am);
        if (P.method < 2) hipLaunchKernelGGL(k_match_dedupe, dim3((nquery_cap + 255) / 256), dim3(256), 0, s, ad);
    }
    hipLaunchKernelGGL(k_compact_matches, dim3(1), dim3(1024), 0, s, ac, out_count_host);
}

void mlaunch_refine(void* stream, svh_p_match* m, const int32_t* count, int cap, int method, int margin,
                    const SobelView& s1p, const SobelView& s2p, const SobelView& s1c,
                    const SobelView& s2c, int parabolic, int32_t* flags, svh_p_match* compacted,
                    int32_t* compacted_count) {
    hipStream_t s = (hipStream_t)stream;
    const RefineJob ar = {m, count, method, margin, s1p, s2p, s1c, s2c, flags};
    if (!parabolic) {
        if (cap > 0) {
            const unsigned gx = (unsigned)(((size_t)cap * 32 + 255) / 256);
            if (t_rec) return t_rec->add(b_refine_group, ar, gx);
            hipLaunchKernelGGL(k_refine_group, dim3(gx), dim3(256), 0, s, ar);
        }
        return;
    }
    // matches whose fit failed are dropped, order preserved (matcher.cpp:1766-1816)
    const CompactMatchesJob ac = {m, flags, count, compacted, compacted_count};
    if (t_rec) {
        if (cap > 0) t_rec->add(b_refine_parabolic, ar, (cap + 127) / 128);
        return t_rec->add(b_compact_matches, ac, 1);
    }
    if (cap > 0) hipLaunchKernelGGL(k_refine_parabolic, dim3((cap + 127) / 128), dim3(128), 0, s, ar);
    hipLaunchKernelGGL(k_compact_matches, dim3(1), dim3(1024), 0, s, ac, (int32_t*)nullptr);
}

}  // namespace svh
