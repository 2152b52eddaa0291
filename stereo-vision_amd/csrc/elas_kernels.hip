// HIP kernels (gfx950 / CDNA4) of the ELAS dense-disparity path.
//
// All of this is integer SAD / rank / stencil work: no MFMA.  The two matching kernels are
// bound by VALU issue (v_sad_u8 and its bookkeeping, measured with the SQ counters), the
// rest by HBM traffic and latency.  Design rules followed throughout:
//   * wavefront = 64 lanes; every per-pixel kernel maps lanes to consecutive u
//     so descriptor traffic is 16 B/lane = 1 KiB per wave instruction;
//   * 8-bit tiles are staged through LDS once and gathered from there;
//   * SAD uses v_sad_u8 (__builtin_amdgcn_sad_u8), arg-min over disparities is
//     a packed (cost<<16|d) wave reduction on DPP (v_min_u32_dpp);
//   * workgroup ids are mapped to work XCD-aware where neighbouring blocks re-read data;
//   * passes over the maps are fused when a tile (or a row) holds all inputs of the next one;
//   * float expressions that decide pixel ownership or d_plane are written with
//     __fmul_rn/__fadd_rn so they are never contracted into FMAs (the reference
//     is built -msse3 without FMA).
// Each kernel cites the reference lines whose result it reproduces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "svh_internal.h"

namespace svh {

namespace {

constexpr int kWave = 64;

// Wave-wide unsigned minimum, result uniform in all lanes.  Six DPP steps on the
// VALU (quad swaps, half-row / row mirrors, then the two row broadcasts of GFX9)
// instead of six ds_bpermute round trips through the LDS crossbar.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_min_step(uint32_t v) {
    // lanes without a valid source (or masked rows) get the identity of min, which lets the
    // compiler fold the DPP move into the v_min_u32 itself (one VALU op per step)
    const uint32_t o =
        (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, kCtrl, kRowMask, 0xF, false);
    return o < v ? o : v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_min_step<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_min_step<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_min_step<0x141, 0xF>(v);   // row_half_mirror
    v = dpp_min_step<0x140, 0xF>(v);   // row_mirror        -> every lane: min of its row of 16
    v = dpp_min_step<0x142, 0xA>(v);   // row_bcast:15      -> rows 1,3 fold in rows 0,2
    v = dpp_min_step<0x143, 0xC>(v);   // row_bcast:31      -> row 3 (lane 63) holds the minimum
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// minimum over each row of 16 lanes, in every lane of that row
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) {
    v = dpp_min_step<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_min_step<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_min_step<0x141, 0xF>(v);   // row_half_mirror
    v = dpp_min_step<0x140, 0xF>(v);   // row_mirror
    return v;
}

// sum |a.byte - b.byte| over 16 bytes (== psadbw lanes 0+4, elas.cpp:406-414)
__device__ __forceinline__ uint32_t sad16(const uint4& a, const uint4& b) {
    uint32_t s = __builtin_amdgcn_sad_u8(a.x, b.x, 0u);
    s = __builtin_amdgcn_sad_u8(a.y, b.y, s);
    s = __builtin_amdgcn_sad_u8(a.z, b.z, s);
    s = __builtin_amdgcn_sad_u8(a.w, b.w, s);
    return s;
}

// same, continuing a running sum (v_sad_u8 adds its third operand for free)
__device__ __forceinline__ uint32_t sad16_acc(const uint4& a, const uint4& b, uint32_t s) {
    s = __builtin_amdgcn_sad_u8(a.x, b.x, s);
    s = __builtin_amdgcn_sad_u8(a.y, b.y, s);
    s = __builtin_amdgcn_sad_u8(a.z, b.z, s);
    s = __builtin_amdgcn_sad_u8(a.w, b.w, s);
    return s;
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// LDS accesses by absolute byte address: keeps the per-candidate arithmetic at one v_sub / v_add
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ uint4 lds_read16(uint32_t addr) {
    const u32x4_t q = *(__attribute__((address_space(3))) const u32x4_t*)(uintptr_t)addr;
    return make_uint4(q.x, q.y, q.z, q.w);
}
__device__ __forceinline__ uint2 lds_read8(uint32_t addr) {
    const u32x2_t q = *(__attribute__((address_space(3))) const u32x2_t*)(uintptr_t)addr;
    return make_uint2(q.x, q.y);
}
__device__ __forceinline__ uint32_t lds_read4(uint32_t addr) {
    return *(__attribute__((address_space(3))) const uint32_t*)(uintptr_t)addr;
}
__device__ __forceinline__ uint32_t lds_read2(uint32_t addr) {
    return *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)addr;
}
// (SAD << 16) accumulated on top of k: the chain that builds a  cost << 16 | rank  key in place (v_sad_hi_u8)
__device__ __forceinline__ int sad_hi16(const uint4& a, const uint4& b, uint32_t k) {
    k = __builtin_amdgcn_sad_hi_u8(a.x, b.x, k);
    k = __builtin_amdgcn_sad_hi_u8(a.y, b.y, k);
    k = __builtin_amdgcn_sad_hi_u8(a.z, b.z, k);
    k = __builtin_amdgcn_sad_hi_u8(a.w, b.w, k);
    return (int)k;
}

// best / second best update: best2 = median(best1, best2, key), best1 = min(best1, key)
// (the strict-"<" chain of elas.cpp:419-427 on keys E<<16|d, two VALU ops)
__device__ __forceinline__ void keep_two(uint32_t key, uint32_t& best1, uint32_t& best2) {
    uint32_t m;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best1), "v"(best2), "v"(key));
    best2 = m;
    best1 = key < best1 ? key : best1;
}

// descriptor texture: sum |byte - 128| (elas.cpp:358-362, 851-855)
__device__ __forceinline__ uint32_t texture16(const uint4& a) {
    const uint4 mid = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    return sad16(a, mid);
}

__device__ __forceinline__ int32_t sat_u8(int32_t x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

// (uint32_t)f assigned to int32_t with x86 cvttss2si semantics (elas.cpp:1081-1082)
__device__ __forceinline__ int32_t f2u2i(float f) { return (int32_t)(uint32_t)(long long)f; }

// which (pair, side) slot a packed triangle index belongs to
__device__ __forceinline__ int tri_slot(const GroupHdr* __restrict__ h, int T, int* first) {
    const int slots = 2 * h->npairs;
    int s = 0, lo = 0;
    while (s < slots - 1 && T >= h->tri_end[s]) {
        lo = h->tri_end[s];
        s++;
    }
    *first = lo;
    return s;
}

// Global -> LDS staging of n 16-byte slots by a whole block.  A plain "dst[i] = src(i)" loop compiles to
// load, wait, store, next load: every pass over the block pays a full memory round trip.  Here kUn loads
// of a thread are in flight before the first is stored.
template <int kUn, typename F>
__device__ __forceinline__ void stage_slots(uint4* dst, int n, int tid, int nthreads, F src_of) {
    for (int i0 = tid; i0 < n; i0 += kUn * nthreads) {
        uint4 v[kUn];
#pragma unroll
        for (int k = 0; k < kUn; k++) {
            const int i = i0 + k * nthreads;
            if (i < n) v[k] = src_of(i);
        }
#pragma unroll
        for (int k = 0; k < kUn; k++) {
            const int i = i0 + k * nthreads;
            if (i < n) dst[i] = v[k];
        }
    }
}

// ---------------------------------------------------------------------------
// E1+E2  3x3 Sobel + 16-byte descriptor, fused, streaming: no LDS, no barriers.
//   filter::sobel3x3          libelas/src/filter.cpp:408-416 (+372-405, 227-267, 176-222)
//   Descriptor::createDescriptor   libelas/src/descriptor.cpp:48-121
// Border descriptors (and odd rows when half) are written as zero.  A wave owns a strip of 58 image
// columns (lane l <-> column strip*58 + l - 3, three halo lanes on each side) and walks down a
// segment of rows.  Per row every lane loads ONE image byte; the 3x3 Sobel needs the last three
// bytes of the lane (registers) and the column sums of the two neighbour lanes (DPP wave shifts);
// the 16 descriptor bytes of row y are assembled from per-row packed words of the last five du
// rows / three dv rows, which ride along in registers as the walk advances.  ~45 VALU operations
// per pixel instead of ~100 for the LDS tile kernel it replaced (three phases with byte gathers),
// which was 9 % of all VALU work of a pair.
// DS_ROWS rows are written per wave (the walk takes DS_ROWS + 6 steps); DS_CHUNK image bytes (one
// per row) are loaded ahead of the arithmetic.
// ---------------------------------------------------------------------------
constexpr int DS_COLS = 58;    // columns written per wave (lanes 3 .. 60)

// The shifted value is made opaque so that the compiler keeps the plain v_mov_b32_dpp: folded into
// a VOP2 operand (v_subrev_u32_dpp ... wave_shl:1) the shift came out wrong on gfx950 (the
// difference prev(S) - next(S) evaluated to 0 everywhere).
__device__ __forceinline__ int lane_prev(int v) {   // lane i <- lane i-1 (lane 0: 0)
    int r = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);   // wave_shr:1
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ int lane_next(int v) {   // lane i <- lane i+1 (lane 63: 0)
    int r = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true);   // wave_shl:1
    asm volatile("" : "+v"(r));
    return r;
}

template <int DS_ROWS, int DS_CHUNK>
__global__ __launch_bounds__(256) void k_descriptor_stream(DevImages img, int W, int H, int half, int strips,
                                                           int nwaves, uint8_t* __restrict__ desc_all) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wid >= nwaves) return;
    const int seg = wid / strips, strip = wid - seg * strips;
    const int pair = blockIdx.y >> 1, im = blockIdx.y & 1;
    const int x = strip * DS_COLS + lane - 3;
    const int ys = seg * DS_ROWS;
    const bool x_in = x >= 0 && x < W;
    const bool x_out = lane >= 3 && lane < 3 + DS_COLS && x < W;
    const bool x_inside = x >= 3 && x < W - 3;
    const uint8_t* __restrict__ src = img.I[im] + (size_t)pair * img.stride + (x_in ? x : 0);
    const int pitch = img.pitch;
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(desc_all + (size_t)blockIdx.y * W * H * 16);
    const uint32_t xo = x_out ? (uint32_t)x : 0u;

    int i1 = 0, i2 = 0;                          // image bytes of rows r-1, r-2
    uint32_t ucar = 0, vcar = 0;                 // own du / dv of the last four centre rows, newest in byte 0
    uint32_t w1 = 0, w2 = 0, w3 = 0;             // du[x-2] | du[x] << 8 | du[x+2] << 16 of rows c-1, c-2, c-3
    uint32_t y1 = 0, y2 = 0;                     // du[x-1] | du[x] << 8 | du[x] << 16 | du[x+1] << 24 of rows c-1, c-2
    uint32_t z1 = 0, z2 = 0;                     // dv[x-1] << 8 | dv[x+1] << 16 of rows c-1, c-2
    for (int r0 = ys - 3; r0 < ys + DS_ROWS + 3; r0 += DS_CHUNK) {
        if (r0 - 3 >= H) break;                  // nothing below the image is written
        int px[DS_CHUNK];
#pragma unroll
        for (int k = 0; k < DS_CHUNK; k++) {
            const int r = r0 + k;
            px[k] = (x_in && r >= 0 && r < H) ? (int)src[(size_t)r * pitch] : 0;
        }
#pragma unroll
        for (int k = 0; k < DS_CHUNK; k++) {
            const int r = r0 + k, i0 = px[k];
            // centre row c = r-1: vertical 1 2 1 and 1 0 -1 of this column, then across the lanes
            const int S = i2 + 2 * i1 + i0, T = i2 - i0;
            const int u = sat_u8(((lane_prev(S) - lane_next(S)) >> 2) + 128);
            const int v = sat_u8(((lane_prev(T) + 2 * T + lane_next(T)) >> 2) + 128);
            i2 = i1;
            i1 = i0;
            const uint32_t uL1 = (uint32_t)lane_prev(u), uR1 = (uint32_t)lane_next(u);
            const uint32_t uL2 = (uint32_t)lane_prev((int)uL1), uR2 = (uint32_t)lane_next((int)uR1);
            const uint32_t Wc = uL2 | ((uint32_t)u << 8) | (uR2 << 16);
            const uint32_t Yc = uL1 | ((uint32_t)u * 0x00010100u) | (uR1 << 24);
            const uint32_t Zc = ((uint32_t)lane_prev(v) << 8) | ((uint32_t)lane_next(v) << 16);
            const uint32_t u_c4 = ucar >> 24;    // du of row c-4
            ucar = (ucar << 8) | (uint32_t)u;
            vcar = (vcar << 8) | (uint32_t)v;    // bytes: c, c-1, c-2, c-3
            // descriptor of row y = c-2 (descriptor.cpp:88-117)
            const int y = r - 3;
            if (y >= ys && y < ys + DS_ROWS && y < H) {
                bool inside = x_inside && y >= 3 && y < H - 3;
                if (half) inside = inside && y >= 4 && (y & 1) == 0;
                uint4 out;
                out.x = u_c4 | (w3 << 8);
                out.y = y2;
                out.z = w1 | ((uint32_t)u << 24);
                out.w = (vcar >> 24) | z2 | ((vcar << 16) & 0xFF000000u);
                if (!inside) out = make_uint4(0, 0, 0, 0);
                if (x_out) {
                    // streamed out, read again only after the whole group's descriptors are written
                    uint4* q = &dst[(uint32_t)(y * W) + xo];
                    __builtin_nontemporal_store(out.x, &q->x);
                    __builtin_nontemporal_store(out.y, &q->y);
                    __builtin_nontemporal_store(out.z, &q->z);
                    __builtin_nontemporal_store(out.w, &q->w);
                }
            }
            w3 = w2; w2 = w1; w1 = Wc;
            y2 = y1; y1 = Yc;
            z2 = z1; z1 = Zc;
        }
    }
}

// ---------------------------------------------------------------------------
// Descriptors on the fly (round 4).  k_descriptor_stream writes 32 N bytes per pair (16 per pixel and image) that
// the support and dense matchers read back once: its duration is that write.  In this form E1 only stores the
// two Sobel planes (du, dv: 1 byte per pixel each, k_sobel_planes) and the matchers assemble the descriptor rows
// they stage in LDS themselves (fly_desc4: 16 aligned 32-bit loads and ~32 v_perm for four adjacent pixels).
// The planes live at the start of the pair's descriptor buffer: slot z = 2 pair + image holds du at +0 and dv at
// + H * pitch, pitch = roundup(W, 4) + 16, column x at byte 8 + x (so the words x-4 .. x+7 of a row always exist).
// ---------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32_unaligned_t __attribute__((aligned(1)));
__device__ __forceinline__ s16x2 dp_pair01(uint32_t w) { return __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, w, 0x0c010c00u)); }
__device__ __forceinline__ s16x2 dp_pair23(uint32_t w) { return __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, w, 0x0c030c02u)); }
// (a.hi, b.lo): the pair one column to the right of a, b being the next pair
__device__ __forceinline__ s16x2 dp_mid(s16x2 a, s16x2 b) {
    return __builtin_bit_cast(s16x2, __builtin_amdgcn_alignbyte(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 2u));
}
__device__ __forceinline__ s16x2 dp_prev(s16x2 v) { return __builtin_bit_cast(s16x2, lane_prev(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ s16x2 dp_next(s16x2 v) { return __builtin_bit_cast(s16x2, lane_next(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ s16x2 dp_sobel_out(s16x2 v) {   // sat_u8((v >> 2) + 128)
    const s16x2 lo = {0, 0}, hi = {255, 255}, off = {128, 128};
    return __builtin_elementwise_min(__builtin_elementwise_max((v >> 2) + off, lo), hi);
}
__device__ __forceinline__ uint32_t dp_bytes(s16x2 lo, s16x2 hi) {   // low bytes of four 16-bit values -> one word
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
}
// v_perm_b32 selector bytes: 0..3 = bytes of `lo` (second operand), 4..7 = bytes of `hi` (first operand)
#define DP_SEL(b0, b1, b2, b3) ((uint32_t)(b0) | (uint32_t)(b1) << 8 | (uint32_t)(b2) << 16 | (uint32_t)(b3) << 24)

__host__ __device__ __forceinline__ int fly_pitch(int W) { return ((W + 3) & ~3) + 16; }

// E1 alone: filter::sobel3x3 (filter.cpp:408-416) of both images of a group into the two planes.  A lane owns the
// four pixels of one image word (packed 16-bit column sums, neighbour pairs by DPP), a wave walks SP_ROWS rows.
constexpr int SP_ROWS = 30;
__global__ __launch_bounds__(256) void k_sobel_planes(DevImages img, int W, int H, int strips, int cols, int nwaves,
                                                      uint8_t* __restrict__ desc_all) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wid >= nwaves) return;
    const int seg = wid / strips, strip = wid - seg * strips;
    const int pair = blockIdx.y >> 1, im = blockIdx.y & 1;
    const int x = strip * cols + 4 * (lane - 1);        // first of this lane's four columns
    const int ys = seg * SP_ROWS;
    const int pitch = fly_pitch(W);
    const bool writes = lane >= 1 && 4 * (lane - 1) < cols && x < W;
    // one unconditional 32-bit load per lane and row from an address inside the row (see k_descriptor_pk's note:
    // the last word of a row is read at W-4 and shifted down, lanes beside the image are masked to zero)
    const bool in_row = x >= 0 && x < W;
    const int over = in_row && x + 3 >= W ? x + 4 - W : 0;
    const uint32_t lane_mask = in_row ? 0xFFFFFFFFu : 0u, lane_shift = 8u * (uint32_t)over;
    const uint8_t* __restrict__ src = img.I[im] + (size_t)pair * img.stride + (in_row ? x - over : 0);
    uint8_t* du = desc_all + (size_t)blockIdx.y * W * H * 16 + 8 + (writes ? x : 0);
    uint8_t* dv = du + (size_t)H * pitch;
    s16x2 a1[2] = {{0, 0}, {0, 0}}, a2[2] = {{0, 0}, {0, 0}};     // image pairs of rows r-1, r-2
    constexpr int kChunk = 16;
    static_assert((SP_ROWS + 2) % kChunk == 0, "the walk is a whole number of load chunks");
    for (int r0 = ys - 1; r0 <= ys + SP_ROWS; r0 += kChunk) {   // centre row c = r - 1 runs over ys-2 .. ys+SP_ROWS-1
        if (r0 - 1 >= H) break;
        uint32_t px[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; k++) {       // the chunk's loads go out before its arithmetic
            const int r = r0 + k;
            const int rc = r < 0 ? 0 : (r < H ? r : H - 1);
            const uint32_t row_mask = (r >= 0 && r < H) ? lane_mask : 0u;
            px[k] = (*reinterpret_cast<const u32_unaligned_t*>(src + (size_t)rc * img.pitch) >> lane_shift) & row_mask;
        }
#pragma unroll
        for (int k = 0; k < kChunk; k++) {
            const int r = r0 + k;
            const s16x2 p0 = dp_pair01(px[k]), p1 = dp_pair23(px[k]);
            const s16x2 two = {2, 2};
            const s16x2 S0 = a2[0] + two * a1[0] + p0, S1 = a2[1] + two * a1[1] + p1;
            const s16x2 T0 = a2[0] - p0, T1 = a2[1] - p1;
            a2[0] = a1[0]; a2[1] = a1[1];
            a1[0] = p0; a1[1] = p1;
            const s16x2 SL = dp_prev(S1), SR = dp_next(S0), TL = dp_prev(T1), TR = dp_next(T0);
            const s16x2 Sm0 = dp_mid(SL, S0), Sm1 = dp_mid(S0, S1), Sm2 = dp_mid(S1, SR);
            const s16x2 Tm0 = dp_mid(TL, T0), Tm1 = dp_mid(T0, T1), Tm2 = dp_mid(T1, TR);
            const uint32_t Dw = dp_bytes(dp_sobel_out(Sm0 - Sm1), dp_sobel_out(Sm1 - Sm2));
            const uint32_t Vw = dp_bytes(dp_sobel_out(Tm0 + two * T0 + Tm1), dp_sobel_out(Tm1 + two * T1 + Tm2));
            const int c = r - 1;
            if (writes && c >= ys && c < ys + SP_ROWS && c < H) {
                *reinterpret_cast<uint32_t*>(du + (size_t)c * pitch) = Dw;
                *reinterpret_cast<uint32_t*>(dv + (size_t)c * pitch) = Vw;
            }
        }
    }
}

// The descriptors (descriptor.cpp:88-117) of the four pixels x .. x+3 (x a multiple of 4, x < W) of row `line`, from
// the planes of one image.  Zero outside columns 3 .. W-4 / rows 3 .. H-4, like the descriptor kernel writes them.
__device__ __forceinline__ void fly_desc4(const uint8_t* __restrict__ du, int W, int H, int x, int line, uint4 out[4]) {
    const int pitch = fly_pitch(W);
    if (line < 3 || line >= H - 3) {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    const uint8_t* dv = du + (size_t)H * pitch;
    const uint8_t* pu = du + (size_t)line * pitch + 8 + x;
    const uint8_t* pv = dv + (size_t)line * pitch + 8 + x;
    auto ld = [](const uint8_t* q) { return *reinterpret_cast<const uint32_t*>(q); };
    const uint32_t Um2 = ld(pu - 2 * pitch), Up2 = ld(pu + 2 * pitch);
    const uint32_t UaL = ld(pu - pitch - 4), Ua = ld(pu - pitch), UaR = ld(pu - pitch + 4);     // row line-1
    const uint32_t UcL = ld(pu - 4), Uc = ld(pu), UcR = ld(pu + 4);                             // row line
    const uint32_t UbL = ld(pu + pitch - 4), Ub = ld(pu + pitch), UbR = ld(pu + pitch + 4);     // row line+1
    const uint32_t Vm1 = ld(pv - pitch), Vp1 = ld(pv + pitch);
    const uint32_t VcL = ld(pv - 4), Vc = ld(pv), VcR = ld(pv + 4);
    uint32_t Wa[4], Wb[4], Y[4], Z[4];
    // du[x-2] | du[x] << 8 | du[x+2] << 16 of rows line-1 (Wa) and line+1 (Wb)
    Wa[0] = __builtin_amdgcn_perm(Ua, UaL, DP_SEL(2, 4, 6, 0x0c));  Wb[0] = __builtin_amdgcn_perm(Ub, UbL, DP_SEL(2, 4, 6, 0x0c));
    Wa[1] = __builtin_amdgcn_perm(Ua, UaL, DP_SEL(3, 5, 7, 0x0c));  Wb[1] = __builtin_amdgcn_perm(Ub, UbL, DP_SEL(3, 5, 7, 0x0c));
    Wa[2] = __builtin_amdgcn_perm(UaR, Ua, DP_SEL(0, 2, 4, 0x0c));  Wb[2] = __builtin_amdgcn_perm(UbR, Ub, DP_SEL(0, 2, 4, 0x0c));
    Wa[3] = __builtin_amdgcn_perm(UaR, Ua, DP_SEL(1, 3, 5, 0x0c));  Wb[3] = __builtin_amdgcn_perm(UbR, Ub, DP_SEL(1, 3, 5, 0x0c));
    // du[x-1] | du[x] << 8 | du[x] << 16 | du[x+1] << 24 of row line
    Y[0] = __builtin_amdgcn_perm(Uc, UcL, DP_SEL(3, 4, 4, 5));
    Y[1] = __builtin_amdgcn_perm(Uc, Uc, DP_SEL(0, 1, 1, 2));
    Y[2] = __builtin_amdgcn_perm(Uc, Uc, DP_SEL(1, 2, 2, 3));
    Y[3] = __builtin_amdgcn_perm(UcR, Uc, DP_SEL(2, 3, 3, 4));
    // dv[x-1] << 8 | dv[x+1] << 16 of row line
    Z[0] = __builtin_amdgcn_perm(Vc, VcL, DP_SEL(0x0c, 3, 5, 0x0c));
    Z[1] = __builtin_amdgcn_perm(Vc, Vc, DP_SEL(0x0c, 0, 2, 0x0c));
    Z[2] = __builtin_amdgcn_perm(Vc, Vc, DP_SEL(0x0c, 1, 3, 0x0c));
    Z[3] = __builtin_amdgcn_perm(VcR, Vc, DP_SEL(0x0c, 2, 4, 0x0c));
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint4 o;
        o.x = __builtin_amdgcn_perm(Wa[i], Um2, DP_SEL(i, 4, 5, 6));          // du(line-2, x) | W(line-1) << 8
        o.y = Y[i];
        o.z = __builtin_amdgcn_perm(Up2, Wb[i], DP_SEL(0, 1, 2, 4 + i));      // W(line+1) | du(line+2, x) << 24
        const uint32_t t = __builtin_amdgcn_perm(Z[i], Vm1, DP_SEL(i, 5, 6, 0x0c));   // dv(line-1, x) | Z(line)
        o.w = __builtin_amdgcn_perm(Vp1, t, DP_SEL(0, 1, 2, 4 + i));          // ... | dv(line+1, x) << 24
        const bool in = x + i >= 3 && x + i < W - 3;
        out[i] = in ? o : make_uint4(0, 0, 0, 0);
    }
}

// The same four descriptors for k_match_list's staging loop (round 6): `base` is wave-uniform and `off` the 32-bit byte
// offset of du(line, x) from it, so that every load is  scalar base + 32-bit lane offset + constant  (the pointer form
// above costs ten 64-bit lane additions per task); the row must be an interior one (3 <= line < H - 3) and the masking
// of the columns outside 3 .. W-4 is the caller's (it concerns the first task of a row and the last two).  `vplane` =
// byte distance of the dv plane from the du plane.
__device__ __forceinline__ void fly_desc4_off(const uint8_t* __restrict__ base, uint32_t off, uint32_t pitch,
                                              uint32_t vplane, uint4 out[4]) {
    // address = uniform base + zero-extended 32-bit lane offset + constant: the form of a global load with a scalar base
    auto ld = [base](uint32_t o, int c) { return *reinterpret_cast<const uint32_t*>(base + (size_t)o + c); };
    const uint32_t oa = off - pitch, ob = off + pitch, ov = off + vplane;
    const uint32_t Um2 = ld(oa - pitch, 0), Up2 = ld(ob + pitch, 0);
    const uint32_t UaL = ld(oa, -4), Ua = ld(oa, 0), UaR = ld(oa, 4);     // row line-1
    const uint32_t UcL = ld(off, -4), Uc = ld(off, 0), UcR = ld(off, 4);  // row line
    const uint32_t UbL = ld(ob, -4), Ub = ld(ob, 0), UbR = ld(ob, 4);     // row line+1
    const uint32_t Vm1 = ld(ov - pitch, 0), Vp1 = ld(ov + pitch, 0);
    const uint32_t VcL = ld(ov, -4), Vc = ld(ov, 0), VcR = ld(ov, 4);
    uint32_t Wa[4], Wb[4], Y[4], Z[4];
    Wa[0] = __builtin_amdgcn_perm(Ua, UaL, DP_SEL(2, 4, 6, 0x0c));  Wb[0] = __builtin_amdgcn_perm(Ub, UbL, DP_SEL(2, 4, 6, 0x0c));
    Wa[1] = __builtin_amdgcn_perm(Ua, UaL, DP_SEL(3, 5, 7, 0x0c));  Wb[1] = __builtin_amdgcn_perm(Ub, UbL, DP_SEL(3, 5, 7, 0x0c));
    Wa[2] = __builtin_amdgcn_perm(UaR, Ua, DP_SEL(0, 2, 4, 0x0c));  Wb[2] = __builtin_amdgcn_perm(UbR, Ub, DP_SEL(0, 2, 4, 0x0c));
    Wa[3] = __builtin_amdgcn_perm(UaR, Ua, DP_SEL(1, 3, 5, 0x0c));  Wb[3] = __builtin_amdgcn_perm(UbR, Ub, DP_SEL(1, 3, 5, 0x0c));
    Y[0] = __builtin_amdgcn_perm(Uc, UcL, DP_SEL(3, 4, 4, 5));
    Y[1] = __builtin_amdgcn_perm(Uc, Uc, DP_SEL(0, 1, 1, 2));
    Y[2] = __builtin_amdgcn_perm(Uc, Uc, DP_SEL(1, 2, 2, 3));
    Y[3] = __builtin_amdgcn_perm(UcR, Uc, DP_SEL(2, 3, 3, 4));
    Z[0] = __builtin_amdgcn_perm(Vc, VcL, DP_SEL(0x0c, 3, 5, 0x0c));
    Z[1] = __builtin_amdgcn_perm(Vc, Vc, DP_SEL(0x0c, 0, 2, 0x0c));
    Z[2] = __builtin_amdgcn_perm(Vc, Vc, DP_SEL(0x0c, 1, 3, 0x0c));
    Z[3] = __builtin_amdgcn_perm(VcR, Vc, DP_SEL(0x0c, 2, 4, 0x0c));
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[i].x = __builtin_amdgcn_perm(Wa[i], Um2, DP_SEL(i, 4, 5, 6));
        out[i].y = Y[i];
        out[i].z = __builtin_amdgcn_perm(Up2, Wb[i], DP_SEL(0, 1, 2, 4 + i));
        const uint32_t t = __builtin_amdgcn_perm(Z[i], Vm1, DP_SEL(i, 5, 6, 0x0c));
        out[i].w = __builtin_amdgcn_perm(Vp1, t, DP_SEL(0, 1, 2, 4 + i));
    }
}

// ---------------------------------------------------------------------------
// E3+E4  support candidate matching
//   Elas::computeMatchingDisparity   libelas/src/elas.cpp:322-445
//   Elas::computeSupportMatches      libelas/src/elas.cpp:449-493
// One wave per lattice candidate (blockIdx.y = pair); lanes = disparities (up to
// 4 rounds of 64).  Energy = SAD over the 4 descriptors at (+-2,+-2) (64 bytes).
// The reference keeps best and second best while scanning d upwards with strict
// "<": that is the smallest and second smallest of the keys (E<<16 | d).
// ---------------------------------------------------------------------------
struct SupportParams {
    int W, H, Wc, Hc, step, npairs;
    int disp_min, disp_max, support_texture, lr_threshold;
    float support_threshold;
};

__device__ __forceinline__ int support_match(const uint8_t* __restrict__ own,
                                             const uint8_t* __restrict__ oth, int u, int v,
                                             bool right, const SupportParams& P, int lane) {
    if (!(u >= 5 && u <= P.W - 6 && v >= 5 && v <= P.H - 6)) return -1;
    const size_t row = (size_t)P.W;  // in 16-byte units
    const uint4* ownq = reinterpret_cast<const uint4*>(own);
    const uint4* othq = reinterpret_cast<const uint4*>(oth);
    const size_t centre = (size_t)v * row + u;
    if ((int)texture16(ownq[centre]) < P.support_texture) return -1;

    const int dmin = P.disp_min > 0 ? P.disp_min : 0;
    int dmax = right ? P.W - u - 5 : u - 5;
    dmax = dmax < P.disp_max ? dmax : P.disp_max;
    if (dmax - dmin < 10) return -1;

    const size_t o0 = centre - 2 * row - 2, o1 = centre - 2 * row + 2;
    const size_t o2 = centre + 2 * row - 2, o3 = centre + 2 * row + 2;
    const uint4 r0 = ownq[o0], r1 = ownq[o1], r2 = ownq[o2], r3 = ownq[o3];

    uint32_t best1 = 0xFFFFFFFFu, best2 = 0xFFFFFFFFu;
    for (int d0 = dmin; d0 <= dmax; d0 += kWave) {
        const int d = d0 + lane;
        if (d <= dmax) {
            const ptrdiff_t sh = right ? d : -d;
            uint32_t e = sad16(r0, othq[o0 + sh]);
            e = sad16_acc(r1, othq[o1 + sh], e);
            e = sad16_acc(r2, othq[o2 + sh], e);
            e = sad16_acc(r3, othq[o3 + sh], e);
            keep_two((e << 16) | (uint32_t)d, best1, best2);
        }
    }
    const uint32_t m1 = wave_min_u32(best1);
    const uint32_t m2 = wave_min_u32(best1 == m1 ? best2 : best1);
    if (m1 == 0xFFFFFFFFu || m2 == 0xFFFFFFFFu) return -1;
    const float e1 = (float)(m1 >> 16), e2 = (float)(m2 >> 16);
    if (e1 < __fmul_rn(P.support_threshold, e2)) return (int)(m1 & 0xFFFFu);
    return -1;
}

__global__ __launch_bounds__(256) void k_support(const uint8_t* __restrict__ desc_all,
                                                 int16_t* __restrict__ dcan_all, SupportParams P) {
    const int cand = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    const int lane = threadIdx.x & 63;
    if (cand >= P.Wc * P.Hc) return;
    const int pair = blockIdx.y;
    const size_t dsz = (size_t)P.W * P.H * 16;
    const uint8_t* desc1 = desc_all + (size_t)(2 * pair) * dsz;
    const uint8_t* desc2 = desc1 + dsz;
    const int vc = cand / P.Wc, uc = cand - vc * P.Wc;
    int out = 0;  // row 0 / column 0 stay at calloc's 0 (elas.cpp:464, 471-479)
    if (uc > 0 && vc > 0) {
        out = -1;
        const int u = uc * P.step, v = vc * P.step;
        const int d = support_match(desc1, desc2, u, v, false, P, lane);
        if (d >= 0) {
            const int d2 = support_match(desc2, desc1, u - d, v, true, P, lane);
            const int diff = d > d2 ? d - d2 : d2 - d;
            if (d2 >= 0 && diff <= P.lr_threshold) out = d;
        }
    }
    if (lane == 0) dcan_all[(size_t)pair * P.Wc * P.Hc + cand] = (int16_t)out;
}

// LDS-staged variant.  The global version above re-reads every descriptor row
// segment once per candidate (~600 MB of L2 traffic per KITTI pair: L2-bound).
// Here one block owns kSB consecutive candidates of one lattice row and stages
// the two descriptor rows v-2 / v+2 of both images, over the column span any
// of its candidates can touch, in LDS once (~35 KB for disp_max 255); all SAD
// operands then come from LDS as conflict-free ds_read_b128 (consecutive lanes
// = consecutive disparities = consecutive 16-byte slots).
// candidates per block (kSB) and threads per block (kST) are template parameters

// stride of a strip row in slots: the width rounded up to 1 (mod 4)
__host__ __device__ __forceinline__ int support_stride(int w) { return ((w + 2) & ~3) + 1; }
struct StripView {
    const uint4* base;   // LDS, two rows at a stride of `w` slots: row 0 = v-2, row 1 = v+2
    int x0, w;
    __device__ __forceinline__ uint4 at(int row, int x) const { return base[row * w + (x - x0)]; }
};

// Four candidates per wave: lanes 16g .. 16g+15 search candidate g, 16 disparities per trip.
// The inner loop is one 64-byte SAD per lane and trip, as with one candidate per wave; what is shared by the four candidates is everything around it -- range set-up, the two
// minimum reductions (4 DPP steps inside a row of 16 instead of 6 across the wave + readlane)
// and the ratio test -- which was ~40 % of the instructions of a search.  A quarter wave reads 16
// consecutive 16-byte slots, so the ds_read_b128 stay conflict-free.  `act`: this lane's
// candidate takes part; u is per lane (uniform inside a row of 16).  Returns d or -1 per lane.
__device__ __forceinline__ int support_match_rows(const StripView& own, const StripView& oth, const int centre_texture,
                                                  int u, bool right, bool act, const SupportParams& P, int gl) {
    act = act && u >= 5 && u <= P.W - 6 && centre_texture >= P.support_texture;
    const int dmin = P.disp_min > 0 ? P.disp_min : 0;
    int dmax = right ? P.W - u - 5 : u - 5;
    dmax = dmax < P.disp_max ? dmax : P.disp_max;
    act = act && dmax - dmin >= 10;
    if (__builtin_amdgcn_ballot_w64(act) == 0) return -1;
    const int us = act ? u : own.x0 + 2;          // idle rows read a slot that exists
    const uint4 r0 = own.at(0, us - 2), r1 = own.at(0, us + 2);
    const uint4 r2 = own.at(1, us - 2), r3 = own.at(1, us + 2);
    // the trip count of the wave is that of its widest row
    const int dm = act ? dmax : -1;
    int top = __builtin_amdgcn_readlane(dm, 0);
    const int t1 = __builtin_amdgcn_readlane(dm, 16), t2 = __builtin_amdgcn_readlane(dm, 32),
              t3 = __builtin_amdgcn_readlane(dm, 48);
    top = top > t1 ? top : t1;
    top = top > t2 ? top : t2;
    top = top > t3 ? top : t3;
    // Which of the 16 disparities of a trip a lane takes is chosen for the LDS: lane gl always
    // reads a slot congruent to gl (mod 16, up to a constant that is the same for every row of 16
    // lanes), whatever column its candidate starts from -- so the 16 lanes of a ds_read_b128 phase,
    // which come from two different rows (see k_support_lds), always cover the 16 slots of the
    // bank array once.  The forward search starts at the lattice column, the backward search at
    // u - d of each candidate: without the rotation its alignment was data dependent.
    const int so = (act ? u : 0) - oth.x0;   // slot of disparity 0 (the +-2 offsets are constants)
    const int off = right ? (gl - so - dmin) & 15 : (so - gl - dmin) & 15;
    uint32_t best1 = 0xFFFFFFFFu, best2 = 0xFFFFFFFFu;
    // Trip t evaluates d = dmin + 16 t + off.  The trips are taken in the order in which the LDS addresses RISE
    // (forward search: slot u - d, so t descends; backward: slot u + d, t ascends): four consecutive trips then
    // read at constant offsets k * 256 from two address registers (rows v-2 and v+2), and the trips in which every
    // lane of the wave is inside its range run without the per-trip range test (round 4: 24 -> ~20 operations per
    // trip).  Keys E << 16 | d make the minimum independent of the order.
    const int T = (top - dmin) / 16 + 1;                                   // trips of the wave (uniform)
    const int tl = act ? (dm - dmin - off >= 0 ? (dm - dmin - off) >> 4 : -1) : T - 1;   // last trip this lane is inside its range
    // j = position in evaluation order: t = right ? j : T - 1 - j.  A lane is valid for j <= tl (right) / j >= T-1-tl
    const int jl = right ? tl : T - 1 - tl;
    // uniform split: right: j in [0, jmin] unmasked, then (jmin, T) masked; forward: j in [0, jmax) masked, then [jmax, T)
    const uint32_t red = wave_min_u32(right ? (uint32_t)(jl + 1) : (uint32_t)(T - jl));   // both >= 0
    const int jsplit = right ? (int)red - 1 : T - (int)red;             // right: jmin; forward: jmax
    const uint32_t oth_addr = lds_addr_of(oth.base);
    const int d_first = right ? dmin + off : dmin + 16 * (T - 1) + off;   // disparity of j = 0
    const int uw_first = right ? (act ? u : oth.x0 + 2) + d_first : (act ? u - d_first : oth.x0 + 2);
    uint32_t a0 = oth_addr + (uint32_t)(uw_first - 2 - oth.x0) * 16u;      // (row 0, uw - 2) of j = 0; + 256 per trip
    const uint32_t row1 = (uint32_t)oth.w * 16u;
    int d = d_first;
    // Round 6: the four slot reads of a trip are REQUESTED together (hipcc's scheduler, left alone, keeps one
    // four-register buffer and waits for every read before its four SADs -- four exposed LDS round trips per trip), and in
    // the unrolled loop the next trip's four are requested before the current trip's SADs start (two register sets,
    // alternating by name: no copies).  The compiler barrier pins the requests where they are written.
    struct Slots { uint4 a, b, c, e; };
    auto fetch = [&](uint32_t addr0, uint32_t addr1) {
        Slots q;
        q.a = lds_read16(addr0); q.b = lds_read16(addr0 + 64u); q.c = lds_read16(addr1); q.e = lds_read16(addr1 + 64u);
        return q;
    };
    auto eval = [&](const Slots& q, int dd) {
        uint32_t key = (uint32_t)sad_hi16(r0, q.a, (uint32_t)dd);
        key = (uint32_t)sad_hi16(r1, q.b, key);
        key = (uint32_t)sad_hi16(r2, q.c, key);
        key = (uint32_t)sad_hi16(r3, q.e, key);
        keep_two(key, best1, best2);
    };
    auto trip = [&](uint32_t addr0, uint32_t addr1, int dd) {
        const Slots q = fetch(addr0, addr1);
        asm volatile("" ::: "memory");
        eval(q, dd);
    };
    const int dstep = right ? 16 : -16;
    int j = 0;
    if (!right) {   // masked head: trips some lanes are outside of
        for (; j < jsplit; j++, a0 += 256u, d += dstep)
            if (j >= jl) trip(a0, a0 + row1, d);
    }
    const int jend = right ? jsplit + 1 : T;
    if (j + 4 <= jend) {
        Slots qa = fetch(a0, a0 + row1), qb;
        for (; j + 4 <= jend; j += 4, a0 += 1024u, d += 4 * dstep) {
            const uint32_t a1 = a0 + row1;
            qb = fetch(a0 + 256u, a1 + 256u);
            asm volatile("" ::: "memory");
            eval(qa, d);
            qa = fetch(a0 + 512u, a1 + 512u);
            asm volatile("" ::: "memory");
            eval(qb, d + dstep);
            qb = fetch(a0 + 768u, a1 + 768u);
            asm volatile("" ::: "memory");
            eval(qa, d + 2 * dstep);
            // (the next turn's first trip; past the last turn it reads slots of this block's strips or the padding
            // behind them and is dropped)
            qa = fetch(a0 + 1024u, a1 + 1024u);
            asm volatile("" ::: "memory");
            eval(qb, d + 3 * dstep);
        }
    }
    for (; j < jend; j++, a0 += 256u, d += dstep) trip(a0, a0 + row1, d);
    if (right) {    // masked tail
        for (; j < T; j++, a0 += 256u, d += dstep)
            if (j <= jl) trip(a0, a0 + row1, d);
    }
    const uint32_t m1 = row_min_u32(best1);
    const uint32_t m2 = row_min_u32(best1 == m1 ? best2 : best1);
    const float e1 = (float)(m1 >> 16), e2 = (float)(m2 >> 16);
    const bool good = m1 != 0xFFFFFFFFu && m2 != 0xFFFFFFFFu && e1 < __fmul_rn(P.support_threshold, e2);
    // (rows that take no part -- no texture, range too short -- ran the unmasked trips on a dummy slot: no result)
    return act && good ? (int)(m1 & 0xFFFFu) : -1;
}

// one descriptor from the Sobel planes (fly form; byte loads)
__device__ __forceinline__ uint4 fly_desc1(const uint8_t* __restrict__ du, int W, int H, int x, int line) {
    if (line < 3 || line >= H - 3 || x < 3 || x >= W - 3) return make_uint4(0, 0, 0, 0);
    const int pitch = fly_pitch(W);
    const uint8_t* pu = du + (size_t)line * pitch + 8 + x;
    const uint8_t* pv = pu + (size_t)H * pitch;
    auto U = [&](int dy, int dx) { return (uint32_t)pu[dy * pitch + dx]; };
    auto V = [&](int dy, int dx) { return (uint32_t)pv[dy * pitch + dx]; };
    uint4 o;
    o.x = U(-2, 0) | U(-1, -2) << 8 | U(-1, 0) << 16 | U(-1, 2) << 24;
    const uint32_t c = U(0, 0);
    o.y = U(0, -1) | c << 8 | c << 16 | U(0, 1) << 24;
    o.z = U(1, -2) | U(1, 0) << 8 | U(1, 2) << 16 | U(2, 0) << 24;
    o.w = V(-1, 0) | V(0, -1) << 8 | V(0, 1) << 16 | V(1, 0) << 24;
    return o;
}

template <int kSB, int kST, bool kFly>
__global__ __launch_bounds__(kST) void k_support_lds(const uint8_t* __restrict__ desc_all,
                                                     int16_t* __restrict__ dcan_all,
                                                     SupportParams P) {
    extern __shared__ uint4 s_strip[];
    // XCD-aware block order.  Workgroups go to the 8 XCDs round-robin by linear id; the chunks of one
    // lattice row re-read each other's strips (every strip is 2*disp_max wider than its candidates), and
    // neighbouring lattice rows (5 px apart) read 4 of their 9 Sobel lines in common.  XCD k therefore takes
    // the k-th contiguous eighth of the (pair, lattice row) list with all chunks of each row (round 2: row
    // q*8+k, which kept the chunks together and spread neighbouring rows over the eight L2s).
    const int chunks = (P.Wc + kSB - 1) / kSB;
    const int bid = blockIdx.x;
    const int per_xcd = (P.Hc * P.npairs + 7) >> 3;            // rows per XCD (the grid is 8 * per_xcd * chunks)
    const int row_id = (bid & 7) * per_xcd + (bid >> 3) / chunks;   // (pair, lattice row) pair index
    if (row_id >= P.Hc * P.npairs) return;
    const int pair = row_id / P.Hc, vc = row_id - pair * P.Hc;
    const int uc0 = ((bid >> 3) % chunks) * kSB;
    const int ncand = (P.Wc - uc0) < kSB ? (P.Wc - uc0) : kSB;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    int16_t* dcan = dcan_all + (size_t)pair * P.Wc * P.Hc + (size_t)vc * P.Wc + uc0;
    const int v = vc * P.step;
    // whole strip outside the matchable rows: results are known without any data
    if (vc == 0 || v < 5 || v > P.H - 6) {
        if (threadIdx.x < ncand) dcan[threadIdx.x] = (vc == 0 || uc0 + (int)threadIdx.x == 0) ? 0 : -1;
        return;
    }
    const size_t N16 = (size_t)P.W * P.H;
    const uint4* d1 = reinterpret_cast<const uint4*>(desc_all) + (size_t)(2 * pair) * N16;
    const uint4* d2 = d1 + N16;
    const int u_lo = uc0 * P.step, u_hi = (uc0 + ncand - 1) * P.step;
    // column spans (see header comment); clipped to the image
    int xl0 = u_lo - P.disp_max - 2, xl1 = u_hi + P.disp_max + 2;
    int xr0 = u_lo - P.disp_max - 2, xr1 = u_hi + 2;
    xl0 = xl0 > 0 ? xl0 : 0;
    xr0 = xr0 > 0 ? xr0 : 0;
    xl1 = xl1 < P.W - 1 ? xl1 : P.W - 1;
    xr1 = xr1 < P.W - 1 ? xr1 : P.W - 1;
    const int wl = xl1 - xl0 + 1, wr = xr1 - xr0 + 1;
    // row strides == 1 (mod 4): the four strip rows then start at slots == 0, 1, 2, 3 (mod 4), see the staging loop
    const int wls = support_stride(wl), wrs = support_stride(wr);
    uint4* sL = s_strip;
    uint4* sR = s_strip + 2 * wls;
    __shared__ uint16_t s_texL[kFly ? kSB : 1];
    uint16_t* s_texR = reinterpret_cast<uint16_t*>(s_strip + 2 * (wls + wrs));   // (fly: wr texture values behind the strips)
    if (kFly) {
        // the four strips (rows v-2 / v+2 of both images) assembled from the Sobel planes, four aligned pixels per task
        const uint8_t* pl1 = desc_all + (size_t)(2 * pair) * N16 * 16;
        const uint32_t fpitch = (uint32_t)fly_pitch(P.W), vplane = (uint32_t)P.H * fpitch, imstride = (uint32_t)(N16 * 16);
        const int al0 = xl0 & ~3, ar0 = xr0 & ~3;
        const int nl = (xl1 - al0) / 4 + 1, nr = (xr1 - ar0) / 4 + 1;      // tasks per row of the left / right strip
        // Lane l takes strip row (l & 3) -- left v-2, left v+2, right v-2, right v+2 -- and the rows start at slots
        // == 0, 1, 2, 3 (mod 4): the 8 lanes of a ds_write_b128 group store to 8 different slots modulo 8, no bank
        // conflicts (round 4, consecutive tasks of one row on consecutive lanes: 4-way on every store).
        const int nmax = nl > nr ? nl : nr;
        for (int task = (int)threadIdx.x; task < 4 * nmax; task += kST) {
            const int row = task & 1, j = task >> 2;
            const bool rgt = (task & 2) != 0;
            if (j >= (rgt ? nr : nl)) continue;
            const int x = (rgt ? ar0 : al0) + 4 * j;
            uint4 o[4];
            // (round 6: scalar base + 32-bit lane offset, see fly_desc4_off; rows v -+ 2 are interior ones here -- v is in
            // [5, H - 6] --, the columns outside 3 .. W-4 are zeroed on the few tasks that touch them)
            fly_desc4_off(pl1, (rgt ? imstride : 0u) + (uint32_t)(v + (row ? 2 : -2)) * fpitch + 8u + (uint32_t)x, fpitch,
                          vplane, o);
            if (x == 0 || x + 4 > P.W - 3) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (x + i < 3 || x + i >= P.W - 3) o[i] = make_uint4(0, 0, 0, 0);
            }
            const int x0 = rgt ? xr0 : xl0, x1 = rgt ? xr1 : xl1, w = rgt ? wrs : wls;
            uint4* dst = (rgt ? sR : sL) + row * w - x0;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (x + i >= x0 && x + i <= x1) dst[x + i] = o[i];
        }
        // descriptor texture of the centre row v: the right image over its strip (the backward search starts at any
        // column), the left image at the block's lattice columns
        for (int task = (int)threadIdx.x; task < nr + kSB; task += kST) {
            if (task < nr) {
                const int x = ar0 + 4 * task;
                uint4 o[4];
                fly_desc4_off(pl1, imstride + (uint32_t)v * fpitch + 8u + (uint32_t)x, fpitch, vplane, o);
                if (x == 0 || x + 4 > P.W - 3) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (x + i < 3 || x + i >= P.W - 3) o[i] = make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (x + i >= xr0 && x + i <= xr1) s_texR[x + i - xr0] = (uint16_t)texture16(o[i]);
            } else {
                const int c = task - nr, u = (uc0 + c) * P.step;
                s_texL[c] = (uint16_t)(c < ncand && u < P.W ? texture16(fly_desc1(pl1, P.W, P.H, u, v)) : 0u);
            }
        }
    } else {
    // (sL and sR are adjacent: one pass over both strips, up to six loads per thread in flight)
    // (the padding slots behind a row are filled with the row's last column)
    stage_slots<6>(s_strip, 2 * (wls + wrs), (int)threadIdx.x, kST, [&](int i) {
        const bool rgt = i >= 2 * wls;
        const int j = rgt ? i - 2 * wls : i, w = rgt ? wrs : wls, wreal = rgt ? wr : wl;
        const int row = j >= w, xx = j - row * w, x = xx < wreal ? xx : wreal - 1;
        return (rgt ? d2 : d1)[(size_t)(v + (row ? 2 : -2)) * P.W + (rgt ? xr0 : xl0) + x];
    });
    }
    __syncthreads();
    const StripView L = {sL, xl0, wls}, R = {sR, xr0, wrs};
    // Four candidates per wave, one per row of 16 lanes (rows 2k and 2k+1 take candidates 8 apart).
    // A ds_read_b128 is served in four phases of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31} and the same in the upper half (MI355X_MICROARCH.md, LDS) -- i.e. half of one row plus
    // half of its neighbour; support_match_rows assigns the disparities of a trip to the lanes of a
    // row so that lane position p always reads a slot congruent to p modulo 16, in the forward and
    // in the backward search alike: every phase covers the 16 slots of the bank array once.
    constexpr int kNW = kST / kWave, kPerRound = 4 * kNW;        // candidates per round: 4 per wave
    static_assert(kSB % kPerRound == 0 && kSB <= 256, "whole rounds; candidate indices fit a byte");
    const int grp = lane >> 4;
    const int gl = lane & 15;
    // forward search: every candidate of the block, kPerRound per round (4 per wave)
    __shared__ int16_t s_fwd[kSB];     // forward disparity per candidate, -1 = none
    __shared__ uint8_t s_todo[kSB];    // candidates that need the backward search, compacted
    __shared__ int s_ntodo;
#pragma unroll
    for (int rep = 0; rep < kSB / kPerRound; rep++) {
        const int c = rep * kPerRound + wave + grp * kNW;
        if (rep * kPerRound >= ncand) break;                     // (block-uniform)
        const bool have = c < ncand;
        const int uc = uc0 + (have ? c : 0), u = uc * P.step;
        const bool in = have && uc > 0 && u >= 5 && u <= P.W - 6;
        const int t1 = !in ? 0 : kFly ? (int)s_texL[c] : (int)texture16(d1[(size_t)v * P.W + u]);
        const int d = support_match_rows(L, R, t1, u, false, in, P, gl);
        if (gl == 0 && c < kSB) s_fwd[c] = (int16_t)(have ? d : -1);
    }
    __syncthreads();
    // Only the candidates whose forward search produced a disparity (about half of them) are
    // searched backwards.  They are compacted first, so that the backward searches fill whole
    // waves (with two or more forward rounds per block: usually all of them, like the forward rounds).
    if (wave == 0) {
        int base = 0;
#pragma unroll
        for (int c0 = 0; c0 < kSB; c0 += kWave) {
            const int c = c0 + lane;
            const int mine = c < ncand ? (int)s_fwd[c] : -1;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(mine >= 0);
            if (mine >= 0) s_todo[base + __builtin_popcountll(mask & ((1ull << lane) - 1))] = (uint8_t)c;
            base += __builtin_popcountll(mask);
            // everything else is settled now: column 0 stays at calloc's 0, no forward match -> -1
            if (c < ncand && mine < 0) dcan[c] = (int16_t)((uc0 + c) > 0 ? -1 : 0);
        }
        if (lane == 0) s_ntodo = base;
    }
    __syncthreads();
    const int ntodo = s_ntodo;
    for (int t0 = 4 * wave; t0 < ntodo; t0 += kPerRound) {
        const int t = t0 + grp;
        const bool have = t < ntodo;
        const int c = have ? (int)s_todo[t] : 0;
        const int d = have ? (int)s_fwd[c] : 0;
        const int uc = uc0 + c, u = uc * P.step;
        const int ub = have ? u - d : 5;   // >= 5 because d <= u-5
        const int t2 = !have ? 0 : kFly ? (int)s_texR[ub - xr0] : (int)texture16(d2[(size_t)v * P.W + ub]);
        const int dd = support_match_rows(R, L, t2, ub, true, have, P, gl);
        const int diff = d > dd ? d - dd : dd - d;
        const int out = (dd >= 0 && diff <= P.lr_threshold) ? d : -1;
        if (have && gl == 0) dcan[c] = (int16_t)out;
    }
}

// ---------------------------------------------------------------------------
// E8 + raster records, one thread per triangle of the group
//   Elas::computeDisparityPlanes   libelas/src/elas.cpp:605-680
//   Matrix::solve (Gauss-Jordan, full pivoting, eps 1e-20, ">=" pivot search)
//                                  libelas/src/matrix.cpp:414-501
//   triangle edges / validity      libelas/src/elas.cpp:1026-1072
// fp64 with IEEE division and no contraction: the same operation sequence as
// the reference's double code, so the float planes come out identical.
// ---------------------------------------------------------------------------
__device__ bool solve3(double A[3][3], double B[3]) {
    bool used[3] = {false, false, false};
    for (int it = 0; it < 3; it++) {
        double big = 0.0;
        int pr = 0, pc = 0;
        for (int j = 0; j < 3; j++) {
            if (used[j]) continue;
            for (int k = 0; k < 3; k++)
                if (!used[k] && fabs(A[j][k]) >= big) {  // ">=": the last maximum wins
                    big = fabs(A[j][k]);
                    pr = j;
                    pc = k;
                }
        }
        used[pc] = true;
        if (pr != pc) {
            for (int l = 0; l < 3; l++) {
                double t = A[pr][l];
                A[pr][l] = A[pc][l];
                A[pc][l] = t;
            }
            double t = B[pr];
            B[pr] = B[pc];
            B[pc] = t;
        }
        if (fabs(A[pc][pc]) < 1e-20) return false;
        const double inv = __ddiv_rn(1.0, A[pc][pc]);
        A[pc][pc] = 1.0;
        for (int l = 0; l < 3; l++) A[pc][l] = __dmul_rn(A[pc][l], inv);
        B[pc] = __dmul_rn(B[pc], inv);
        for (int r = 0; r < 3; r++) {
            if (r == pc) continue;
            const double f = A[r][pc];
            A[r][pc] = 0.0;
            for (int l = 0; l < 3; l++) A[r][l] = __dsub_rn(A[r][l], __dmul_rn(A[pc][l], f));
            B[r] = __dsub_rn(B[r], __dmul_rn(B[pc], f));
        }
    }
    return true;
}

__global__ __launch_bounds__(256) void k_prior(GroupDev G, int total_tri_arg) {
    // two lanes per triangle: lane rs = 0 fits the plane in left-image coordinates (t1a..c),
    // rs = 1 in right-image coordinates (t2a..c); the even lane then builds the raster record.
    // total_tri_arg < 0: the header was built on the device and holds the count; the grid is a
    // bound and blocks stride over the triangles that exist
    const int total_tri = total_tri_arg >= 0 ? total_tri_arg : G.hdr->total_tri;
    for (int gid = blockIdx.x * 256 + threadIdx.x; (gid - (int)threadIdx.x) < 2 * total_tri; gid += gridDim.x * 256) {
    const int T = gid >> 1, rs = gid & 1;
    const bool live = T < total_tri;
    const int Tc = live ? T : 0;
    int first;
    const int slot = tri_slot(G.hdr, Tc, &first);
    const int pair = slot >> 1, side = slot & 1;
    const int32_t* sup = G.support + 3 * (size_t)G.hdr->sup_off[pair];
    const int32_t* c = G.tri + 3 * (size_t)Tc;
    int32_t su[3], sv[3], sd[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int32_t* s = sup + 3 * c[k];
        su[k] = s[0];
        sv[k] = s[1];
        sd[k] = s[2];
    }
    float mine[3];
    {
        double A[3][3], B[3];
        for (int r = 0; r < 3; r++) {
            A[r][0] = (double)(rs ? su[r] - sd[r] : su[r]);
            A[r][1] = (double)sv[r];
            A[r][2] = 1.0;
            B[r] = (double)sd[r];
        }
        if (solve3(A, B)) {
            mine[0] = (float)B[0];
            mine[1] = (float)B[1];
            mine[2] = (float)B[2];
        } else {
            mine[0] = mine[1] = mine[2] = 0.f;
        }
    }
    float pl[6];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float other = __shfl_xor(mine[k], 1);   // partner lane's plane (all lanes take part)
        pl[k] = rs ? other : mine[k];
        pl[3 + k] = rs ? mine[k] : other;
    }
    if (!live || rs) continue;
#pragma unroll
    for (int k = 0; k < 6; k++) G.planes[6 * (size_t)T + k] = pl[k];

    // corners sorted by u with the reference's exchange loop (not stable on ties)
    float tu[3], tv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        tu[k] = side ? (float)(su[k] - sd[k]) : (float)su[k];
        tv[k] = (float)sv[k];
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < j; k++)
            if (tu[k] > tu[j]) {
                float a = tu[j]; tu[j] = tu[k]; tu[k] = a;
                float b = tv[j]; tv[j] = tv[k]; tv[k] = b;
            }
    TriRaster r;
    const float Au = tu[0], Av = tv[0], Bu = tu[1], Bv = tv[1], Cu = tu[2], Cv = tv[2];
    r.uA = (int32_t)Au;
    r.uB = (int32_t)Bu;
    r.uC = (int32_t)Cu;
    float ABa = 0.f, ACa = 0.f, BCa = 0.f;
    if (r.uA != r.uB) ABa = __fdiv_rn(__fsub_rn(Av, Bv), __fsub_rn(Au, Bu));
    if (r.uA != r.uC) ACa = __fdiv_rn(__fsub_rn(Av, Cv), __fsub_rn(Au, Cu));
    if (r.uB != r.uC) BCa = __fdiv_rn(__fsub_rn(Bv, Cv), __fsub_rn(Bu, Cu));
    r.ABa = ABa;
    r.ACa = ACa;
    r.BCa = BCa;
    r.ABb = __fsub_rn(Av, __fmul_rn(ABa, Au));
    r.ACb = __fsub_rn(Av, __fmul_rn(ACa, Au));
    r.BCb = __fsub_rn(Bv, __fmul_rn(BCa, Bu));
    const float pa = side ? pl[3] : pl[0];
    const float pd = side ? pl[0] : pl[3];
    r.pa = pa;
    r.pb = side ? pl[4] : pl[1];
    r.pc = side ? pl[5] : pl[2];
    r.valid = ((double)fabsf(pa) < 0.7 && (double)fabsf(pd) < 0.7) ? 1 : 0;
    r.slot = slot;      // k_owner takes these from the record instead of scanning tri_end[]
    r.first = first;
    r.pad_[0] = 0;
    G.raster[T] = r;
    }
}

// ---------------------------------------------------------------------------
// E9  Elas::createGrid   libelas/src/elas.cpp:684-780
// The grid is kept as one bit set of disparities per 20x20 cell (disp_max+1
// bits).  k_grid_seed marks d-1..d+1 of every support point (left: cell of u,
// right: cell of u-d); k_grid_dilate is the reference's flat 3x3 walk over cells
// gw+1 .. cells-gw-2 (columns wrap, first/last rows stay empty).  Bit order =
// ascending disparity = the reference's list order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_seed(GroupDev G, int total_sup_arg, int gw, int gh,
                                                   int gwords, int grid_size, int disp_max) {
    const int total_sup = total_sup_arg >= 0 ? total_sup_arg : G.hdr->total_sup;
    for (int S = blockIdx.x * 256 + threadIdx.x; S < total_sup; S += gridDim.x * 256) {
    int pair = 0;
    while (pair < G.hdr->npairs - 1 && S >= G.hdr->sup_off[pair + 1]) pair++;
    const int32_t* s = G.support + 3 * (size_t)S;
    const int xc = s[0], yc = s[1], dc = s[2];
    const int y = (int)floorf(__fdiv_rn((float)yc, (float)grid_size));
    const int xl = (int)floorf((float)(xc / grid_size));  // integer division first (elas.cpp:712)
    const int xr = (int)floorf(__fdiv_rn((float)(xc - dc), (float)grid_size));
    const int lo = dc - 1 > 0 ? dc - 1 : 0, hi = dc + 1 < disp_max ? dc + 1 : disp_max;
    const size_t cells = (size_t)gw * gh;
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const int x = side ? xr : xl;
        if (x < 0 || x >= gw || y < 0 || y >= gh) continue;
        uint32_t* cell = G.seed + (((size_t)(2 * pair + side) * cells) + (size_t)y * gw + x) * gwords;
        for (int dd = lo; dd <= hi; dd++) atomicOr(&cell[dd >> 5], 1u << (dd & 31));
    }
    }
}

__global__ __launch_bounds__(256) void k_grid_dilate(GroupDev G, int slots, int gw, int gh,
                                                     int gwords) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int cells = gw * gh;
    const int per = cells * gwords;
    if (i >= slots * per) return;
    const int z = i / per, rem = i - z * per;
    const int c = rem / gwords, w = rem - c * gwords;
    uint32_t o = 0;
    if (c >= gw + 1 && c <= cells - gw - 2) {
        const uint32_t* base = G.seed + (size_t)z * per + w;
        const int nb[9] = {-gw - 1, -gw, -gw + 1, -1, 0, 1, gw - 1, gw, gw + 1};
#pragma unroll
        for (int k = 0; k < 9; k++) o |= base[(size_t)(c + nb[k]) * gwords];
    }
    G.mask[i] = o;
}

// Candidate records for k_match_list (disp_max <= 255): one wave per cell turns the cell's bit set (up to 256 bits) into
// ML_CAP = 32 uint16: [0..27] the candidates, ascending, as 16 * d (the byte offset of the candidate in a
// descriptor row), padded with the last one; [28..30] the last candidate; [31] their number.  Cells with more
// than 28 candidates are decoded from the bit set by the matcher itself.
__global__ __launch_bounds__(256) void k_grid_list(GroupDev G, int ncells, int gwords) {
    const int lane = threadIdx.x & 63;
    const int cell = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    if (cell >= ncells) return;
    const uint32_t* bits = G.mask + (size_t)cell * gwords;     // gwords <= 8 (disp_max <= 255)
    const uint32_t wq = lane < gwords ? bits[lane] : 0u;
    uint16_t* rec = G.lists + (size_t)cell * 32;
    int n = 0, last = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint64_t m = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wq, 2 * q) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wq, 2 * q + 1) << 32);
        if ((m >> lane) & 1) {
            const int idx = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (idx < 28) rec[idx] = (uint16_t)((64 * q + lane) * 16);
        }
        if (m) last = 64 * q + 63 - __builtin_clzll(m);
        n += __builtin_popcountll(m);
    }
    // (the padding is written by other lanes than the candidates: no ordering needed)
    if (lane < 31 && lane >= n) rec[lane] = (uint16_t)(last * 16);
    if (lane == 31) rec[31] = (uint16_t)n;
    if (lane >= 28 && lane < 31 && lane < n) rec[lane] = (uint16_t)(last * 16);
}

// ---------------------------------------------------------------------------
// E10 (ownership)  Elas::computeDisparity triangle rasterisation
//   libelas/src/elas.cpp:1003-1115
// The reference walks triangles in list order and lets a later triangle
// overwrite an earlier one on the few pixels both cover; findMatch's early-outs
// depend on the pixel only, so "owner = highest triangle index covering the
// pixel" is exact.  One wave per triangle: 16 columns x 4 row phases per step.
// Two passes instead of one atomicMax per pixel:
// ---------------------------------------------------------------------------
// kFix = false: plain stores (on the rare multiply covered pixel an arbitrary
// coverer lands); kFix = true: every triangle re-reads its pixels and raises the
// ones a lower index won with atomicMax.  Per-pixel atomics are thereby limited
// to the handful of contested pixels (4-53 per image in the survey's probes).
template <bool kFix>
__global__ __launch_bounds__(256) void k_owner(GroupDev G, int total_tri_arg, int W, int H, int sub, int fix_all) {
    const int total_tri = total_tri_arg >= 0 ? total_tri_arg : G.hdr->total_tri;   // see k_prior
    const int lane = threadIdx.x & 63;
    // XCD-aware order (launched with a multiple of 8 workgroups): XCD k takes the k-th eighth of the triangles THERE
    // ARE (the launch is sized for a bound when the count is only known on the device), which Triangle emits in an
    // order that keeps neighbours close -- the partial lines two neighbouring triangles write meet in one L2
    const int need = (((total_tri + 3) >> 2) + 7) >> 3, have = (int)(gridDim.x >> 3);
    const int per_xcd = need < have ? need : have;          // workgroups per XCD that take part
    if ((int)(blockIdx.x >> 3) >= per_xcd) return;
    const int bid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    for (int T = __builtin_amdgcn_readfirstlane((int)((bid * 256 + threadIdx.x) >> 6)); T < total_tri;
         T += per_xcd * 32) {
    const TriRaster tr = G.raster[T];
    const int slot = tr.slot, first = tr.first;
    // stored value = owner_base + 1 + triangle index: everything <= owner_base is a leftover of
    // an earlier group and reads as "no triangle", so the maps need no clearing between groups
    const int t = G.owner_base + 1 + (T - first);
    int32_t* owner = G.owner + (size_t)slot * W * H;
    const int cl = lane & 15, rp = lane >> 4;
#pragma unroll
    for (int part = 0; part < 2; part++) {
        const int lo = part ? tr.uB : tr.uA, hi = part ? tr.uC : tr.uB;
        if (lo == hi) continue;
        const float ea = part ? tr.BCa : tr.ABa, eb = part ? tr.BCb : tr.ABb;
        // (the matchers return at once in the two columns at either end of a row -- elas.cpp:797-798 -- so those columns
        // are not written: their words stay stale = "no triangle", and k_match_list needs no column test of its own)
        const int ulo = lo > 2 ? lo : 2, uhi = hi < W - 2 ? hi : W - 2;
        for (int u = ulo + cl; u < uhi; u += 16) {
            if (sub && (u & 1)) continue;
            const float fu = (float)u;
            const int v1 = f2u2i(__fadd_rn(__fmul_rn(tr.ACa, fu), tr.ACb));
            const int v2 = f2u2i(__fadd_rn(__fmul_rn(ea, fu), eb));
            int va = v1 < v2 ? v1 : v2, vb = v1 < v2 ? v2 : v1;
            va = va > 0 ? va : 0;
            vb = vb < H ? vb : H;
            if (kFix && !fix_all) {
                // Only the ends of a column span can be contested.  The triangles partition the
                // plane, so on the line x = u their exact spans [a, b] touch at most in a point; a
                // rasterised span is [floor(a + e1), floor(b + e2)) with |e| << 1 from the float edge
                // lines, hence two spans can share at most the one row in which an integer falls
                // between two evaluations of the same boundary: the last row of the lower span, the
                // first of the upper one.  The pass re-checks the two first and two last rows of
                // every span (one row of margin): rows va, va+1, vb-2, vb-1 on the four row phases.
                const int v = rp < 2 ? va + rp : vb - 4 + rp;
                const bool mine = v >= va && v < vb && (rp < 2 || v >= va + 2);
                if (mine && !(sub && (v & 1))) {
                    int32_t* px = &owner[(size_t)v * W + u];
                    // a plain (possibly L1-stale) read is enough: owners only grow, so a stale
                    // value can only trigger a redundant atomicMax, never suppress a needed one
                    if (*px < t) atomicMax(px, t);
                }
            } else {
                for (int v = va + rp; v < vb; v += 4) {
                    if (sub && (v & 1)) continue;
                    int32_t* px = &owner[(size_t)v * W + u];
                    if (kFix) {   // SVH_OWNER_FIX_ALL: every pixel of the span (the exhaustive form, for tests)
                        if (*px < t) atomicMax(px, t);
                    } else {
                        *px = t;
                    }
                }
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------
// E11  Elas::findMatch + updatePosteriorMinimum   libelas/src/elas.cpp:784-955
// One thread per disparity-map pixel (blockIdx.z = 2*pair + side); lanes are
// consecutive u, so for a common candidate disparity the 64 descriptor loads of
// a wave are one contiguous 1 KiB segment of the other image's descriptor row.
// ---------------------------------------------------------------------------
struct MatchParams {
    int W, H, DW, DH, gw, gh, gwords, grid_size, sub;
    int disp_max, match_texture, plane_radius, npairs;
    uint32_t grid_magic;   // floor(2^32 / grid_size) + 1: u / grid_size == mulhi(u, magic), u < 2^16
};

// kLds: one block = one row of the disparity map; the OTHER image's descriptor
// row is staged in LDS once (19.9 KB at W = 1242) and every candidate SAD reads
// it from there instead of ~13 L2 reads per pixel; no row is loaded twice.
template <bool kLds>
__global__ __launch_bounds__(kLds ? 512 : 256) void k_match(GroupDev G, MatchParams P) {
    extern __shared__ uint4 s_row[];
    __shared__ int s_P[64];
    const int z = blockIdx.z, pair = z >> 1, side = z & 1;
    if (!G.hdr->active[pair]) return;
    const int y = kLds ? blockIdx.y : blockIdx.y * 4 + threadIdx.y;
    {
        // prior table: only |d - d_plane| <= plane_radius is ever indexed
        const int tl = kLds ? (int)threadIdx.x : (int)(threadIdx.y * 64 + threadIdx.x);
        if (tl < 64) s_P[tl] = tl <= P.disp_max ? G.P[tl] : 0;
    }
    const size_t N = (size_t)P.W * P.H;
    const int mul = P.sub ? 2 : 1;
    const int v = y * mul;
    int line = v < P.H - 3 ? v : P.H - 3;
    line = line > 2 ? line : 2;
    const uint4* oth_line =
        reinterpret_cast<const uint4*>(G.desc + (size_t)(z ^ 1) * N * 16) + (size_t)line * P.W;
    const int s0 = 0;
    if (kLds)
        for (int i = threadIdx.x; i < P.W; i += blockDim.x) s_row[i] = oth_line[i];
    __syncthreads();
    for (int x = kLds ? (int)threadIdx.x : (int)(blockIdx.x * 64 + threadIdx.x); x < P.DW;
         x += kLds ? (int)blockDim.x : P.DW) {
    if (y >= P.DH) return;
    const int u = x * mul;
    float out = -10.f;
    const int tri0 = z ? G.hdr->tri_end[z - 1] : 0;
    const int t = G.owner[(size_t)z * N + (size_t)v * P.W + u] - G.owner_base - 1;
    if (t >= 0 && u >= 2 && u < P.W - 2) {
        const uint4* own_line =
            reinterpret_cast<const uint4*>(G.desc + (size_t)z * N * 16) + (size_t)line * P.W;
        const uint4 own = own_line[u];
        if ((int)texture16(own) >= P.match_texture) {
            const float4 pl = *reinterpret_cast<const float4*>(G.raster + tri0 + t);   // pa pb pc valid
            const float pa = pl.x, pb = pl.y, pc = pl.z;
            const int valid = __float_as_int(pl.w);
            const int d_plane = (int)__fadd_rn(
                __fadd_rn(__fmul_rn(pa, (float)u), __fmul_rn(pb, (float)v)), pc);
            int dlo = d_plane - P.plane_radius;
            dlo = dlo > 0 ? dlo : 0;
            int dhi = d_plane + P.plane_radius;
            dhi = dhi < P.disp_max ? dhi : P.disp_max;
            const int cell = (v / P.grid_size) * P.gw + u / P.grid_size;
            const uint32_t* bits = G.mask + ((size_t)z * P.gw * P.gh + cell) * P.gwords;
            int min_val = 10000, min_d = -1;
            // candidate disparities of the cell, ascending; the 32-byte bit set of the common
            // disp_max = 255 case is fetched with two 16-byte loads before the loop starts
            uint32_t wbuf[8];
            const bool fast8 = P.gwords == 8;
            if (fast8) {
                const uint4 lo4 = reinterpret_cast<const uint4*>(bits)[0];
                const uint4 hi4 = reinterpret_cast<const uint4*>(bits)[1];
                wbuf[0] = lo4.x; wbuf[1] = lo4.y; wbuf[2] = lo4.z; wbuf[3] = lo4.w;
                wbuf[4] = hi4.x; wbuf[5] = hi4.y; wbuf[6] = hi4.z; wbuf[7] = hi4.w;
            }
            for (int w = 0; w < P.gwords; w++) {
                uint32_t b;
                if (fast8) {
                    b = wbuf[0];
#pragma unroll
                    for (int q = 1; q < 8; q++) b = (w == q) ? wbuf[q] : b;
                } else {
                    b = bits[w];
                }
                while (b) {
                    const int dc = w * 32 + __builtin_ctz(b);
                    b &= b - 1;
                    if (dc < dlo || dc > dhi) {
                        const int uw = side ? u + dc : u - dc;
                        if (uw < 2 || uw >= P.W - 2) continue;
                        const uint4 o = kLds ? s_row[uw - s0] : oth_line[uw];
                        const int val = (int)sad16(own, o);
                        if (val < min_val) {
                            min_val = val;
                            min_d = dc;
                        }
                    }
                }
            }
            for (int dc = dlo; dc <= dhi; dc++) {
                const int uw = side ? u + dc : u - dc;
                if (uw < 2 || uw >= P.W - 2) continue;
                int dd = dc - d_plane;
                dd = dd < 0 ? -dd : dd;
                const uint4 o = kLds ? s_row[uw - s0] : oth_line[uw];
                // (two separate loads: selecting between an LDS and a global POINTER would make the
                // access a flat load)
                int prior = s_P[dd < 64 ? dd : 63];
                asm volatile("" : "+v"(prior));   // keeps the two loads from being merged back
                if (dd >= 64) prior = G.P[dd];
                const int val = (int)sad16(own, o) + (valid ? prior : 0);
                if (val < min_val) {
                    min_val = val;
                    min_d = dc;
                }
            }
            out = min_d >= 0 ? (float)min_d : -1.f;
        }
    }
    G.Draw[(size_t)z * P.DW * P.DH + (size_t)y * P.DW + x] = out;
    }
}

// ---------------------------------------------------------------------------
// E11, issue-lean variant (the default).  k_match above is bound by instruction
// issue (VALU + SALU, rocprofv3 SQ counters), not by memory, so this version
// spends fewer instructions per candidate:
//  * the reference's "first minimum wins" scan order (cell candidates outside
//    the plane band ascending, then the band ascending; elas.cpp:868-953) is
//    encoded in the key  cost*1024 + rank  with rank = d for cell candidates and
//    512 + d for band candidates; the winner is then a plain minimum, so the
//    candidates can be evaluated in any order, two per loop trip (two LDS reads
//    in flight, one v_min3), without the ordered compare/select chain;
//  * band bits are cleared from the cell's bit set up front (no per-candidate
//    band test) and the warp-range test is skipped when the whole wave is far
//    enough from the image border (wave-uniform branch);
//  * u / grid_size is one v_mul_hi.
// Needs |cost| < 2^20 and d < 512; launch_match falls back to k_match otherwise.
// ---------------------------------------------------------------------------
template <bool kCheck>
__device__ __forceinline__ void scan_word(uint32_t b, int base, int pos, int W, const uint4& own,
                                          const uint4* row, int& best) {
    while (b) {
        const int i0 = __builtin_ctz(b);
        b &= b - 1;
        const bool has1 = b != 0;
        const int i1 = has1 ? __builtin_ctz(b) : i0;   // odd count: evaluate the same one twice
        b &= b - 1;
        const int dc0 = base + i0, dc1 = base + i1;
        int uw0 = pos + dc0, uw1 = pos + dc1;
        bool ok0 = true, ok1 = true;
        if (kCheck) {
            ok0 = (uint32_t)(uw0 - 2) < (uint32_t)(W - 4);
            ok1 = (uint32_t)(uw1 - 2) < (uint32_t)(W - 4);
            uw0 = ok0 ? uw0 : 2;
            uw1 = ok1 ? uw1 : 2;
        }
        const uint4 o0 = row[uw0], o1 = row[uw1];
        int k0 = (int)(sad16(own, o0) << 10) + dc0;
        int k1 = (int)(sad16(own, o1) << 10) + dc1;
        if (kCheck) {
            k0 = ok0 ? k0 : 0x7FFFFFFF;
            k1 = ok1 ? k1 : 0x7FFFFFFF;
        }
        const int k = k0 < k1 ? k0 : k1;
        best = k < best ? k : best;
    }
}

template <bool kCheck>
__device__ __forceinline__ float match_pixel_keyed(const uint4& own, const float4& pl, int u, int v, int pos,
                                                   const uint4* row, const uint32_t* __restrict__ bits,
                                                   const int* s_P, const int32_t* __restrict__ gP,
                                                   const MatchParams& P) {
    const int valid = __float_as_int(pl.w);
    const int d_plane =
        (int)__fadd_rn(__fadd_rn(__fmul_rn(pl.x, (float)u), __fmul_rn(pl.y, (float)v)), pl.z);
    int dlo = d_plane - P.plane_radius;
    dlo = dlo > 0 ? dlo : 0;
    int dhi = d_plane + P.plane_radius;
    dhi = dhi < P.disp_max ? dhi : P.disp_max;
    const int len = dhi - dlo + 1;   // <= 2*radius+1 <= 31; <= 0: empty band
    // clear the band from the bit set: it spans at most two words
    int wlo = -9;
    uint32_t m0 = ~0u, m1 = ~0u;
    if (len > 0) {
        const uint32_t bm = (1u << len) - 1u;
        const int sh = dlo & 31;
        wlo = dlo >> 5;
        m0 = ~(bm << sh);
        m1 = sh ? ~(bm >> (32 - sh)) : ~0u;
    }
    const int vmask = valid ? -1 : 0;   // planes of invalid triangles carry no prior (elas.cpp:921-927)
    int best = 0x7FFFFFFF;
    if (P.gwords == 8) {
        const uint4 lo4 = reinterpret_cast<const uint4*>(bits)[0];
        const uint4 hi4 = reinterpret_cast<const uint4*>(bits)[1];
        const uint32_t wb[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            // the cells of one wave hold similar candidate sets, clustered in a few of the eight
            // words: a word that is empty in every lane is skipped before any per-lane work
            if (__builtin_amdgcn_ballot_w64(wb[q] != 0) == 0) continue;
            const uint32_t keep = q == wlo ? m0 : (q == wlo + 1 ? m1 : ~0u);
            scan_word<kCheck>(wb[q] & keep, q * 32, pos, P.W, own, row, best);
        }
    } else {
        for (int q = 0; q < P.gwords; q++) {
            const uint32_t keep = q == wlo ? m0 : (q == wlo + 1 ? m1 : ~0u);
            scan_word<kCheck>(bits[q] & keep, q * 32, pos, P.W, own, row, best);
        }
    }
    // the band, two disparities per trip, with the plane prior
    for (int dc = dlo; dc <= dhi; dc += 2) {
        const int dc1 = dc + 1 <= dhi ? dc + 1 : dc;
        int uw0 = pos + dc, uw1 = pos + dc1;
        bool ok0 = true, ok1 = true;
        if (kCheck) {
            ok0 = (uint32_t)(uw0 - 2) < (uint32_t)(P.W - 4);
            ok1 = (uint32_t)(uw1 - 2) < (uint32_t)(P.W - 4);
            uw0 = ok0 ? uw0 : 2;
            uw1 = ok1 ? uw1 : 2;
        }
        const uint4 o0 = row[uw0], o1 = row[uw1];
        int dd0 = dc - d_plane, dd1 = dc1 - d_plane;
        dd0 = dd0 < 0 ? -dd0 : dd0;
        dd1 = dd1 < 0 ? -dd1 : dd1;
        // dd <= plane_radius <= 15 here (launch_match), always inside the LDS copy of the table
        const int p0 = s_P[dd0] & vmask;
        const int p1 = s_P[dd1] & vmask;
        int k0 = ((int)sad16(own, o0) + p0) * 1024 + 512 + dc;
        int k1 = ((int)sad16(own, o1) + p1) * 1024 + 512 + dc1;
        if (kCheck) {
            k0 = ok0 ? k0 : 0x7FFFFFFF;
            k1 = ok1 ? k1 : 0x7FFFFFFF;
        }
        const int k = k0 < k1 ? k0 : k1;
        best = k < best ? k : best;
    }
    return best != 0x7FFFFFFF ? (float)(best & 511) : -1.f;
}

// kLr: the left/right consistency check (E12, elas.cpp:1122-1204) of a row only needs the raw
// left and right disparities of that same row, which this block has just produced: they stay
// in LDS, and the checked maps go straight to `out` -- no raw-map round trip, no k_lr launch.
template <bool kLr>
__global__ __launch_bounds__(512) void k_match_keyed(GroupDev G, MatchParams P, DevMaps out, int write_raw,
                                                     float lr_threshold) {
    // One block = one image row of one pair, BOTH disparity maps: the left-map pass compares
    // L[row] with R[row], the right-map pass R[row] with L[row], so the two descriptor rows
    // (2 x 19.9 KB at W = 1242) are staged in LDS once and serve as "own" and "other" row of
    // both passes; the first half of the block matches the left map, the second half the right.
    extern __shared__ uint4 s_rows[];   // [2][W]: row of image 1, row of image 2; then raw [2][DW]
    __shared__ int s_P[64];
    // (pair, image row); XCD-aware order as in k_match_list: the grid is a multiple of 8, XCD k takes the k-th
    // contiguous eighth of the rows
    const int total_rows = P.DH * P.npairs;
    const int row_id = (int)(blockIdx.x & 7) * ((total_rows + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (row_id >= total_rows) return;
    const int pair = row_id / P.DH, y = row_id - pair * P.DH;
    if (!G.hdr->active[pair]) return;
    if (threadIdx.x < 64) s_P[threadIdx.x] = (int)threadIdx.x <= P.disp_max ? G.P[threadIdx.x] : 0;
    const size_t N = (size_t)P.W * P.H;
    const int mul = P.sub ? 2 : 1;
    const int v = y * mul;
    int line = v < P.H - 3 ? v : P.H - 3;
    line = line > 2 ? line : 2;
    // the image-2 row is stored REVERSED: a left-map candidate u - d is then slot (W-1-u) + d, a
    // right-map candidate u + d slot u + d of the image-1 row -- position + disparity on both
    // sides, one v_lshl_add per address
    if (G.desc_fly) {
        // round 5: the two rows assembled from the Sobel planes (as k_match_list does), so that subsampling and
        // disp_max > 255 run without the 32 N bytes of descriptor maps too; tasks alternate between the images
        const int nq = (P.W + 3) >> 2;
        for (int task = (int)threadIdx.x; task < 2 * nq; task += (int)blockDim.x) {
            const int im = task & 1, x = 4 * (task >> 1);
            uint4 o[4];
            fly_desc4(G.desc + (size_t)(2 * pair + im) * N * 16, P.W, P.H, x, line, o);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (x + i < P.W) s_rows[im ? 2 * P.W - 1 - (x + i) : x + i] = o[i];
        }
    } else {
        const uint4* l1 = reinterpret_cast<const uint4*>(G.desc + (size_t)(2 * pair) * N * 16) + (size_t)line * P.W;
        const uint4* l2 = l1 + N;
        stage_slots<5>(s_rows, 2 * P.W, (int)threadIdx.x, (int)blockDim.x,
                       [&](int i) { return i < P.W ? l1[i] : l2[2 * P.W - 1 - i]; });
    }
    __syncthreads();
    const int half = blockDim.x >> 1;
    const int side = (int)threadIdx.x >= half;        // wave-uniform: half is a multiple of 64
    const int z = 2 * pair + side;
    const uint4* oth_row = s_rows + (1 - side) * P.W;
    const int tri0 = z ? G.hdr->tri_end[z - 1] : 0;
    const int32_t* own_t = G.owner + (size_t)z * N + (size_t)v * P.W;
    const uint32_t* row_bits =
        G.mask + ((size_t)z * P.gw * P.gh + (size_t)(v / P.grid_size) * P.gw) * P.gwords;
    float* out_row = G.Draw + (size_t)z * P.DW * P.DH + (size_t)y * P.DW;
    float* s_raw = reinterpret_cast<float*>(s_rows + 2 * P.W);   // [2][DW] raw disparities of the row
    for (int x = (int)threadIdx.x - side * half; x < P.DW; x += half) {
        const int u = x * mul;
        float out = -10.f;
        const int t = own_t[u] - G.owner_base - 1;   // < 0: no triangle of this group covers the pixel
        const bool live = t >= 0 && u >= 2 && u < P.W - 2;
        // every disparity up to disp_max warps inside [2, W-2) for the whole wave?
        const bool inner = side ? u + P.disp_max < P.W - 2 : u - P.disp_max >= 2;
        const bool wave_inner = __builtin_amdgcn_ballot_w64(live && !inner) == 0;
        if (live) {
            const int pos = side ? u : P.W - 1 - u;          // this pixel in the other row's slot order
            const uint4 own = side ? s_rows[2 * P.W - 1 - u] : s_rows[u];
            if ((int)texture16(own) >= P.match_texture) {
                const float4 pl = *reinterpret_cast<const float4*>(G.raster + tri0 + t);
                const uint32_t* bits = row_bits + __umulhi((uint32_t)u, P.grid_magic) * (uint32_t)P.gwords;
                out = wave_inner
                          ? match_pixel_keyed<false>(own, pl, u, v, pos, oth_row, bits, s_P, G.P, P)
                          : match_pixel_keyed<true>(own, pl, u, v, pos, oth_row, bits, s_P, G.P, P);
            }
        }
        if (!kLr || write_raw) out_row[x] = out;
        if (kLr) s_raw[side * P.DW + x] = out;
    }
    if (!kLr) return;
#if SVH_ML_PROBE == 6
    return;
#endif
    __syncthreads();
    // E12: keep d iff the other map, at the warped position, agrees within lr_threshold
    float* D = out.D[side] + (size_t)pair * out.stride[side] + (size_t)y * P.DW;
    const float* mine = s_raw + side * P.DW;
    const float* other = s_raw + (1 - side) * P.DW;
    for (int x = (int)threadIdx.x - side * half; x < P.DW; x += half) {
        const float d = mine[x];
        const float fx = (float)x;
        const float step = P.sub ? d / 2 : d;
        const float uw = side ? fx + step : fx - step;
        float o = -10.f;
        if (d >= 0 && uw >= 0 && uw < (float)P.DW)
            if (!(fabsf(other[(int)uw] - d) > lr_threshold)) o = d;
        D[x] = o;
    }
}

// ---------------------------------------------------------------------------
// E11 + E12, list form (round 4; the default for the presets).
//   Elas::findMatch / updatePosteriorMinimum     libelas/src/elas.cpp:784-955
//   Elas::leftRightConsistencyCheck              libelas/src/elas.cpp:1122-1204
// Block structure of k_match_keyed (one block = one image row of one pair, both maps, the two
// descriptor rows staged once, raw disparities and the L/R check in LDS), with the scan rebuilt:
//  * candidates come from per-cell LISTS (k_grid_list: 32 uint16 per cell, 16*d, padded with the last
//    one), staged in LDS with the rows: a lane reads four candidates with one ds_read_b64 instead of
//    decoding its cell's bit set (~19 operations per candidate before);
//  * a candidate costs one address operation, four v_sad_hi_u8 and half a v_min3: the instruction
//    accumulates (SAD << 16) on top of its third operand, so a chain started on the candidate's rank
//    (its list index; 512 + d in the plane band) IS the key  cost << 16 | rank  of the keyed minimum;
//  * cell candidates inside the plane band are not excluded per lane (elas.cpp:871): the band scan
//    evaluates the same disparity with its prior, and while every prior of the band is negative
//    (and the plane valid) that key is strictly smaller, so the extra key never wins; waves that
//    cannot rely on this, or sit next to the image border, take the checked form of the loops;
//  * rows and lists land by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write
//    pass; owner words and planes of all of a thread's pixels are requested before the scan starts.
// ---------------------------------------------------------------------------
// SVH_ML_PROBE (tools/Makefile `probe`, never in the product build): cut-down instances of k_match_list whose
// instruction counters, subtracted from one another, give the kernel's VALU split (profiles/r06_match_split.txt):
//   1 staging only   2 + own descriptor, texture, live vote   3 + plane, plan, votes   4 fast form without the band
//   5 fast form without the cell loop   6 everything but the L/R pass   7 event counters (g_ml_cnt)   8 no cold redo
#ifndef SVH_ML_PROBE
#define SVH_ML_PROBE 0
#endif
#if SVH_ML_PROBE == 7
__device__ unsigned long long g_ml_cnt[16];
#define ML_CNT(i, n) do { if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_ml_cnt[i], (unsigned long long)(n)); } while (0)
#else
#define ML_CNT(i, n) do { } while (0)
#endif
constexpr int ML_CAP = 32;     // uint16 per cell record: [0..27] candidates, [28..30] last candidate, [31] count
constexpr int ML_FAST = 28;    // cells with more candidates are decoded from their bit set

__device__ __forceinline__ int min3i(int a, int b, int c) {
    const int m = b < c ? b : c;
    return a < m ? a : m;
}
// n 16-byte slots global -> LDS by the whole block, lane-linear (LDS-DMA)
__device__ __forceinline__ void dma_slots(uint4* dst, const uint4* src, int n, int wave, int nwaves, int lane) {
    for (int j0 = wave * kWave; j0 < n; j0 += nwaves * kWave) {
        const int j = j0 + lane;
        if (j < n)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j),
                                             (__attribute__((address_space(3))) void*)(dst + j0), 16, 0, 0);
    }
}

struct MatchList {
    int kIters;      // pixels per thread (template argument of the launch)
    int half;        // threads per image side (a multiple of 64; the block is 2 * half)
    int xcd;         // 1: blockIdx -> row in eight contiguous runs, one per XCD
    int Wr;          // raw-row stride (int16)
    int Ws;          // descriptor-row stride in LDS (slots): >= 4 * ceil(W / 4) and == 2 (mod 4), see the staging loop
};

// per-pixel quantities both forms of the scan need
struct PixelPlan {
    int valid, d_plane, dlo, dhi, n, dmax;
    bool cell_in, band_in;
};
template <int kSide>
__device__ __forceinline__ PixelPlan ml_plan(const float4& pl, int u, int v, uint32_t lrec, const MatchParams& P) {
    PixelPlan q;
    q.valid = __float_as_int(pl.w);
    q.d_plane = (int)__fadd_rn(__fadd_rn(__fmul_rn(pl.x, (float)u), __fmul_rn(pl.y, (float)v)), pl.z);
    const int rad = P.plane_radius;
    q.dlo = q.d_plane - rad;
    q.dlo = q.dlo > 0 ? q.dlo : 0;
    q.dhi = q.d_plane + rad;
    q.dhi = q.dhi < P.disp_max ? q.dhi : P.disp_max;
    const uint32_t tail = lds_read4(lrec + 2 * (ML_CAP - 2));   // [30] last candidate, [31] count
    q.n = (int)(tail >> 16);
    q.dmax = (int)(tail & 0xFFFFu) >> 4;
    q.cell_in = kSide ? u + q.dmax < P.W - 2 : u - q.dmax >= 2;
    q.band_in = q.d_plane - rad >= 0 && q.d_plane + rad <= P.disp_max &&
                (kSide ? u + q.d_plane + rad < P.W - 2 : u - q.d_plane - rad >= 2);
    return q;
}

// min(a, b, c) as ONE operation (hipcc splits min(min(a, b), c) of four keys and the running best into three)
__device__ __forceinline__ int min3_op(int a, int b, int c) {
    int r;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Wave-uniform constants of the hot form (scalar registers).  Band slot j = the j-th of the five band candidates in
// LDS ADDRESS order: disparity d_plane - 2 + j in the right map (addresses rise with d), d_plane + 2 - j in the left.
struct FastK {
    int cell_c;           // cell_in  <=>  16 * dmax <= (right map ? cell_c - 16 u : 16 u - cell_c)
    int band_c, band_m;   // band_in  <=>  (unsigned)(d_plane - rad) < clamp(right map ? band_c - u : u - band_c, 0, band_m)
    uint32_t key[5];      // 512 + (disparity of slot j) - (d_plane - 2) + (P[|j - 2|] << 16): key start of a valid plane
    uint32_t key0[5];     // the same without the prior (invalid plane)
    uint32_t p16[3];      // P[0], P[1], P[2] << 16
};

// The reference's scan of one pixel, hot form (round 6: rebuilt around what the instruction split of round 5's form
// showed -- profiles/r06_match_split.txt): at most ML_FAST candidates in the cell and none of them warps out of the
// row (anything else is redone by ml_pixel_checked).  Wave-uniform switches:
//   excl     -- some lane's plane is invalid (its band has no prior) or a band prior is not negative: cell
//               candidates inside the band are skipped as the reference does (elas.cpp:871), +3 operations each;
//   band_ok  -- every lane's band lies inside [0, disp_max] and inside the row; otherwise the five band keys
//               are masked one by one.
// rowaddr: LDS address of the other image's slot at this pixel's column; lrec: LDS address of the cell's record;
// tail: the record's last word (count << 16 | 16 * largest candidate); t1 = d_plane - plane_radius.
//  * the padding of a record repeats its last candidate, and a repeat carries a higher rank than the original, so the
//    lanes of a trip are NOT masked by their own count any more: a wave's trips run on one execution mask (cells
//    without any candidate leave the loop's mask as a whole), and what ends the loop is the wave's longest list;
//  * two trips per loop turn with the next record words requested a trip ahead into alternating registers (no copies);
//  * min3 on (best, key, key): two operations per four candidates.
template <int kSide>
__device__ __forceinline__ int ml_fast(const uint4& own, uint32_t rowaddr, uint32_t lrec, uint32_t tail, int t1, int u,
                                       bool valid, bool excl, bool band_ok, bool edge, uint32_t rowlo, uint2 ca,
                                       const FastK& K, const int* s_band, const MatchParams& P) {
    const int rad = P.plane_radius;
    int best = 0x7FFFFFFF;
    const uint32_t n = tail >> 16;
    // lowest LDS address of the band's slots (rad == 2: five slots of 16 bytes from here)
    const uint32_t bandlo = kSide ? rowaddr + 16u * (uint32_t)t1 : rowaddr - 16u * (uint32_t)t1 - 32u * (uint32_t)rad;
#define ML_ADDR(c) (kSide ? rowaddr + (c) : rowaddr - (c))
#define ML_TRIP4(cc, i)                                                                                       \
    {                                                                                                         \
        const uint32_t c0 = (cc).x & 0xFFFFu, c1 = (cc).x >> 16, c2 = (cc).y & 0xFFFFu, c3 = (cc).y >> 16;    \
        const uint4 o0 = lds_read16(ML_ADDR(c0)), o1 = lds_read16(ML_ADDR(c1));                               \
        const uint4 o2 = lds_read16(ML_ADDR(c2)), o3 = lds_read16(ML_ADDR(c3));                               \
        best = min3_op(best, sad_hi16(own, o0, (i)), sad_hi16(own, o1, (i) + 1u));                            \
        best = min3_op(best, sad_hi16(own, o2, (i) + 2u), sad_hi16(own, o3, (i) + 3u));                       \
    }
    // (exclusion: a candidate whose slot lies among the band's, i.e. whose address is bandlo .. bandlo + 32 rad)
#define ML_TRIP4X(cc, i)                                                                                      \
    {                                                                                                         \
        const uint32_t c0 = (cc).x & 0xFFFFu, c1 = (cc).x >> 16, c2 = (cc).y & 0xFFFFu, c3 = (cc).y >> 16;    \
        const uint32_t a0 = ML_ADDR(c0), a1 = ML_ADDR(c1), a2 = ML_ADDR(c2), a3 = ML_ADDR(c3);                \
        const uint4 o0 = lds_read16(a0), o1 = lds_read16(a1), o2 = lds_read16(a2), o3 = lds_read16(a3);       \
        const int k0 = a0 - bandlo <= blen ? 0x7FFFFFFF : sad_hi16(own, o0, (i));                             \
        const int k1 = a1 - bandlo <= blen ? 0x7FFFFFFF : sad_hi16(own, o1, (i) + 1u);                        \
        const int k2 = a2 - bandlo <= blen ? 0x7FFFFFFF : sad_hi16(own, o2, (i) + 2u);                        \
        const int k3 = a3 - bandlo <= blen ? 0x7FFFFFFF : sad_hi16(own, o3, (i) + 3u);                        \
        best = min3_op(best, k0, k1);                                                                         \
        best = min3_op(best, k2, k3);                                                                         \
    }
    // (edge: a wave beside the image border, where some lane's largest candidate warps out of the row -- round 5 sent
    // such waves to the checked form, 684 operations per pixel, 2.9 % of the kernel for 1.2 % of its pixels: the
    // exclusion trip with a second range test, the slot's address against the row's columns 2 .. W-3)
#define ML_TRIP4E(cc, i)                                                                                      \
    {                                                                                                         \
        const uint32_t c0 = (cc).x & 0xFFFFu, c1 = (cc).x >> 16, c2 = (cc).y & 0xFFFFu, c3 = (cc).y >> 16;    \
        const uint32_t a0 = ML_ADDR(c0), a1 = ML_ADDR(c1), a2 = ML_ADDR(c2), a3 = ML_ADDR(c3);                \
        const uint4 o0 = lds_read16(a0), o1 = lds_read16(a1), o2 = lds_read16(a2), o3 = lds_read16(a3);       \
        const int k0 = a0 - bandlo <= blen || a0 - rowlo > rowspan ? 0x7FFFFFFF : sad_hi16(own, o0, (i));     \
        const int k1 = a1 - bandlo <= blen || a1 - rowlo > rowspan ? 0x7FFFFFFF : sad_hi16(own, o1, (i) + 1u); \
        const int k2 = a2 - bandlo <= blen || a2 - rowlo > rowspan ? 0x7FFFFFFF : sad_hi16(own, o2, (i) + 2u); \
        const int k3 = a3 - bandlo <= blen || a3 - rowlo > rowspan ? 0x7FFFFFFF : sad_hi16(own, o3, (i) + 3u); \
        best = min3_op(best, k0, k1);                                                                         \
        best = min3_op(best, k2, k3);                                                                         \
    }
#if SVH_ML_PROBE != 5
    if (n != 0) {
        uint2 cb;      // (ca: the record's first four candidates, requested a pixel ahead by the kernel)
        if (edge) {
            const uint32_t blen = 32u * (uint32_t)rad, rowspan = 16u * (uint32_t)(P.W - 5);
#pragma unroll 1
            for (uint32_t i = 0; __builtin_amdgcn_ballot_w64(i < n) != 0; i += 4) {
                ML_CNT(12, 1);
                ML_TRIP4E(ca, i)
                ca = lds_read8(lrec + 2 * i + 8);
            }
        } else if (!excl) {
            for (uint32_t i = 0;; i += 8) {
                cb = lds_read8(lrec + 2 * i + 8);
                ML_CNT(7, 1);
                ML_TRIP4(ca, i)
                __builtin_amdgcn_sched_barrier(0);   // (the second trip's reads stay behind the first trip's keys: registers)
                if (__builtin_amdgcn_ballot_w64(i + 4 < n) == 0) break;
                ca = lds_read8(lrec + 2 * i + 16);
                ML_CNT(7, 1);
                ML_TRIP4(cb, i + 4u)
                __builtin_amdgcn_sched_barrier(0);
                if (__builtin_amdgcn_ballot_w64(i + 8 < n) == 0) break;
            }
        } else {
            const uint32_t blen = 32u * (uint32_t)rad;
            for (uint32_t i = 0;; i += 8) {
                cb = lds_read8(lrec + 2 * i + 8);
                ML_CNT(9, 1);
                ML_TRIP4X(ca, i)
                __builtin_amdgcn_sched_barrier(0);
                if (__builtin_amdgcn_ballot_w64(i + 4 < n) == 0) break;
                ca = lds_read8(lrec + 2 * i + 16);
                ML_CNT(9, 1);
                ML_TRIP4X(cb, i + 4u)
                __builtin_amdgcn_sched_barrier(0);
                if (__builtin_amdgcn_ballot_w64(i + 8 < n) == 0) break;
            }
        }
    }
#endif
#undef ML_TRIP4
#undef ML_TRIP4X
#undef ML_TRIP4E
#undef ML_ADDR
#if SVH_ML_PROBE == 4
    if (best == 0x7FFFFFFF) return -1;
    return (int)(lds_read2(lrec + 2 * (best & 0xFFFF)) >> 4);
#endif
    // ---- the plane band with its prior; rank = 512 + d
    if (rad == 2) {
        // the presets' radius: five slots at constant offsets from one address, key starts = t1 + a scalar
        const uint4 o0 = lds_read16(bandlo), o1 = lds_read16(bandlo + 16u), o2 = lds_read16(bandlo + 32u);
        const uint4 o3 = lds_read16(bandlo + 48u), o4 = lds_read16(bandlo + 64u);
        int key0, key1, key2, key3, key4;
        if (!excl) {      // (every live lane's plane is valid)
            key0 = sad_hi16(own, o0, (uint32_t)t1 + K.key[0]);
            key1 = sad_hi16(own, o1, (uint32_t)t1 + K.key[1]);
            key2 = sad_hi16(own, o2, (uint32_t)t1 + K.key[2]);
            key3 = sad_hi16(own, o3, (uint32_t)t1 + K.key[3]);
            key4 = sad_hi16(own, o4, (uint32_t)t1 + K.key[4]);
        } else {
            const uint32_t p0 = valid ? K.p16[0] : 0u, p1 = valid ? K.p16[1] : 0u, p2 = valid ? K.p16[2] : 0u;
            key0 = sad_hi16(own, o0, (uint32_t)t1 + K.key0[0] + p2);
            key1 = sad_hi16(own, o1, (uint32_t)t1 + K.key0[1] + p1);
            key2 = sad_hi16(own, o2, (uint32_t)t1 + K.key0[2] + p0);
            key3 = sad_hi16(own, o3, (uint32_t)t1 + K.key0[3] + p1);
            key4 = sad_hi16(own, o4, (uint32_t)t1 + K.key0[4] + p2);
        }
        if (!band_ok) {
            // slot j holds disparity dj = kSide ? t1 + j : t1 + 4 - j, warped column u +- dj
            const uint32_t dm = (uint32_t)P.disp_max, wm = (uint32_t)(P.W - 4);
#define ML_SLOT_OK(j) ((uint32_t)(kSide ? t1 + (j) : t1 + 4 - (j)) <= dm &&                                   \
                       (uint32_t)(kSide ? u + t1 + (j) - 2 : u - t1 - 4 + (j) - 2) < wm)
            key0 = ML_SLOT_OK(0) ? key0 : 0x7FFFFFFF;
            key1 = ML_SLOT_OK(1) ? key1 : 0x7FFFFFFF;
            key2 = ML_SLOT_OK(2) ? key2 : 0x7FFFFFFF;
            key3 = ML_SLOT_OK(3) ? key3 : 0x7FFFFFFF;
            key4 = ML_SLOT_OK(4) ? key4 : 0x7FFFFFFF;
#undef ML_SLOT_OK
        }
        best = min3_op(best, key0, key1);
        best = min3_op(best, key2, key3);
        best = best < key4 ? best : key4;
    } else {
        // (other radii: the caller sends waves with a clipped band to the checked form)
        const int nb = 2 * rad + 1;
        const uint32_t a0 = kSide ? bandlo : bandlo + 32u * (uint32_t)rad;   // slot of d_plane - rad
        const uint32_t rk0 = (uint32_t)(512 + t1);
        for (int k = 0; k < nb; k += 2) {
            const int k1 = k + 1 < nb ? k + 1 : k;
            const uint4 o0 = lds_read16(kSide ? a0 + 16u * k : a0 - 16u * k);
            const uint4 o1 = lds_read16(kSide ? a0 + 16u * k1 : a0 - 16u * k1);
            const int key0 = sad_hi16(own, o0, rk0 + (uint32_t)k + (valid ? (uint32_t)s_band[k] : 0u));
            const int key1 = sad_hi16(own, o1, rk0 + (uint32_t)k1 + (valid ? (uint32_t)s_band[k1] : 0u));
            best = min3_op(best, key0, key1);
        }
    }
    if (best == 0x7FFFFFFF) return -1;
    const int rank = best & 0xFFFF;
    return rank >= 512 ? rank - 512 : (int)(lds_read2(lrec + 2 * rank) >> 4);
}

// One pixel of a wave through the hot form: the plan (record, plane disparity, the tests that decide the form -- on every
// lane, live or not, so that the votes are taken outside the divergent part; lanes that are not live hold a valid
// record address and plane 0), the votes, the scan.  *is_cold (wave-uniform): the wave has to take the checked form.
//  * the ballot of ONE comparison is that comparison's mask, and combining masks is scalar work; the ballot of a
//    conjunction makes hipcc turn the combined mask into a lane value and compare it again (round 5: 6 operations);
//  * cell test: 16 dmax <= 16 u - 32 (left map) / <= 16 (W - 3) - 16 u (right map), and at most ML_FAST candidates;
//  * band test: (unsigned)(d_plane - rad) < clamp(u - 2 rad - 1 | W - 2 - 2 rad - u, 0, disp_max - 2 rad + 1).
template <int kSide>
__device__ __forceinline__ int ml_pixel(const uint4& own, const float4& pl, int u, float vf, bool live, uint64_t m_live,
                                        uint32_t lrec, uint32_t tail, uint2 ca, uint32_t oth_base, int neg_prior,
                                        const FastK& K, const int* s_band, const MatchParams& P, bool* is_cold) {
    const int rad = P.plane_radius;
    const uint32_t u16 = (uint32_t)u * 16u;
    const uint32_t rowaddr = oth_base + u16;
    const int d_plane = (int)__fadd_rn(__fadd_rn(__fmul_rn(pl.x, (float)u), __fmul_rn(pl.y, vf)), pl.z);
    const int t1 = d_plane - rad;
    const uint32_t lim = kSide ? (uint32_t)K.cell_c - u16 : u16 - (uint32_t)K.cell_c;
    const uint64_t m_warp = __builtin_amdgcn_ballot_w64((tail & 0xFFFFu) > lim);          // a candidate leaves the row
    const uint64_t m_many = __builtin_amdgcn_ballot_w64(tail >= ((uint32_t)(ML_FAST + 1) << 16));
    int bl = kSide ? K.band_c - u : u - K.band_c;
    bl = bl < K.band_m ? bl : K.band_m;
    bl = bl > 0 ? bl : 0;
    const bool band_ok = (m_live & __builtin_amdgcn_ballot_w64((uint32_t)t1 >= (uint32_t)bl)) == 0;
    const bool edge = (m_live & m_warp) != 0;
    *is_cold = (m_live & m_many) != 0 || (rad != 2 && (!band_ok || edge));
    if (*is_cold) {
        ML_CNT(6, 1);
        return -10;
    }
    // (an edge wave takes the exclusion forms throughout: its band keys are masked one by one)
    const bool excl = edge || !neg_prior || (m_live & __builtin_amdgcn_ballot_w64(__float_as_int(pl.w) == 0)) != 0;
    ML_CNT(3, 1); ML_CNT(4, excl ? 1 : 0); ML_CNT(5, band_ok ? 0 : 1); ML_CNT(13, edge ? 1 : 0);
    ML_CNT(11, __builtin_popcountll(m_live));
    int res = -10;
#if SVH_ML_PROBE == 3
    if (live) res = (d_plane ^ (int)tail) + (excl ? 1 : 0) + (band_ok ? 2 : 0) + (int)(rowaddr & 1u);
#else
    if (live) res = ml_fast<kSide>(own, rowaddr, lrec, tail, t1, u, __float_as_int(pl.w) != 0, excl, band_ok && !edge, edge,
                                   oth_base + 32u, ca, K, s_band, P);
#endif
    return res;
}

// checked form: per-candidate range and band tests (waves next to the image border, invalid planes,
// non-negative priors, cells with more candidates than a record).  rank = d (cell), 512 + d (band)
template <int kSide>
__device__ __forceinline__ int ml_pixel_checked(const uint4& own, const PixelPlan& q, int u, uint32_t rowaddr,
                                                uint32_t lrec, const int* s_band, const MatchParams& P,
                                                const uint32_t* __restrict__ bits) {
    const uint32_t blen = (uint32_t)(q.dhi - q.dlo);
    const bool has_band = q.dhi >= q.dlo;
    const int dlo = q.dlo, rad = P.plane_radius;
    int best = 0x7FFFFFFF;
    if (__builtin_amdgcn_ballot_w64(q.n > ML_FAST) == 0) {
        for (int i = 0; __builtin_amdgcn_ballot_w64(i < q.n) != 0; i += 2) {
            if (i >= q.n) continue;
            const uint32_t cc = lds_read4(lrec + 2 * i);
            const uint32_t c0 = cc & 0xFFFFu, c1 = cc >> 16;
            const int d0 = (int)(c0 >> 4), d1 = (int)(c1 >> 4);
            const int u0 = kSide ? u + d0 : u - d0, u1 = kSide ? u + d1 : u - d1;
            const bool ok0 = (uint32_t)(u0 - 2) < (uint32_t)(P.W - 4) && !(has_band && (uint32_t)(d0 - dlo) <= blen);
            const bool ok1 = (uint32_t)(u1 - 2) < (uint32_t)(P.W - 4) && !(has_band && (uint32_t)(d1 - dlo) <= blen);
            const uint4 o0 = lds_read16(ok0 ? (kSide ? rowaddr + c0 : rowaddr - c0) : rowaddr);
            const uint4 o1 = lds_read16(ok1 ? (kSide ? rowaddr + c1 : rowaddr - c1) : rowaddr);
            const int k0 = ok0 ? sad_hi16(own, o0, (uint32_t)d0) : 0x7FFFFFFF;
            const int k1 = ok1 ? sad_hi16(own, o1, (uint32_t)d1) : 0x7FFFFFFF;
            best = min3i(best, k0, k1);
        }
    } else {
        // a cell of the wave holds more candidates than a record: every lane decodes its cell's bit set
#pragma unroll 1
        for (int w = 0; w < P.gwords; w++) {
            uint32_t b = bits[w];
            while (b) {
                const int dc = w * 32 + __builtin_ctz(b);
                b &= b - 1;
                const int uw = kSide ? u + dc : u - dc;
                if ((uint32_t)(uw - 2) >= (uint32_t)(P.W - 4) || (has_band && (uint32_t)(dc - dlo) <= blen)) continue;
                const uint4 o = lds_read16(kSide ? rowaddr + 16u * dc : rowaddr - 16u * dc);
                const int key = sad_hi16(own, o, (uint32_t)dc);
                best = key < best ? key : best;
            }
        }
    }
    for (int dc = q.dlo; dc <= q.dhi; dc++) {
        const int uw = kSide ? u + dc : u - dc;
        const bool ok = (uint32_t)(uw - 2) < (uint32_t)(P.W - 4);
        const uint4 o = lds_read16(ok ? (kSide ? rowaddr + 16u * dc : rowaddr - 16u * dc) : rowaddr);
        int dd = dc - q.d_plane;
        dd = dd < 0 ? -dd : dd;
        const uint32_t pr = q.valid ? (uint32_t)s_band[rad + dd] : 0u;
        const int key = sad_hi16(own, o, (uint32_t)(512 + dc) + pr);
        best = ok && key < best ? key : best;
    }
    return best != 0x7FFFFFFF ? (best & 511) : -1;
}

template <bool kLr, int kIters>
__global__ __launch_bounds__(kIters <= 5 ? 768 : 512) __attribute__((amdgpu_waves_per_eu(kIters <= 5 ? 6 : 4, 8))) void k_match_list(GroupDev G, MatchParams P, MatchList Q, DevMaps out,
                                                    int write_raw, float lr_threshold) {
    extern __shared__ uint4 s_dyn[];   // rows [2][W] | cell records [2][gw][ML_CAP] u16 | raw [2][Wr] int16
    __shared__ int s_band[32];         // P[|k - radius|] << 16
    __shared__ int s_neg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = (int)(blockDim.x >> 6);
    // XCD-aware row order (Q.xcd): workgroups go to the 8 XCDs round-robin by id, and the blocks of neighbouring image
    // rows read the same Sobel lines (4 of 5), the same cell records (20 rows per cell row) and neighbouring owner
    // rows.  XCD k therefore takes the k-th eighth of the launch's rows, in order: what neighbours share is served by
    // that XCD's L2.
    int row_id = blockIdx.x;
    if (Q.xcd) {
        const int total = P.DH * P.npairs, chunk = (total + 7) >> 3;
        row_id = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if (row_id >= total) return;
    }
    // subsampling (round 5): the map row y is image row v = 2 y, the map column x image column u = 2 x; descriptors,
    // cells, planes and ownership stay in image coordinates (elas.cpp:1098-1113)
    const int mul = P.sub ? 2 : 1;
    const int pair = row_id / P.DH, y = row_id - pair * P.DH, v = y * mul;
    if (!G.hdr->active[pair]) return;
    const size_t N = (size_t)P.W * P.H;
    uint4* s_rows = s_dyn;
    uint16_t* s_rec = reinterpret_cast<uint16_t*>(s_dyn + 2 * Q.Ws);
    int16_t* s_raw = reinterpret_cast<int16_t*>(s_rec + 2 * P.gw * ML_CAP);
    int line = v < P.H - 3 ? v : P.H - 3;
    line = line > 2 ? line : 2;
    const int cr = v / P.grid_size, cells = P.gw * P.gh;
    {
        if (G.desc_fly) {
            // The two descriptor rows are assembled from the Sobel planes, four adjacent pixels per task.  A lane
            // stores its four 16-byte slots one after the other, and a ds_write_b128 is served in groups of 8 lanes
            // over 32 banks: with consecutive tasks on consecutive lanes all 8 lanes of a group hit one of two slots
            // modulo 8 (4-way conflicts on every store: round 4's 1.4 M conflict cycles per 4-pair launch).  Round 5:
            // even lanes take the left image, odd lanes the right one, and the second row starts at a slot == 2
            // (mod 4), so a group's 8 stores fall on four different slots modulo 8 (2-way: 16 LDS cycles against
            // the 13 the store's data transfer takes anyway).  No data is moved between lanes.
            // Round 6: the row stride Ws covers the whole last task (Ws >= 4 nq), so the four stores are unconditional;
            // the loads go through one uniform base + a 32-bit lane offset; the two border rows whose descriptors are
            // all zero are a branch of the block, the columns outside 3 .. W-4 a branch of the three tasks they touch.
            const int nq = (P.W + 3) >> 2;
            const uint8_t* pbase = G.desc + (size_t)(2 * pair) * N * 16;
            const uint32_t pitch = (uint32_t)fly_pitch(P.W), vplane = (uint32_t)P.H * pitch;
            const uint32_t imstride = (uint32_t)(N * 16), off0 = (uint32_t)line * pitch + 8u;
            if (line < 3 || line >= P.H - 3) {
                for (int j = tid; j < 2 * Q.Ws; j += (int)blockDim.x) s_rows[j] = make_uint4(0, 0, 0, 0);
            } else {
                for (int task = tid; task < 2 * nq; task += (int)blockDim.x) {
                    const int im = task & 1, x = 4 * (task >> 1);
                    uint4 o[4];
                    fly_desc4_off(pbase, (im ? imstride : 0u) + off0 + (uint32_t)x, pitch, vplane, o);
                    if (x == 0 || x + 4 > P.W - 3) {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (x + i < 3 || x + i >= P.W - 3) o[i] = make_uint4(0, 0, 0, 0);
                    }
                    uint4* dst = s_rows + im * Q.Ws + x;
#pragma unroll
                    for (int i = 0; i < 4; i++) dst[i] = o[i];
                }
            }
        } else {
            const uint4* l1 = reinterpret_cast<const uint4*>(G.desc) + (size_t)(2 * pair) * N + (size_t)line * P.W;
            dma_slots(s_rows, l1, P.W, wave, nwaves, lane);
            dma_slots(s_rows + Q.Ws, l1 + N, P.W, wave, nwaves, lane);
        }
        const int rq = P.gw * ML_CAP * 2 / 16;   // uint4 per side
        const uint4* r1 = reinterpret_cast<const uint4*>(G.lists + ((size_t)(2 * pair) * cells + (size_t)cr * P.gw) * ML_CAP);
        const uint4* r2 = reinterpret_cast<const uint4*>(G.lists + ((size_t)(2 * pair + 1) * cells + (size_t)cr * P.gw) * ML_CAP);
        dma_slots(reinterpret_cast<uint4*>(s_rec), r1, rq, wave, nwaves, lane);
        dma_slots(reinterpret_cast<uint4*>(s_rec) + rq, r2, rq, wave, nwaves, lane);
    }
    if (tid < 32) {
        const int dd = tid - P.plane_radius;
        s_band[tid] = tid <= 2 * P.plane_radius ? (int)((uint32_t)G.P[dd < 0 ? -dd : dd] << 16) : 0;
    }
    if (tid == 64) {
        int neg = 1;
        for (int dd = 0; dd <= P.plane_radius; dd++) neg &= G.P[dd] < 0 ? 1 : 0;
        s_neg = neg;
    }
    const int half = blockDim.x >> 1;
    const int side = wave >= (half >> 6) ? 1 : 0;   // a SCALAR: half is a multiple of 64 (the constants of FastK depend on it)
    const int z = 2 * pair + side;
    const int x0 = tid - side * half;
    const int32_t* own_t = G.owner + (size_t)z * N + (size_t)v * P.W;
    // owner words of all of the thread's pixels, then (after the barrier drained them) their planes.  Round 6: k_owner
    // leaves the two columns at either end of a row alone (findMatch returns there, elas.cpp:797-798), so a stale word
    // says "no triangle" and the column test is gone from this loop.
    int tk[kIters];
#pragma unroll
    for (int k = 0; k < kIters; k++) {
        const int u = (x0 + k * half) * mul;
        tk[k] = own_t[u < P.W ? u : P.W - 1];
    }
    const int tri0 = z ? G.hdr->tri_end[z - 1] : 0;
    __syncthreads();
#if SVH_ML_PROBE == 1
    if (tid == 0) G.Draw[(size_t)z * P.DW * P.DH + (size_t)y * P.DW] = (float)(s_rows[5].x + tk[0]);
    return;
#endif
    const int own1 = G.owner_base + 1;
    // plane records through one scalar base + a 32-bit lane offset; the plane of the owning triangle is requested one
    // pixel ahead (all of them at once would cost a sixth wave per SIMD)
    const char* rbase = reinterpret_cast<const char*>(G.raster + tri0);
    auto plane_of = [rbase](int t) {
        const uint32_t off = (uint32_t)(t > 0 ? t : 0) * (uint32_t)sizeof(TriRaster);
        return *reinterpret_cast<const float4*>(rbase + (size_t)off);
    };
    float4 pl_next = plane_of(tk[0] - own1);
    const int neg_prior = s_neg, rad = P.plane_radius;
    // scalar constants of the hot form (see FastK)
    FastK K;
    {
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane(s_band[rad]);
        const uint32_t p1 = (uint32_t)__builtin_amdgcn_readfirstlane(s_band[rad + 1]);
        const uint32_t p2 = (uint32_t)__builtin_amdgcn_readfirstlane(s_band[rad + 2]);
        K.p16[0] = p0; K.p16[1] = p1; K.p16[2] = p2;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            K.key0[j] = 512u + (uint32_t)(side ? j : 4 - j);
            K.key[j] = K.key0[j] + (j == 2 ? p0 : (j == 1 || j == 3) ? p1 : p2);
        }
        K.cell_c = side ? 16 * (P.W - 3) : 32;
        K.band_c = side ? P.W - 2 - 2 * rad : 2 * rad + 1;
        K.band_m = P.disp_max - 2 * rad + 1;
        K.band_m = K.band_m > 0 ? K.band_m : 0;
    }
    const float vf = (float)v;
    const uint32_t rows_addr = lds_addr_of(s_rows), rec_addr = lds_addr_of(s_rec) + (uint32_t)(side * P.gw * ML_CAP * 2);
    const uint32_t own_base = rows_addr + (uint32_t)(side * Q.Ws) * 16u, oth_base = rows_addr + (uint32_t)((1 - side) * Q.Ws) * 16u;
    float* out_row = G.Draw + (size_t)z * P.DW * P.DH + (size_t)y * P.DW;
    int16_t* raw_row = s_raw + side * Q.Wr;
    uint32_t cold = 0;     // pixels (bit k) of this wave that need the checked form
    // Round 6: what a pixel's scan starts from -- its own descriptor, the address of its cell's record, the record's
    // last word (count, largest candidate) and first four candidates -- is requested ONE PIXEL AHEAD, like the plane: these
    // were three dependent LDS round trips at the head of every pixel (descriptor -> texture vote -> record word -> plan
    // -> first candidates), with nothing of the wave's own to cover them.  (Columns beyond the row read LDS bytes that
    // the x < DW test then never looks at.)
    struct Head { uint4 own; uint32_t lrec, tail; uint2 ca; };
    static_assert(ML_CAP * 2 == 64, "record size of the shift below");
    auto head_of = [&](int k) {
        Head h;
        const uint32_t u = (uint32_t)((x0 + k * half) * mul);
        h.own = lds_read16(own_base + u * 16u);
        const uint32_t cell = __umulhi(u, P.grid_magic);
        asm("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(h.lrec) : "v"(cell), "s"(rec_addr));   // + cell * 2 ML_CAP
        h.tail = lds_read4(h.lrec + 2 * (ML_CAP - 2));   // [30] last candidate, [31] count
        h.ca = lds_read8(h.lrec);
        return h;
    };
    Head hd_next = head_of(0);
#pragma unroll
    for (int k = 0; k < kIters; k++) {
        const int x = x0 + k * half, u = x * mul;
        const float4 pl = pl_next;
        const Head hd = hd_next;
        if (k + 1 < kIters) {
            pl_next = plane_of(tk[k + 1] - own1);
            hd_next = head_of(k + 1);
            asm volatile("" ::: "memory");   // the requests go out HERE (hipcc would sink them to their uses, a pixel later)
        }
        if (x < P.DW) {
            int res = -10;
            const uint4 own = hd.own;
            // (votes: the ballot of ONE comparison is that comparison's mask -- see ml_pixel)
            const bool owned = tk[k] >= own1, textured = (int)texture16(own) >= P.match_texture;
            const bool live = owned && textured;
            const uint64_t m_live = __builtin_amdgcn_ballot_w64(owned) & __builtin_amdgcn_ballot_w64(textured);
            ML_CNT(0, 1);
#if SVH_ML_PROBE == 2
            res = live ? 0 : -10;
            if (false) {
#else
            if (m_live != 0) {
#endif
                ML_CNT(1, 1); ML_CNT(2, __builtin_popcountll(m_live));
                bool is_cold;
                res = side ? ml_pixel<1>(own, pl, u, vf, live, m_live, hd.lrec, hd.tail, hd.ca, oth_base, neg_prior, K, s_band, P, &is_cold)
                           : ml_pixel<0>(own, pl, u, vf, live, m_live, hd.lrec, hd.tail, hd.ca, oth_base, neg_prior, K, s_band, P, &is_cold);
                if (is_cold) cold |= 1u << k;
            }
            if (!kLr || write_raw) out_row[x] = (float)res;
            if (kLr) raw_row[x] = (int16_t)res;
        }
    }
    // the waves that could not take the fast form redo those pixels (everything reloaded: this is the cold path)
    cold = (uint32_t)__builtin_amdgcn_readfirstlane((int)cold);   // (uniform already: set under a wave-wide vote)
#if SVH_ML_PROBE >= 2 && SVH_ML_PROBE <= 5 || SVH_ML_PROBE == 8
    cold = 0;
#endif
#pragma unroll 1
    for (int k = 0; cold >> k; k++) {
        if (!((cold >> k) & 1)) continue;
        const int x = x0 + k * half, u = x * mul;
        if (x >= P.DW) continue;
        const int t = own_t[u] - G.owner_base - 1;   // (u < W: x < DW; columns 0, 1, W-2, W-1 are never owned)
        const uint4 own = lds_read16(own_base + (uint32_t)(u < P.W ? u : 0) * 16u);
        if (!(t >= 0 && (int)texture16(own) >= P.match_texture)) continue;
        const float4 pl = *reinterpret_cast<const float4*>(G.raster + (tri0 + t));
        const uint32_t c = __umulhi((uint32_t)u, P.grid_magic);
        const uint32_t lrec = rec_addr + c * (ML_CAP * 2);
        const uint32_t rowaddr = oth_base + (uint32_t)u * 16u;
        const uint32_t* bits = G.mask + ((size_t)z * cells + (size_t)cr * P.gw + c) * P.gwords;
        const int res = side ? ml_pixel_checked<1>(own, ml_plan<1>(pl, u, v, lrec, P), u, rowaddr, lrec, s_band, P, bits)
                             : ml_pixel_checked<0>(own, ml_plan<0>(pl, u, v, lrec, P), u, rowaddr, lrec, s_band, P, bits);
        if (!kLr || write_raw) out_row[x] = (float)res;
        if (kLr) raw_row[x] = (int16_t)res;
    }
    if (!kLr) return;
#if SVH_ML_PROBE == 6
    return;
#endif
    __syncthreads();
    // E12: keep d iff the other map, at the warped position, agrees within lr_threshold
    float* D = out.D[side] + (size_t)pair * out.stride[side] + (size_t)y * P.DW;
    const int16_t* other = s_raw + (1 - side) * Q.Wr;
    if (P.sub) {
        for (int x = x0; x < P.DW; x += half) {
            const int d = raw_row[x];
            float o = -10.f;
            // the reference's float form (elas.cpp:1150-1175): the warped column is x -+ d / 2
            const float fd = (float)d, step = fd / 2, uwf = side ? (float)x + step : (float)x - step;
            if (d >= 0 && uwf >= 0 && uwf < (float)P.DW)
                if (!(fabsf((float)other[(int)uwf] - fd) > lr_threshold)) o = fd;
            D[x] = o;
        }
        return;
    }
    // Full resolution (round 6): everything is an integer -- |other - d| <= lr_threshold  <=>  |other - d| <= floor(thr)
    // (no pixel passes a negative threshold), one unsigned comparison; the other map's value is read without a branch
    // (an index outside the row reads LDS bytes that the range test then discards).
    const int ithr = lr_threshold >= 0.f ? (int)fminf(floorf(lr_threshold), 1048576.f) : -1;
    const uint32_t span = ithr >= 0 ? 2u * (uint32_t)ithr + 1u : 0u;
    const uint32_t raw_a = lds_addr_of(raw_row), oth_a = lds_addr_of(other);
    char* Db = reinterpret_cast<char*>(D);     // scalar base + 32-bit lane offset
#define ML_LR_ROW(WARP)                                                                                        \
    _Pragma("unroll") for (int k = 0; k < kIters; k++) {                                                       \
        const int x = x0 + k * half;                                                                           \
        if (x < P.DW) {                                                                                        \
            const int d = (int)(int16_t)lds_read2(raw_a + 2u * (uint32_t)x);                                   \
            const int uw = WARP;                                                                               \
            const int o = (int)(int16_t)lds_read2(oth_a + 2u * (uint32_t)uw);                                  \
            const bool keep = d >= 0 && (uint32_t)uw < (uint32_t)P.DW && (uint32_t)(o - d + ithr) < span;      \
            *reinterpret_cast<float*>(Db + (size_t)(4u * (uint32_t)x)) = keep ? (float)d : -10.f;              \
        }                                                                                                      \
    }
    if (side) { ML_LR_ROW(x + d) } else { ML_LR_ROW(x - d) }
#undef ML_LR_ROW
}

// ---------------------------------------------------------------------------
// E12  Elas::leftRightConsistencyCheck   libelas/src/elas.cpp:1122-1204
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lr(GroupDev G, DevMaps out, int DW, int DH, int sub,
                                            float lr_threshold) {
    const int pair = blockIdx.z;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= DW || y >= DH) return;
    if (!G.hdr->active[pair]) return;   // outputs of a failed pair stay untouched
    const size_t DN = (size_t)DW * DH;
    const float* R1 = G.Draw + (size_t)(2 * pair) * DN;
    const float* R2 = R1 + DN;
    float* D1 = out.D[0] + (size_t)pair * out.stride[0];
    float* D2 = out.D[1] + (size_t)pair * out.stride[1];
    const size_t rowo = (size_t)y * DW;
    const float d1 = R1[rowo + x], d2 = R2[rowo + x];
    const float fx = (float)x;
    const float uw1 = sub ? fx - d1 / 2 : fx - d1;
    const float uw2 = sub ? fx + d2 / 2 : fx + d2;
    float o1 = -10.f, o2 = -10.f;
    if (d1 >= 0 && uw1 >= 0 && uw1 < (float)DW)
        if (!(fabsf(R2[rowo + (int)uw1] - d1) > lr_threshold)) o1 = d1;
    if (d2 >= 0 && uw2 >= 0 && uw2 < (float)DW)
        if (!(fabsf(R1[rowo + (int)uw2] - d2) > lr_threshold)) o2 = d2;
    D1[rowo + x] = o1;
    D2[rowo + x] = o2;
}

// map of post-processing slot z = pair*nside + side
__device__ __forceinline__ float* post_map(const DevMaps& m, int z, int nside, int* pair) {
    const int p = z / nside, s = z - p * nside;
    *pair = p;
    return m.D[s] + (size_t)p * m.stride[s];
}

// ---------------------------------------------------------------------------
// E13  Elas::removeSmallSegments   libelas/src/elas.cpp:1208-1326
// The flood fill joins 4-neighbours that are both valid and differ by at most
// speckle_sim_threshold: a symmetric relation, so segments are the connected
// components of that graph and a parallel union-find gives the same sets.
//
//   k_seg_tile   : per 64x16 tile, union-find in LDS; label = tile-local root
//   k_seg_border : global unions only across tile borders (path-halving finds)
//   k_seg_sum    : local sizes of merged components added to their root
//   k_seg_mask   : pixels of components below speckle_size become -10
// ---------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int32_t* L, int x) {
    int p = L[x];
    while (p != x) {
        x = p;
        p = L[x];
    }
    return x;
}

// find with path halving; stale or lost shortcuts are harmless because every
// value ever stored in L[x] is an ancestor of x
__device__ __forceinline__ int uf_find_halve(int32_t* L, int x) {
    int p = L[x];
    while (p != x) {
        const int gp = L[p];
        if (gp != p) L[x] = gp;
        x = p;
        p = gp;
    }
    return x;
}

__device__ __forceinline__ void uf_union(int32_t* L, int a, int b) {
    for (;;) {
        a = uf_find_halve(L, a);
        b = uf_find_halve(L, b);
        if (a == b) return;
        if (a > b) {
            int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[b], a);  // hang the larger root under the smaller
        if (old == b) return;
        b = old;
    }
}

__device__ __forceinline__ bool seg_joined(float a, float b, float thr) {
    return a >= 0 && b >= 0 && fabsf(a - b) <= thr;
}

// Two-level labelling.  Level 1 (k_seg_tile): every 64x16 tile is labelled by a
// union-find that lives entirely in LDS (LDS atomics, ~50 ns hops instead of
// ~0.7 us through L2); a pixel's global label is the index of its tile-local
// root and RL[root] holds the local component size.  Level 2 (k_seg_border):
// only pixels on tile borders issue global unions, between tile-local roots, so
// the global forest has one node per local component and is at most #tiles
// deep.  k_seg_sum adds the local sizes of merged components into their root.
constexpr int CX = 64, CY = 16;

// (relaxed workgroup-scope loads: re-read on every hop like a volatile access, but they stay
// ds_read_b32 -- a volatile int* loses the LDS address space and compiles to flat loads)
__device__ __forceinline__ int lds_find(int* L, int x) {
    int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return x;
}

__device__ __forceinline__ void lds_union(int* L, int a, int b) {
    for (;;) {
        a = lds_find(L, a);
        b = lds_find(L, b);
        if (a == b) return;
        if (a > b) {
            int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[b], a);
        if (old == b) return;
        b = old;
    }
}

// XCD-aware tile order of the tile kernels (k_seg_tile, k_gap_tile, k_mean_tile).  Workgroups go to the 8 XCDs
// round-robin by id, and neighbouring tiles read each other's halo (and share 128-byte lines along their common
// border): launched as a 1-D grid of xcd_blocks(tiles) workgroups, XCD k takes the k-th eighth of the tiles in
// row-major order -- whole maps when the launch has 8 or more -- so the halo is served by that XCD's L2.
// nz == 0: the plain 3-D launch (blockIdx = tile x, tile y, map).
struct TileIdx {
    int x, y, z, gx, gy;
};
__device__ __forceinline__ bool tile_index(int tw, int th, int DW, int DH, int nz, TileIdx* t) {
    t->gx = (DW + tw - 1) / tw;
    t->gy = (DH + th - 1) / th;
    if (nz == 0) {
        t->x = blockIdx.x; t->y = blockIdx.y; t->z = blockIdx.z;
        return true;
    }
    const int per = t->gx * t->gy, total = per * nz, chunk = (total + 7) >> 3;
    const int lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (lin >= total) return false;
    t->z = lin / per;
    const int r = lin - t->z * per;
    t->y = r / t->gx;
    t->x = r - t->y * t->gx;
    return true;
}
inline unsigned xcd_blocks(int total) { return (unsigned)(((total + 7) >> 3) << 3); }

__global__ __launch_bounds__(256) void k_seg_tile(GroupDev G, DevMaps m, PostScratch S, int nside,
                                                  int DW, int DH, float thr, int nz) {
    __shared__ float sD[CY][CX];
    __shared__ int sL[CX * CY];
    __shared__ int sC[CX * CY];
    __shared__ int s_nr;           // tile-local roots listed so far
    TileIdx tb;
    if (!tile_index(CX, CY, DW, DH, nz, &tb)) return;   // (block-uniform, before any barrier)
    int pair;
    const float* D = post_map(m, tb.z, nside, &pair);
    if (!G.hdr->active[pair]) return;
    if (threadIdx.y == 0 && threadIdx.x == 0) s_nr = 0;   // (three barriers before its first use)
    const size_t zo = (size_t)tb.z * DW * DH;
    const int x0 = tb.x * CX, y0 = tb.y * CY;
    const int tx = threadIdx.x;   // lane: a wave owns whole tile rows ty = threadIdx.y + 4k
    // step 1: horizontal runs per row with one ballot; label = first pixel of the run
    int len[CY / 4];
    float dk[CY / 4];
#pragma unroll
    for (int k = 0; k < CY / 4; k++) {   // the four rows of this thread: all loads in flight before the first use
        const int gx = x0 + tx, gy = y0 + (int)threadIdx.y + 4 * k;
        dk[k] = (gx < DW && gy < DH) ? D[gy * DW + gx] : -10.f;
    }
#pragma unroll
    for (int k = 0; k < CY / 4; k++) {
        const int ty = threadIdx.y + 4 * k;
        const float d = dk[k];
        const bool valid = d >= 0;
        const float dl = __shfl_up(d, 1, kWave);
        const bool start = valid && (tx == 0 || !seg_joined(d, dl, thr));
        const unsigned long long starts = __ballot(start);
        const unsigned long long invalid = __ballot(!valid);
        int label = -1;
        len[k] = 0;
        if (valid) {
            const unsigned long long upto = starts & (~0ull >> (63 - tx));
            label = ty * CX + (63 - __clzll((long long)upto));
            if (start) {
                const unsigned long long stops = ((starts | invalid) >> tx) >> 1;
                len[k] = stops ? __ffsll((long long)stops) : (kWave - tx);
            }
        }
        sD[ty][tx] = d;
        sL[ty * CX + tx] = label;
        sC[ty * CX + tx] = 0;
    }
    __syncthreads();
    // step 2: unions between runs of adjacent rows (skipped when the pixel to the
    // left already joins the same two runs)
#pragma unroll
    for (int k = 0; k < CY / 4; k++) {
        const int ty = threadIdx.y + 4 * k, i = ty * CX + tx;
        if (ty == 0) continue;
        const float d = sD[ty][tx], du = sD[ty - 1][tx];
        if (!seg_joined(d, du, thr)) continue;
        if (tx > 0) {
            const float dl = sD[ty][tx - 1], dul = sD[ty - 1][tx - 1];
            if (seg_joined(d, dl, thr) && seg_joined(dl, dul, thr) && seg_joined(du, dul, thr)) continue;
        }
        lds_union(sL, sL[i], sL[i - CX]);
    }
    __syncthreads();
    // step 3: run starts find their root and add their length to it
#pragma unroll
    for (int k = 0; k < CY / 4; k++) {
        const int ty = threadIdx.y + 4 * k, i = ty * CX + tx;
        if (len[k] > 0) {
            const int r = lds_find(sL, i);
            atomicAdd(&sC[r], len[k]);
            len[k] = r;               // remembered for the compression below
        } else {
            len[k] = -1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CY / 4; k++)
        if (len[k] >= 0) sL[(threadIdx.y + 4 * k) * CX + tx] = len[k];   // start -> root
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CY / 4; k++) {
        const int ty = threadIdx.y + 4 * k, i = ty * CX + tx;
        const int gx = x0 + tx, gy = y0 + ty;
        if (gx >= DW || gy >= DH) continue;
        const int gi = gy * DW + gx;
        int label = -1, size = 0;
        const int s = sL[i];
        if (s >= 0) {
            const int r = sL[s];      // pixel -> run start -> root (or start == root)
            const int root = sL[r] == r ? r : sL[r];
            const int ry = root / CX, rx = root - ry * CX;
            label = (y0 + ry) * DW + (x0 + rx);
            if (root == i) size = sC[i];   // this pixel is the tile-local root
        }
        S.labels[zo + gi] = label;
        // The tile-local roots are listed per tile (in `tmp`, which is free until k_gap_tile; a tile's
        // list starts where its pixels would start if the map were stored tile by tile, so it can hold
        // every pixel of the tile): k_seg_sum visits a few roots per tile instead of scanning the counts
        // of every pixel.  Positions from an LDS counter, the tile's count is written once.
        if (size > 0) {
            // the size is stored at the tile-local roots only: counts are never read anywhere else (labels
            // point at roots), so what an earlier group left at the other pixels does not matter
            S.counts[zo + gi] = size;
            const int bh = min(CY, DH - y0);                    // rows of this band of tiles
            reinterpret_cast<int32_t*>(S.tmp)[zo + (size_t)y0 * DW + (size_t)x0 * bh + atomicAdd(&s_nr, 1)] = gi;
        }
    }
    __syncthreads();
    if (threadIdx.y == 0 && tx == 0)
        S.nroots[((size_t)tb.z * tb.gy + tb.y) * tb.gx + tb.x] = s_nr;
}

// unions across tile borders: one thread per pixel of a tile's first row / column
__global__ __launch_bounds__(256) void k_seg_border(GroupDev G, DevMaps m, PostScratch S, int nside,
                                                    int DW, int DH, float thr, int nhor) {
    int pair;
    const float* D = post_map(m, blockIdx.y, nside, &pair);
    if (!G.hdr->active[pair]) return;
    int32_t* L = S.labels + (size_t)blockIdx.y * DW * DH;
    const int e = blockIdx.x * 256 + threadIdx.x;
    int x, y, step;  // neighbour across the border is (x,y) - step
    if (e < nhor) {             // horizontal borders: rows CY, 2CY, ...
        const int b = e / DW;
        x = e - b * DW;
        y = (b + 1) * CY;
        step = DW;
        if (y >= DH) return;
    } else {                    // vertical borders: columns CX, 2CX, ...
        const int e2 = e - nhor;
        const int nvb = (DW - 1) / CX;      // number of interior vertical borders
        if (nvb <= 0) return;
        const int b = e2 % nvb;
        y = e2 / nvb;
        x = (b + 1) * CX;
        step = 1;
        if (y >= DH || x >= DW) return;
    }
    const int i = y * DW + x;
    const float d = D[i], q = D[i - step];
    if (!seg_joined(d, q, thr)) return;
    // skip when the previous pixel along the border already joins the same two components
    const int along = step == 1 ? DW : 1;
    const bool has_prev = step == 1 ? (y % CY != 0) : (x % CX != 0);
    if (has_prev) {
        const float dp = D[i - along], qp = D[i - step - along];
        if (seg_joined(d, dp, thr) && seg_joined(q, qp, thr) && seg_joined(dp, qp, thr)) return;
    }
    uf_union(L, L[i], L[i - step]);
}

// add the local size of every merged tile-local component to its global root: one wave per tile of
// k_seg_tile walks that tile's list of roots
__global__ __launch_bounds__(256) void k_seg_sum(GroupDev G, PostScratch S, int nside, int DW, int DH,
                                                 int min_size) {
    if (!G.hdr->active[blockIdx.y / nside]) return;
    const int n = DW * DH;
    const size_t zo = (size_t)blockIdx.y * n;
    const int tcols = (DW + CX - 1) / CX, trows = (DH + CY - 1) / CY;
    const int tile = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= tcols * trows) return;
    const int ty = tile / tcols, tx = tile - ty * tcols;
    const int y0 = ty * CY, x0 = tx * CX, bh = min(CY, DH - y0);
    const int nroots = S.nroots[(size_t)blockIdx.y * tcols * trows + tile];
    const int32_t* list = reinterpret_cast<const int32_t*>(S.tmp) + zo + (size_t)y0 * DW + (size_t)x0 * bh;
    int32_t* L = S.labels + zo;
    for (int k = lane; k < nroots; k += 64) {
        // a tile-local root that is not a global root is never added to, so its entry still is the
        // tile-local size written by k_seg_tile (the seam merges of k_seg_border changed labels only)
        const int i = list[k];
        const int size = S.counts[zo + i];
        const int root = uf_find(L, i);
        if (root != i) {
            L[i] = root;
            int32_t* c = S.counts + zo + root;
            // only "below speckle_size or not" is ever asked of a count
            if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < min_size)
                atomicAdd(c, size);
        }
    }
}

__global__ __launch_bounds__(256) void k_seg_mask(GroupDev G, DevMaps m, PostScratch S, int nside,
                                                  int n, int min_size) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int pair;
    float* D = post_map(m, blockIdx.y, nside, &pair);
    if (!G.hdr->active[pair]) return;
    const size_t zo = (size_t)blockIdx.y * n;
    const int s = S.labels[zo + i];
    if (s < 0) {
        // an invalid pixel is a segment of one pixel (elas.cpp:1244-1317)
        if (1 < min_size) D[i] = -10.f;
    } else if (S.counts[zo + S.labels[zo + s]] < min_size) {
        D[i] = -10.f;
    }
}

// ---------------------------------------------------------------------------
// E14  Elas::gapInterpolation   libelas/src/elas.cpp:1330-1530
// A run of invalid pixels is filled iff its length is in [1,gap] and it has a
// valid pixel on both sides inside the line.  Both ends are original valid
// pixels, so every line is a pure function of its input: the row pass and the
// column pass are two out-of-place per-pixel kernels that look at most `gap`
// pixels each way.  (Sequential per-line kernels below handle large gaps and the
// add_corners extrapolation.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float gap_value(float d1, float d2) {
    return fabsf(d1 - d2) < 3.0f ? (d1 + d2) / 2 : (d1 < d2 ? d1 : d2);  // min(d1,d2): std::min
}

template <bool kCols>
__global__ __launch_bounds__(256) void k_gap_local(GroupDev G, DevMaps m, PostScratch S, int nside,
                                                   int DW, int DH, int gap) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= DW || y >= DH) return;
    int pair;
    float* D = post_map(m, blockIdx.z, nside, &pair);
    if (!G.hdr->active[pair]) return;
    float* tmp = S.tmp + (size_t)blockIdx.z * DW * DH;
    const float* in = kCols ? tmp : D;     // rows: D -> tmp, columns: tmp -> D
    float* out = kCols ? D : tmp;
    const int i = y * DW + x;
    const int stride = kCols ? DW : 1;
    const int pos = kCols ? y : x, len = kCols ? DH : DW;
    float val = in[i];
    if (!(val >= 0)) {
        if (gap <= 4) {
            // the common small gap widths: all neighbours are fetched at once (independent loads)
            // and the nearest valid one on each side is picked afterwards
            float nl[4], nr[4];
#pragma unroll
            for (int k = 1; k <= 4; k++) {
                nl[k - 1] = (k <= gap && pos - k >= 0) ? in[i - k * stride] : -1.f;
                nr[k - 1] = (k <= gap && pos + k < len) ? in[i + k * stride] : -1.f;
            }
            int l = 0, r = 0;
            float vl = 0.f, vr = 0.f;
#pragma unroll
            for (int k = 4; k >= 1; k--) {
                if (nl[k - 1] >= 0) { l = k; vl = nl[k - 1]; }
                if (nr[k - 1] >= 0) { r = k; vr = nr[k - 1]; }
            }
            if (l && r && r <= gap - l + 1) val = gap_value(vl, vr);
        } else {
            int l = 0, r = 0;
            for (int k = 1; k <= gap && pos - k >= 0; k++)
                if (in[i - k * stride] >= 0) {
                    l = k;
                    break;
                }
            if (l) {
                for (int k = 1; k <= gap - l + 1 && pos + k < len; k++)
                    if (in[i + k * stride] >= 0) {
                        r = k;
                        break;
                    }
                if (r) val = gap_value(in[i - l * stride], in[i + r * stride]);
            }
        }
    }
    out[i] = val;
}

// general per-line version (any gap width, add_corners extrapolation), in place
template <bool kCols>
__global__ void k_gap_lines(GroupDev G, DevMaps m, int nside, int DW, int DH, int gap,
                            int add_corners) {
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    const int nlines = kCols ? DW : DH;
    if (line >= nlines) return;
    int pair;
    float* D = post_map(m, blockIdx.y, nside, &pair);
    if (!G.hdr->active[pair]) return;
    const int len = kCols ? DH : DW;
    const int stride = kCols ? DW : 1;
    float* base = D + (kCols ? line : (size_t)line * DW);
    int count = 0;
    for (int p = 0; p < len; p++) {
        if (base[(size_t)p * stride] >= 0) {
            if (count >= 1 && count <= gap) {
                const int first = p - count, last = p - 1;
                if (first > 0 && last < len - 1) {
                    const float di = gap_value(base[(size_t)(first - 1) * stride],
                                               base[(size_t)(last + 1) * stride]);
                    for (int q = first; q <= last; q++) base[(size_t)q * stride] = di;
                }
            }
            count = 0;
        } else {
            count++;
        }
    }
    if (add_corners) {
        for (int p = 0; p < len; p++)
            if (base[(size_t)p * stride] >= 0) {
                const float val = base[(size_t)p * stride];
                for (int q = (p - gap > 0 ? p - gap : 0); q < p; q++) base[(size_t)q * stride] = val;
                break;
            }
        for (int p = len - 1; p >= 0; p--)
            if (base[(size_t)p * stride] >= 0) {
                const float val = base[(size_t)p * stride];
                const int qe = p + gap < len - 1 ? p + gap : len - 1;
                for (int q = p; q <= qe; q++) base[(size_t)q * stride] = val;
                break;
            }
    }
}

// Parallel forms of the general version (round 5; MIDDLEBURY: ipol_gap_width 5000 + add_corners, where the
// one-thread-per-line walk above was 21 % of the kernel time: a thread walking along a ROW reads 4 bytes per 64-byte
// line, a thread walking down a COLUMN waits for one dependent load after the other).  A run of invalid pixels is
// filled from the two valid pixels that bound it, and those are ORIGINAL values (the reference's scan never writes a
// valid pixel and reads only the pixel before a run and the one that ends it, elas.cpp:1352-1399 / 1446-1491), so
// every pixel can look up its bounds independently: L = nearest valid index before it, R = nearest valid index after
// it, fill iff both exist and R - L - 1 <= gap.  The extrapolation (add_corners, :1401-1436 / :1493-1528) concerns
// only the pixels before the first / after the last valid pixel of the line, which the interior fill never touches.
//
// rows: one wave per row.  The row is staged in LDS in chunks of 64; the valid mask of a chunk is a ballot, L / R
// inside the chunk are bit scans of that mask, across chunks two short uniform loops (last valid index before /
// first valid index after each chunk).
constexpr int kGapChunks = 48;   // rows up to 3072 px (4 rows x 12 KB + the tables below stay under the 64 KB a launch gets without opting in); wider rows take k_gap_lines
__global__ __launch_bounds__(256) void k_gap_rows_scan(GroupDev G, DevMaps m, int nside, int DW, int DH, int gap,
                                                       int add_corners) {
    extern __shared__ float s_gapline[];                 // [4][nch * 64]
    __shared__ unsigned long long s_mask[4][kGapChunks];
    __shared__ int s_prev[4][kGapChunks], s_next[4][kGapChunks];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row = blockIdx.x * 4 + wave;
    if (row >= DH) return;                               // (wave-uniform; the kernel has no block barrier)
    int pair;
    float* D = post_map(m, blockIdx.y, nside, &pair);
    if (!G.hdr->active[pair]) return;
    float* base = D + (size_t)row * DW;
    const int nch = (DW + 63) >> 6;
    float* line = s_gapline + wave * (nch * 64);
    constexpr int kNone = 1 << 30;
    int run = -1;
    for (int k = 0; k < nch; k++) {
        const int p = 64 * k + lane;
        const float v = p < DW ? base[p] : -1.f;
        line[p] = v;
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(v >= 0);
        s_mask[wave][k] = mk;                            // (every lane stores the same value)
        s_prev[wave][k] = run;
        if (mk) run = 64 * k + 63 - __builtin_clzll(mk);
    }
    const int pl = run;                                  // last valid pixel of the row, -1 = none
    int runn = kNone;
    for (int k = nch - 1; k >= 0; k--) {
        const unsigned long long mk = s_mask[wave][k];
        s_next[wave][k] = runn;
        if (mk) runn = 64 * k + __builtin_ctzll(mk);
    }
    const int pf = runn;                                 // first valid pixel, kNone = none
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < nch; k++) {
        const int p = 64 * k + lane;
        const float v = line[p];
        if (p >= DW || v >= 0) continue;
        const unsigned long long mk = s_mask[wave][k];
        const unsigned long long lower = mk & ((1ull << lane) - 1ull), upper = mk >> lane;
        const int L = lower ? 64 * k + 63 - __builtin_clzll(lower) : s_prev[wave][k];
        const int R = upper ? p + __builtin_ctzll(upper) : s_next[wave][k];
        float o = v;
        if (L >= 0 && R != kNone) {
            if (R - L - 1 <= gap) o = gap_value(line[L], line[R]);
        } else if (add_corners) {
            if (L < 0 && R != kNone && p >= pf - gap) o = line[pf];
            if (L >= 0 && R == kNone && p <= pl + gap) o = line[pl];
        }
        if (o >= 0) base[p] = o;
    }
}

// columns: 64 columns x kGapSeg row segments per block.  Every thread summarises its segment (first / last valid
// pixel), the summaries meet in LDS, and each thread then fills its own segment: runs that end inside it from the
// valid pixel that ends them, the run that reaches the end of the segment from the next segment's first valid pixel.
constexpr int kGapSeg = 8;
__global__ __launch_bounds__(64 * kGapSeg) void k_gap_cols_seg(GroupDev G, DevMaps m, int nside, int DW, int DH,
                                                              int gap, int add_corners) {
    __shared__ int s_first[kGapSeg][64], s_last[kGapSeg][64];
    __shared__ float s_fval[kGapSeg][64], s_lval[kGapSeg][64];
    const int tx = threadIdx.x, seg = threadIdx.y;
    const int col = blockIdx.x * 64 + tx;
    int pair;
    float* D = post_map(m, blockIdx.y, nside, &pair);
    if (!G.hdr->active[pair]) return;                    // (block-uniform)
    const bool on = col < DW;
    const int seg_len = (DH + kGapSeg - 1) / kGapSeg;
    const int r0 = seg * seg_len, r1 = DH < r0 + seg_len ? DH : r0 + seg_len;
    float* base = D + (on ? col : 0);
    constexpr int kB = 8;                                // loads in flight per thread
    int fi = -1, li = -1;
    float fv = 0.f, lv = 0.f;
    if (on) {
        for (int p0 = r0; p0 < r1; p0 += kB) {
            float v[kB];
#pragma unroll
            for (int j = 0; j < kB; j++) v[j] = p0 + j < r1 ? base[(size_t)(p0 + j) * DW] : -1.f;
#pragma unroll
            for (int j = 0; j < kB; j++)
                if (v[j] >= 0) {
                    if (fi < 0) { fi = p0 + j; fv = v[j]; }
                    li = p0 + j;
                    lv = v[j];
                }
        }
    }
    s_first[seg][tx] = fi; s_fval[seg][tx] = fv;
    s_last[seg][tx] = li;  s_lval[seg][tx] = lv;
    __syncthreads();
    if (!on) return;
    int Li = -1, Ri = -1, pf = -1, pl = -1;
    float Lv = 0.f, Rv = 0.f, pfv = 0.f, plv = 0.f;
    for (int s2 = 0; s2 < kGapSeg; s2++) {
        const int f2 = s_first[s2][tx], l2 = s_last[s2][tx];
        if (f2 < 0) continue;
        if (pf < 0) { pf = f2; pfv = s_fval[s2][tx]; }
        pl = l2; plv = s_lval[s2][tx];
        if (s2 < seg) { Li = l2; Lv = s_lval[s2][tx]; }
        if (s2 > seg && Ri < 0) { Ri = f2; Rv = s_fval[s2][tx]; }
    }
    int last = Li;
    float lastv = Lv;
    if (fi >= 0) {
        for (int p0 = r0; p0 < r1; p0 += kB) {
            float v[kB];
#pragma unroll
            for (int j = 0; j < kB; j++) v[j] = p0 + j < r1 ? base[(size_t)(p0 + j) * DW] : -1.f;
#pragma unroll
            for (int j = 0; j < kB; j++) {
                const int p = p0 + j;
                if (v[j] >= 0) {
                    const int count = p - last - 1;
                    if (last >= 0 && count >= 1 && count <= gap) {
                        const float di = gap_value(lastv, v[j]);
                        for (int q = last + 1 > r0 ? last + 1 : r0; q < p; q++) base[(size_t)q * DW] = di;
                    }
                    last = p;
                    lastv = v[j];
                }
            }
        }
    }
    // the run that reaches the end of the segment (the whole segment when it holds no valid pixel)
    if (last < r1 - 1 && last >= 0 && Ri >= 0 && Ri - last - 1 <= gap) {
        const float di = gap_value(lastv, Rv);
        for (int q = last + 1 > r0 ? last + 1 : r0; q < r1; q++) base[(size_t)q * DW] = di;
    }
    if (add_corners && pf >= 0) {
        int q0 = pf - gap > 0 ? pf - gap : 0, q1 = pf;                     // [q0, q1): the first valid value upwards
        q0 = q0 > r0 ? q0 : r0;
        q1 = q1 < r1 ? q1 : r1;
        for (int q = q0; q < q1; q++) base[(size_t)q * DW] = pfv;
        int e0 = pl + 1, e1 = (pl + gap < DH - 1 ? pl + gap : DH - 1) + 1; // [e0, e1): the last valid value downwards
        e0 = e0 > r0 ? e0 : r0;
        e1 = e1 < r1 ? e1 : r1;
        for (int q = e0; q < e1; q++) base[(size_t)q * DW] = plv;
    }
}

// ---------------------------------------------------------------------------
// E15  Elas::adaptiveMean   libelas/src/elas.cpp:1535-1754
// weight = max(0, 4 - float_from_bits(bits(val - centre) & 0x4F000000)): the
// reference's "abs mask" is an int->float conversion of 0x7FFFFFFF
// (elas.cpp:1571).  Lanes of the SSE registers are ring slots (position % taps);
// the 8-tap branch first adds slot j and j+4, then sums ((l0+l1)+l2)+l3.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float am_weight(float val, float centre) {
    const float mm = __uint_as_float(__float_as_uint(__fsub_rn(val, centre)) & 0x4F000000u);
    const float w = __fsub_rn(4.0f, mm);
    return w > 0.0f ? w : 0.0f;
}

template <bool kCols, int kTaps>
__global__ __launch_bounds__(256) void k_adaptive_mean(GroupDev G, DevMaps m, PostScratch S,
                                                       int nside, int DW, int DH) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= DW || y >= DH) return;
    int pair;
    float* D = post_map(m, blockIdx.z, nside, &pair);
    if (!G.hdr->active[pair]) return;
    float* tmp = S.tmp + (size_t)blockIdx.z * DW * DH;
    // horizontal: in = D (negatives read as -10), out = tmp
    // vertical  : in = tmp, value kept where the filter does not fire = D, out = D
    const float* in = kCols ? tmp : D;
    float* out = kCols ? D : tmp;
    const int i = y * DW + x;
    constexpr int back = kTaps == 8 ? 3 : 1;      // centre = newest - back
    constexpr int lead = kTaps - 1;
    const int stride = kCols ? DW : 1;
    const int pos = kCols ? y : x, len = kCols ? DH : DW;
    const int other = kCols ? x : y, olen = kCols ? DW : DH;
    float res = D[i];
    // D_tmp (elas.cpp:1548-1560) is malloc'ed, set to -10 where D is invalid and otherwise left
    // UNINITIALISED until the horizontal pass writes it.  Both allocations are fresh multi-MB
    // blocks (zero pages), and the pinned reference runs with zero-filled allocations: a valid
    // pixel the horizontal pass does not write reads back as 0 in the vertical pass.  That only
    // matters when valid pixels reach the 3 border rows/columns (add_corners).
    if (!kCols) res = res < 0 ? -10.f : 0.f;
    // lines 3..olen-4; centres lead-back .. len-1-back
    if (other >= 3 && other < olen - 3 && pos >= lead - back && pos <= len - 1 - back) {
        const int first = pos + back - lead;  // oldest tap position
        float centre = in[i];
        if (!kCols && centre < 0) centre = -10.f;
        // ring slot s holds the tap whose position % kTaps == s (elas.cpp:1674, 1722): load the
        // taps in slot order directly -- tap index k = (s - first) mod kTaps -- instead of
        // loading them in position order and rotating with a select chain
        float ws[kTaps], fs[kTaps];
#pragma unroll
        for (int s = 0; s < kTaps; s++) {
            const int k = (s - first) & (kTaps - 1);   // kTaps is 4 or 8
            float t = in[i + (first + k - pos) * stride];
            if (!kCols && t < 0) t = -10.f;
            const float w = am_weight(t, centre);
            ws[s] = w;
            fs[s] = __fmul_rn(t, w);
        }
        float wl[4], fl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (kTaps == 8) {
                wl[j] = __fadd_rn(ws[j], ws[j + kTaps / 2]);
                fl[j] = __fadd_rn(fs[j], fs[j + kTaps / 2]);
            } else {
                wl[j] = ws[j];
                fl[j] = fs[j];
            }
        }
        const float wsum = __fadd_rn(__fadd_rn(__fadd_rn(wl[0], wl[1]), wl[2]), wl[3]);
        const float fsum = __fadd_rn(__fadd_rn(__fadd_rn(fl[0], fl[1]), fl[2]), fl[3]);
        if (wsum > 0) {
            const float dv = __fdiv_rn(fsum, wsum);
            if (dv >= 0) res = dv;
        }
    }
    out[i] = res;
}

// ---------------------------------------------------------------------------
// E16  Elas::median   libelas/src/elas.cpp:1758-1838 (separable 7-tap)
// horizontal: D -> tmp (tmp is calloc'ed in the reference: 0 outside the
// interior); vertical: gates on D, reads tmp, writes D.
// ---------------------------------------------------------------------------
template <bool kCols>
__global__ __launch_bounds__(256) void k_median(GroupDev G, DevMaps m, PostScratch S, int nside,
                                                int DW, int DH) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= DW || y >= DH) return;
    int pair;
    float* D = post_map(m, blockIdx.z, nside, &pair);
    if (!G.hdr->active[pair]) return;
    float* tmp = S.tmp + (size_t)blockIdx.z * DW * DH;
    const float* in = kCols ? tmp : D;
    float* out = kCols ? D : tmp;
    const int i = y * DW + x;
    const bool interior = x >= 3 && x < DW - 3 && y >= 3 && y < DH - 3;
    const float gate = D[i];
    float res = kCols ? gate : 0.f;
    if (interior) {
        if (gate >= 0) {
            const int stride = kCols ? DW : 1;
            float v[7];
#pragma unroll
            for (int k = 0; k < 7; k++) v[k] = in[i + (k - 3) * stride];
            // the reference insertion-sorts the window and takes element 3 (elas.cpp:1790-1800, 1818-1828): the
            // fourth smallest VALUE, which a 13-exchange selection network yields without data-dependent loops
            // (disparity maps hold no NaN and no -0: equal values are interchangeable)
#define SVH_CX(a, b) { const float lo_ = fminf(v[a], v[b]), hi_ = fmaxf(v[a], v[b]); v[a] = lo_; v[b] = hi_; }
            SVH_CX(0, 5) SVH_CX(0, 3) SVH_CX(1, 6) SVH_CX(2, 4) SVH_CX(0, 1) SVH_CX(3, 5) SVH_CX(2, 6)
            SVH_CX(2, 3) SVH_CX(3, 6) SVH_CX(4, 5) SVH_CX(1, 4) SVH_CX(1, 3) SVH_CX(3, 4)
#undef SVH_CX
            res = v[3];
        } else {
            res = gate;
        }
    }
    out[i] = res;
}


inline dim3 grid2d(int w, int h, int z = 1) { return dim3((w + 63) / 64, (h + 3) / 4, z); }

// brackets one kernel launch with the context's timer, if any
struct Timed {
    Profiler* p;
    Timed(const LaunchCtx& cx, const char* name) : p(cx.prof) {
        if (p) p->begin(name);
    }
    ~Timed() {
        if (p) p->end();
    }
};

#define LAUNCH(name, kernel, grid, block, ...)                                                  \
    do {                                                                                        \
        Timed timed_(cx, name);                                                                 \
        hipLaunchKernelGGL(kernel, grid, block, 0, (hipStream_t)cx.stream, __VA_ARGS__);        \
    } while (0)

}  // namespace

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_descriptor(const LaunchCtx& cx, const DevImages& img, int32_t g, int32_t W, int32_t H,
                       int32_t half, uint8_t* desc, bool fly) {
    if (fly) {
        // E1 only: the Sobel planes; the matchers assemble the descriptors they stage (see k_sobel_planes)
        const int strips = (W + 4 * 62 - 1) / (4 * 62);
        const int cols = 4 * ((W + 4 * strips - 1) / (4 * strips));
        const int segs = (H + SP_ROWS - 1) / SP_ROWS, nwaves = strips * segs;
        LAUNCH("k_descriptor", k_sobel_planes, dim3((nwaves + 3) / 4, 2 * g), dim3(256), img, W, H, strips, cols,
               nwaves, desc);
        return;
    }
    // 26 rows per wave, 16 loads in flight: shorter walks (more waves) measured marginally better
    // than 42 .. 90 rows, all within 2 %
    constexpr int rows = 26, chunk = 16;
    static_assert((rows + 6) % chunk == 0, "the walk is a whole number of load chunks");
    const int strips = (W + DS_COLS - 1) / DS_COLS, segs = (H + rows - 1) / rows, nwaves = strips * segs;
    LAUNCH("k_descriptor", (k_descriptor_stream<rows, chunk>), dim3((nwaves + 3) / 4, 2 * g), dim3(256), img, W, H,
           half, strips, nwaves, desc);
}

// candidates per block of the LDS-staged support kernel, 0 when the strips do not fit (generic kernel)
static int support_strip(const svh_elas_params& p, const Dims& d, size_t* lds_out) {
    static const int sb_env = svh::env("SVH_SUPPORT_SB") ? atoi(svh::env("SVH_SUPPORT_SB")) : 64;
    constexpr size_t kSupportStatic = 512;
    auto strip_lds = [&](int sb) {
        const int span = (sb - 1) * d.step;
        const size_t wl = (size_t)std::min(d.W, span + 2 * p.disp_max + 5);
        const size_t wr = (size_t)std::min(d.W, span + p.disp_max + 5);
        return 2 * (size_t)(support_stride((int)wl) + support_stride((int)wr)) * sizeof(uint4);
    };
    int sb = sb_env == 32 ? 32 : 64;
    if (sb == 64 && strip_lds(64) + 2 * (size_t)d.W + 16 + kSupportStatic > 64 * 1024) sb = 32;
    // (+ the right strip's texture values of the fly form: 2 bytes per column, counted always)
    const size_t lds = strip_lds(sb) + 2 * (size_t)std::min(d.W, (sb - 1) * d.step + p.disp_max + 5) + 16;
    if (lds_out) *lds_out = lds;
    return lds + kSupportStatic <= 64 * 1024 ? sb : 0;
}

void launch_support(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                    const uint8_t* desc, int16_t* dcan, bool fly) {
    SupportParams P;
    P.W = d.W; P.H = d.H; P.Wc = d.Wc; P.Hc = d.Hc; P.step = d.step; P.npairs = g;
    P.disp_min = p.disp_min; P.disp_max = p.disp_max;
    P.support_texture = p.support_texture; P.lr_threshold = p.lr_threshold;
    P.support_threshold = p.support_threshold;
    // LDS need of the staged kernel: two rows of both strips (16 B slots).  32 candidates per
    // 512-thread block was the best of the block shapes tried (16/256 ... 128/1024: all within 3 %).
    // candidates per 512-thread block: 64 (two forward rounds).  32: 341 vs 316 us per isolated 32-pair
    // launch.  A whole lattice row per 1024-thread block (79 KB of LDS) is faster alone (304 us) and slower
    // in the pipeline (29.1 vs 29.5 k pairs/s): two such blocks take a CU's LDS away from everything else.
    // strip by fit: 64 candidates per block, else 32 (large candidate_stepsize or disp_max), else the generic
    // kernel; the kernel's static LDS (s_fwd, s_todo, s_ntodo) counts against the 64 KB limit
    size_t lds = 0;
    const int sb = support_strip(p, d, &lds);
    if (sb) {
        Timed timed_(cx, "k_support");
        // (pairs * lattice rows rounded up to 8) * chunks blocks: XCD-aware order, see the kernel
        const int chunks = (d.Wc + sb - 1) / sb;
        const dim3 grid((unsigned)(((d.Hc * g + 7) / 8) * 8 * chunks), 1, 1);
        hipStream_t st = (hipStream_t)cx.stream;
        if (sb == 32 && fly) hipLaunchKernelGGL((k_support_lds<32, 512, true>), grid, dim3(512), lds, st, desc, dcan, P);
        else if (sb == 32) hipLaunchKernelGGL((k_support_lds<32, 512, false>), grid, dim3(512), lds, st, desc, dcan, P);
        else if (fly) hipLaunchKernelGGL((k_support_lds<64, 512, true>), grid, dim3(512), lds, st, desc, dcan, P);
        else hipLaunchKernelGGL((k_support_lds<64, 512, false>), grid, dim3(512), lds, st, desc, dcan, P);
    } else {
        const int cands = d.Wc * d.Hc;
        LAUNCH("k_support", k_support, dim3((cands + 3) / 4, g), dim3(256), desc, dcan, P);
    }
}

void launch_prior(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                  int32_t total_sup, int32_t total_tri, const GroupDev& G) {
    hipStream_t s = (hipStream_t)cx.stream;
    const int cells = d.gw * d.gh;
    const size_t words = (size_t)2 * g * cells * d.gwords;
    (void)hipMemsetAsync(G.seed, 0, words * sizeof(uint32_t), s);
    // device-built header: the launches are sized for a typical count (a lattice at ~15 % density)
    // and stride over the rest
    const int sup_bound = std::max(256, g * d.Wc * d.Hc / 6), tri_bound = 4 * sup_bound;
    const int nt = total_tri >= 0 ? total_tri : tri_bound, ns = total_sup >= 0 ? total_sup : sup_bound;
    if (nt > 0) LAUNCH("k_prior", k_prior, dim3((2 * nt + 255) / 256), dim3(256), G, total_tri);
    if (ns > 0)
        LAUNCH("k_grid_seed", k_grid_seed, dim3((ns + 255) / 256), dim3(256), G, total_sup,
               d.gw, d.gh, d.gwords, p.grid_size, p.disp_max);
    LAUNCH("k_grid_dilate", k_grid_dilate, dim3((unsigned)((words + 255) / 256)), dim3(256), G,
           2 * g, d.gw, d.gh, d.gwords);
    if (d.gwords <= 8 && G.lists)
        LAUNCH("k_grid_list", k_grid_list, dim3((unsigned)((2 * g * cells + 3) / 4)), dim3(256), G, 2 * g * cells,
               d.gwords);
}

void launch_owner(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                  int32_t total_tri, const GroupDev& G) {
    // no clearing: the engine hands every group a fresh owner_base above all values stored so far
    if (total_tri == 0) return;
    const int nt = total_tri >= 0 ? total_tri : 4 * std::max(256, g * d.Wc * d.Hc / 6);   // see launch_prior
    // The span-ends form of the fix pass relies on two float evaluations a*u + b of one triangle edge
    // differing by less than one row (see k_owner).  |a*u| can reach H * (W + 2*disp_max) for an edge that
    // climbs the whole image within one column; beyond 2^22 one ulp of that product is half a row and the
    // margin is gone, so such geometries (not 1920x1080 at disp_max 255: 2.6e6) take the exhaustive pass.
    const bool wide = (double)d.H * ((double)d.W + 2.0 * p.disp_max) > 4194304.0;
    const int fix_all = svh::env("SVH_OWNER_FIX_ALL") ? atoi(svh::env("SVH_OWNER_FIX_ALL")) : (wide ? 1 : 0);   // read per launch: tests toggle it
    const dim3 go(xcd_blocks((nt + 3) / 4));   // (a multiple of 8: see the kernel's block order)
    LAUNCH("k_owner", k_owner<false>, go, dim3(256), G, total_tri, d.W, d.H,
           p.subsampling, 0);
    LAUNCH("k_owner_fix", k_owner<true>, go, dim3(256), G, total_tri, d.W, d.H,
           p.subsampling, fix_all);
}

// Does the dense matcher take its list form (k_match_list) for these parameters?  Used by launch_match and -- before
// the descriptor stage -- by the engine, which lets E1 write the Sobel planes only when BOTH matchers stage their
// descriptor rows themselves (descriptors_on_the_fly).
static bool match_list_usable(const svh_elas_params& p, const Dims& d, int32_t prior_absmax, int32_t plane_radius,
                              bool have_lists, MatchList* Qout, size_t* lds_out) {
    static const bool ordered = svh::env("SVH_MATCH_ORDERED") != nullptr;
    const bool keyed_ok = !ordered && prior_absmax < (1 << 19) && p.disp_max < 512 && plane_radius <= 15 &&
                          d.W < 65536 && p.grid_size > 1;
    static const bool list_off = svh::env("SVH_MATCH_LIST") && atoi(svh::env("SVH_MATCH_LIST")) == 0;
    static const bool list_sub = !(svh::env("SVH_MATCH_LIST_SUB") && atoi(svh::env("SVH_MATCH_LIST_SUB")) == 0);
    if (!(keyed_ok && !list_off && (!p.subsampling || list_sub) && d.gwords <= 8 && have_lists && prior_absmax < 28000 &&
          d.DW <= 8 * 256))
        return false;
    MatchList Q;
    Q.Wr = (d.W + 7) / 8 * 8;
    Q.Ws = ((d.W + 3) & ~3) + 2;   // covers the last staging task of four slots, == 2 (mod 4)
    // threads per side: 256; rows of 1281 .. 1920 px, whose LDS footprint lets two blocks share a CU, take 384 (round
    // 5: 2 x 12 waves = 6 per SIMD with the 5-pixel instance instead of 2 x 8 = 4 per SIMD with the 8-pixel one)
    static const bool wide768 = !(svh::env("SVH_MATCH_WIDE768") && atoi(svh::env("SVH_MATCH_WIDE768")) == 0);
    int iters = (d.DW + 255) / 256;
    Q.half = std::min(256, ((d.DW + iters - 1) / iters + 63) / 64 * 64);
    if (iters > 5 && wide768 && d.DW <= 5 * 384) {
        iters = (d.DW + 383) / 384;
        Q.half = std::min(384, ((d.DW + iters - 1) / iters + 63) / 64 * 64);
    }
    Q.kIters = iters <= 5 ? 5 : 8;
    static const bool xcd_rows = !(svh::env("SVH_MATCH_XCD") && atoi(svh::env("SVH_MATCH_XCD")) == 0);
    Q.xcd = xcd_rows ? 1 : 0;
    const size_t ldsl = (size_t)2 * Q.Ws * sizeof(uint4) + (size_t)2 * d.gw * ML_CAP * sizeof(uint16_t) +
                        (size_t)2 * Q.Wr * sizeof(int16_t);
    constexpr size_t kListStatic = 256;   // s_band, s_neg
    bool ok = ldsl + kListStatic <= 160 * 1024;
    if (ok && ldsl + kListStatic > 64 * 1024) {
        ok = hipFuncSetAttribute((const void*)k_match_list<true, 5>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - (int)kListStatic) == hipSuccess &&
             hipFuncSetAttribute((const void*)k_match_list<false, 5>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - (int)kListStatic) == hipSuccess &&
             hipFuncSetAttribute((const void*)k_match_list<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - (int)kListStatic) == hipSuccess &&
             hipFuncSetAttribute((const void*)k_match_list<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - (int)kListStatic) == hipSuccess;
        if (!ok) (void)hipGetLastError();
    }
    if (Qout) *Qout = Q;
    if (lds_out) *lds_out = ldsl;
    return ok;
}

// Does the dense matcher take its keyed form (k_match_keyed: subsampling, disp_max > 255, no candidate records) when
// the list form is not usable?  with_lr: the raw rows of the fused L/R check count against the LDS (the pipeline's
// form; the decision before E1 asks for it).  *lds2_out: dynamic LDS of the launch.
static bool match_keyed_usable(const svh_elas_params& p, const Dims& d, int32_t prior_absmax, int32_t plane_radius,
                               bool with_lr, size_t* lds2_out) {
    static const bool ordered = svh::env("SVH_MATCH_ORDERED") != nullptr;
    const bool keyed_ok = !ordered && prior_absmax < (1 << 19) && p.disp_max < 512 && plane_radius <= 15 &&
                          d.W < 65536 && p.grid_size > 1;
    const size_t lds = (size_t)d.W * sizeof(uint4);
    const size_t lds2 = 2 * lds + (with_lr ? (size_t)2 * d.DW * sizeof(float) : 0);
    if (lds2_out) *lds2_out = lds2;
    // rows up to 1920 px (77 KB with the raw-disparity rows) still take the keyed kernel: two blocks
    // per CU, measured 2-5 % ahead of the ordered fallback on 1920x1080 since the kernel got leaner
    constexpr size_t keyed_lds_max = 96 * 1024;
    constexpr size_t kStaticLds = 64 * sizeof(int);   // s_P of both match kernels counts against the limit
    bool use_keyed = keyed_ok && lds2 + kStaticLds <= keyed_lds_max;
    if (use_keyed && lds2 + kStaticLds > 64 * 1024) {
        // opt in on whichever device is current (a few us, wide rows only); a refusal sends the
        // launch down the ordered path instead of failing later as a launch error
        use_keyed = hipFuncSetAttribute((const void*)k_match_keyed<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)keyed_lds_max) == hipSuccess &&
                    hipFuncSetAttribute((const void*)k_match_keyed<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)keyed_lds_max) == hipSuccess;
        if (!use_keyed) (void)hipGetLastError();
    }
    return use_keyed;
}

// E1 writes only the Sobel planes when BOTH matchers assemble their descriptor rows themselves: the LDS support
// kernel, and the list form or (round 5: subsampling, disp_max > 255) the keyed form of the dense matcher.
bool descriptors_on_the_fly(const svh_elas_params& p, const Dims& d, int32_t prior_absmax, int32_t plane_radius,
                            bool have_lists) {
    static const bool off = svh::env("SVH_DESC_FLY") && atoi(svh::env("SVH_DESC_FLY")) == 0;
    static const bool keyed_fly = !(svh::env("SVH_DESC_FLY_KEYED") && atoi(svh::env("SVH_DESC_FLY_KEYED")) == 0);
    if (off || d.W < 8 || support_strip(p, d, nullptr) == 0) return false;
    if (match_list_usable(p, d, prior_absmax, plane_radius, have_lists, nullptr, nullptr)) return true;
    return keyed_fly && match_keyed_usable(p, d, prior_absmax, plane_radius, true, nullptr);
}

// returns false with *error set when the group cannot be matched (a HIP refusal between the descriptor stage and
// here: the caller reports SVH_ERR_HIP for the group)
bool launch_match(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                  const GroupDev& G, const DevMaps* lr_out, bool write_raw, const char** error) {
    if (error) *error = nullptr;
    MatchParams P;
    P.W = d.W; P.H = d.H; P.DW = d.DW; P.DH = d.DH; P.gw = d.gw; P.gh = d.gh; P.gwords = d.gwords;
    P.grid_size = p.grid_size; P.sub = p.subsampling; P.disp_max = p.disp_max;
    P.match_texture = p.match_texture; P.plane_radius = G.plane_radius;
    P.grid_magic = (uint32_t)(0x100000000ull / (uint64_t)p.grid_size) + 1u;
    P.npairs = g;
    const size_t lds = (size_t)d.W * sizeof(uint4);
    constexpr size_t kStaticLds = 64 * sizeof(int);   // s_P of both match kernels counts against the limit
    // the keyed kernel packs cost and scan rank into one int32 (see k_match_keyed)
    size_t lds2 = 0;
    const bool use_keyed = match_keyed_usable(p, d, G.prior_absmax, G.plane_radius, lr_out != nullptr, &lds2);
    // round 4: the list form of the keyed kernel (per-cell candidate records, v_sad_hi_u8 keys, LDS-DMA staging)
    {
        MatchList Q{};
        size_t ldsl = 0;
        const bool ok = match_list_usable(p, d, G.prior_absmax, G.plane_radius, G.lists != nullptr, &Q, &ldsl);
        const int half = Q.half;
        if (!ok && !use_keyed && G.desc_fly) {
            // descriptors_on_the_fly() and this selection are the same functions of the same arguments; what can
            // differ between the two calls is the driver's answer to the LDS opt-in.  The descriptor maps of this
            // group do not exist, so it cannot fall back: the group fails, the next one decides again.
            if (error) *error = "both row matchers were refused after the descriptor stage had left only the Sobel planes";
            return false;
        }
        if (ok) {
            Timed timed_(cx, "k_match");
            const dim3 grid((unsigned)(Q.xcd ? (d.DH * g + 7) / 8 * 8 : d.DH * g)), block(2 * half);
            hipStream_t s = (hipStream_t)cx.stream;
            DevMaps none{};
            const DevMaps& o = lr_out ? *lr_out : none;
            const int wr = lr_out ? (write_raw ? 1 : 0) : 1;
            const float thr = lr_out ? (float)p.lr_threshold : 0.f;
            if (lr_out) {
                if (Q.kIters == 5) hipLaunchKernelGGL((k_match_list<true, 5>), grid, block, ldsl, s, G, P, Q, o, wr, thr);
                else hipLaunchKernelGGL((k_match_list<true, 8>), grid, block, ldsl, s, G, P, Q, o, wr, thr);
                return true;
            }
            if (Q.kIters == 5) hipLaunchKernelGGL((k_match_list<false, 5>), grid, block, ldsl, s, G, P, Q, o, wr, thr);
            else hipLaunchKernelGGL((k_match_list<false, 8>), grid, block, ldsl, s, G, P, Q, o, wr, thr);
            return false;
        }
    }
    if (use_keyed) {
        Timed timed_(cx, "k_match");
        // threads per map and row block: the row is covered in `iters` equal passes with little idle tail
        // threads per map (64..256; 256 measured best: isolated 32-pair launch 359 us vs 450 at 512 and 526 at 128)
        static const int mt = std::min(256, std::max(64, svh::env("SVH_MATCH_THREADS") ? atoi(svh::env("SVH_MATCH_THREADS")) : 256));
        const int iters = (d.DW + mt - 1) / mt;
        const int half = std::min(256, ((d.DW + iters - 1) / iters + 63) / 64 * 64);
        const dim3 grid(xcd_blocks(d.DH * g)), block(2 * half);   // (a multiple of 8: the kernel's XCD-aware row order)
        hipStream_t s = (hipStream_t)cx.stream;
        if (lr_out) {
            hipLaunchKernelGGL(k_match_keyed<true>, grid, block, lds2, s, G, P, *lr_out, write_raw ? 1 : 0,
                               (float)p.lr_threshold);
            return true;   // the L/R check is done
        }
        DevMaps none{};
        hipLaunchKernelGGL(k_match_keyed<false>, grid, block, lds2, s, G, P, none, 1, 0.f);
    } else if (lds + kStaticLds <= 64 * 1024) {
        Timed timed_(cx, "k_match");
        static const int mt = svh::env("SVH_MATCH_THREADS") ? atoi(svh::env("SVH_MATCH_THREADS")) : 256;
        const int iters = (d.DW + mt - 1) / mt;
        const int threads = std::min(512, ((d.DW + iters - 1) / iters + 63) / 64 * 64);
        hipLaunchKernelGGL(k_match<true>, dim3(1, d.DH, 2 * g), dim3(threads), lds,
                           (hipStream_t)cx.stream, G, P);
    } else {
        LAUNCH("k_match", k_match<false>, grid2d(d.DW, d.DH, 2 * g), dim3(64, 4), G, P);
    }
    return false;
}

void launch_lr(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
               const GroupDev& G, const DevMaps& out) {
    LAUNCH("k_lr", k_lr, grid2d(d.DW, d.DH, g), dim3(64, 4), G, out, d.DW, d.DH, p.subsampling,
           (float)p.lr_threshold);
}

void launch_segments(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                     int32_t nside, const GroupDev& G, const DevMaps& out, const PostScratch& S, bool mask) {
    const int n = d.DW * d.DH, z = g * nside;
    int min_size = p.speckle_size;
    if (p.subsampling) min_size = (int)(sqrtf((float)p.speckle_size) * 2);  // elas.cpp:1218
    const dim3 lin((n + 255) / 256, z), b256(256);
    launch_segments_label(cx, p, d, g, nside, G, out, S);
    // mask == false: the caller follows up with the tile kernels, which apply the verdict on load
    if (mask) LAUNCH("k_seg_mask", k_seg_mask, lin, b256, G, out, S, nside, n, min_size);
}


void launch_segments_label(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                           int32_t nside, const GroupDev& G, const DevMaps& in, const PostScratch& S) {
    const int n = d.DW * d.DH, z = g * nside;
    int min_size = p.speckle_size;
    if (p.subsampling) min_size = (int)(sqrtf((float)p.speckle_size) * 2);  // elas.cpp:1218
    const dim3 lin((n + 255) / 256, z), b256(256);
    const dim3 tiles((d.DW + CX - 1) / CX, (d.DH + CY - 1) / CY, z);
    static const bool tile_xcd = !(svh::env("SVH_TILE_XCD") && atoi(svh::env("SVH_TILE_XCD")) == 0);
    if (tile_xcd)
        LAUNCH("k_seg_tile", k_seg_tile, dim3(xcd_blocks((int)(tiles.x * tiles.y * tiles.z))), dim3(CX, 4), G, in, S, nside,
               d.DW, d.DH, p.speckle_sim_threshold, z);
    else
        LAUNCH("k_seg_tile", k_seg_tile, tiles, dim3(CX, 4), G, in, S, nside, d.DW, d.DH,
               p.speckle_sim_threshold, 0);
    const int nhor = ((d.DH - 1) / CY) * d.DW;              // pixels on interior horizontal borders
    const int nver = ((d.DW - 1) / CX) * d.DH;              // ... and vertical ones
    if (nhor + nver > 0)
        LAUNCH("k_seg_border", k_seg_border, dim3((nhor + nver + 255) / 256, z), b256, G, in, S, nside,
               d.DW, d.DH, p.speckle_sim_threshold, nhor);
    LAUNCH("k_seg_sum", k_seg_sum, dim3((tiles.x * tiles.y + 3) / 4, z), b256, G, S, nside, d.DW, d.DH, min_size);
}


void launch_gap(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                int32_t nside, const GroupDev& G, const DevMaps& out, const PostScratch& S) {
    int gap = p.ipol_gap_width;
    if (p.subsampling) gap = p.ipol_gap_width / 2 + 1;  // elas.cpp:1340
    const int z = g * nside;
    if (gap <= 16 && !p.add_corners) {
        LAUNCH("k_gap_rows", k_gap_local<false>, grid2d(d.DW, d.DH, z), dim3(64, 4), G, out, S, nside,
               d.DW, d.DH, gap);
        LAUNCH("k_gap_cols", k_gap_local<true>, grid2d(d.DW, d.DH, z), dim3(64, 4), G, out, S, nside,
               d.DW, d.DH, gap);
    } else {
        static const bool seq = svh::env("SVH_GAP_SEQ") && atoi(svh::env("SVH_GAP_SEQ")) != 0;   // (A/B: the one-thread-per-line form)
        const int nch = (d.DW + 63) / 64;
        if (!seq && nch <= kGapChunks) {
            {
                Timed timed_(cx, "k_gap_rows_scan");
                hipLaunchKernelGGL(k_gap_rows_scan, dim3((d.DH + 3) / 4, z), dim3(256), (size_t)4 * nch * 64 * sizeof(float),
                                   (hipStream_t)cx.stream, G, out, nside, d.DW, d.DH, gap, (int)p.add_corners);
            }
            LAUNCH("k_gap_cols_seg", k_gap_cols_seg, dim3((d.DW + 63) / 64, z), dim3(64, kGapSeg), G, out,
                   nside, d.DW, d.DH, gap, p.add_corners);
            return;
        }
        LAUNCH("k_gap_rows_seq", k_gap_lines<false>, dim3((d.DH + 63) / 64, z), dim3(64), G, out,
               nside, d.DW, d.DH, gap, p.add_corners);
        LAUNCH("k_gap_cols_seq", k_gap_lines<true>, dim3((d.DW + 63) / 64, z), dim3(64), G, out,
               nside, d.DW, d.DH, gap, p.add_corners);
    }
}

// ---------------------------------------------------------------------------
// Tile versions of E14 + E15 for the default configuration (gap <= 4, no corner extrapolation,
// adaptive mean, no median): the row pass and the column pass of a filter run in ONE kernel on a
// 64x32 tile whose halo (the column pass needs row-pass results a few rows up and down) is
// recomputed in LDS.  gap: D -> tmp, mean: tmp -> D, so two map round trips instead of four.
// The arithmetic, bounds and tap order are those of k_gap_local / k_adaptive_mean above.
// ---------------------------------------------------------------------------
constexpr int QX = 64, QY = 32;

// nearest valid neighbour on each side (<= gap away, inside [0, len)) -> interpolated value
__device__ __forceinline__ float gap_pick(float val, const float* line, int stride, int pos, int len, int gap) {
    // Round 6: the eight neighbours are read up front, together (they lie inside the tile's halo whatever the pixel
    // is); reading one only when the one before it was not decisive made eight dependent LDS round trips of a
    // handful of comparisons.  The compiler barrier keeps hipcc from sinking the reads back under the conditions.
    float a[4], b[4];
#pragma unroll
    for (int k = 1; k <= 4; k++) {
        a[k - 1] = line[-k * stride];
        b[k - 1] = line[k * stride];
    }
    asm volatile("" ::: "memory");
    if (val >= 0) return val;
    int l = 0, r = 0;
    float vl = 0.f, vr = 0.f;
#pragma unroll
    for (int k = 4; k >= 1; k--) {
        if (k <= gap && pos - k >= 0 && a[k - 1] >= 0) { l = k; vl = a[k - 1]; }
        if (k <= gap && pos + k < len && b[k - 1] >= 0) { r = k; vr = b[k - 1]; }
    }
    return (l && r && r <= gap - l + 1) ? gap_value(vl, vr) : val;
}

// min_size > 0: the speckle verdict (E13, k_seg_mask) is applied while the tile is loaded --
// a pixel of a component below min_size reads as -10 -- so the masked map never goes to memory
__global__ __launch_bounds__(256) void k_gap_tile(GroupDev G, DevMaps m, PostScratch S, int nside, int DW,
                                                  int DH, int gap, int min_size, int nz) {
    constexpr int HG = 4;
    __shared__ float sA[QY + 2 * HG][QX + 2 * HG];   // D with halo
    __shared__ float sB[QY + 2 * HG][QX];            // row-pass result, rows with halo
    TileIdx tb;
    if (!tile_index(QX, QY, DW, DH, nz, &tb)) return;   // (block-uniform, before any barrier)
    int pair;
    const float* D = post_map(m, tb.z, nside, &pair);
    if (!G.hdr->active[pair]) return;
    float* out = S.tmp + (size_t)tb.z * DW * DH;
    const int x0 = tb.x * QX, y0 = tb.y * QY;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    {
        // Rows wave, wave + 4, ... of the tile: 64 columns by the lanes, the 2 HG columns beyond them
        // as a second, short pass (row-wise addressing costs a fraction of splitting a flat index).
        // The label -> root -> size look-ups are dependent loads: each level is fetched for all
        // of a thread's entries before the next one starts.
        constexpr int AW = QX + 2 * HG, AH = QY + 2 * HG, NR = AH / 4, NE = NR + (AH * 8 + 255) / 256;
        static_assert(AH % 4 == 0 && AW - 64 == 8, "tile shape");
        const int32_t* L = S.labels + (size_t)tb.z * DW * DH;
        const int32_t* Cn = S.counts + (size_t)tb.z * DW * DH;
        const int lane = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
        float val[NE];
        int lab[NE];
        int er[NE], ec[NE];
#pragma unroll
        for (int k = 0; k < NE; k++) {
            if (k < NR) {
                er[k] = wave + 4 * k;
                ec[k] = lane;
            } else {
                const int e = tid + 256 * (k - NR);
                er[k] = e >> 3;
                ec[k] = 64 + (e & 7);
            }
            const int gy = y0 - HG + er[k], gx = x0 - HG + ec[k];
            const bool in = er[k] < AH && gy >= 0 && gy < DH && gx >= 0 && gx < DW;
            val[k] = in ? D[(size_t)gy * DW + gx] : -10.f;
            lab[k] = (in && min_size > 0) ? L[(size_t)gy * DW + gx] : -1;
        }
        if (min_size > 0) {
#pragma unroll
            for (int k = 0; k < NE; k++)
                if (lab[k] >= 0) lab[k] = L[lab[k]];           // tile root -> component root
#pragma unroll
            for (int k = 0; k < NE; k++)
                if (lab[k] >= 0 && Cn[lab[k]] < min_size) val[k] = -10.f;
        }
#pragma unroll
        for (int k = 0; k < NE; k++)
            if (er[k] < AH) sA[er[k]][ec[k]] = val[k];
    }
    __syncthreads();
    for (int i = tid; i < (QY + 2 * HG) * QX; i += 256) {
        const int r = i >> 6, c = i & 63;
        const int gx = x0 + c;
        sB[r][c] = gx < DW ? gap_pick(sA[r][c + HG], &sA[r][c + HG], 1, gx, DW, gap) : -10.f;
    }
    __syncthreads();
    for (int i = tid; i < QY * QX; i += 256) {
        const int r = i >> 6, c = i & 63;
        const int gy = y0 + r, gx = x0 + c;
        if (gy < DH && gx < DW)
            out[(size_t)gy * DW + gx] = gap_pick(sB[r + HG][c], &sB[r + HG][c], QX, gy, DH, gap);
    }
}

// One 8- or 4-tap adaptive-mean evaluation with the taps in window order (tap k lies k - HL places
// from the centre).  The reference keeps the window in a ring of kTaps slots and adds the slots in
// slot order (elas.cpp:1570-1640): slot s holds tap (s - first) & (kTaps - 1), first = pos + back -
// lead.  Slots j and j + 4 are added first (one SSE add of the two halves; the sum of a pair does not
// depend on the order of its two terms), which pairs tap m with tap m + 4 whatever `first` is, and the
// four pair sums are then added from slot 0 upwards -- so the ring position only rotates the order of
// that last chain, by kRot = first & 3.  kRot is a template argument: every wave of k_mean_tile
// works on positions of one residue, so the taps are read at constant LDS offsets and no select or
// address arithmetic is spent on the ring.
//
// The weights are carried at a quarter of the reference's: w/4 = max(0, 1 - mm/4) is one fused
// multiply-add with the clamp modifier instead of a subtraction and a maximum (v_max_f32 issues at
// half the rate of v_fma_f32, profiles/r03_microbench_valu.txt).  Scaling by 2^-2 is exact and
// commutes with every rounding in the chain (products, pair sums, the two chains; no value of a
// disparity map is near the denormal range), so both sums come out at exactly a quarter and their
// quotient is the reference's, bit for bit.
__device__ __forceinline__ float am_weight_q(float val, float centre) {
    const float mm = __uint_as_float(__float_as_uint(__fsub_rn(val, centre)) & 0x4F000000u);
    return __builtin_amdgcn_fmed3f(__fmaf_rn(mm, -0.25f, 1.0f), 0.0f, 1.0f);
}

template <int kTaps, int kRot, typename Tap>
__device__ __forceinline__ bool am_eval_rot(Tap tap, float centre, float* res) {
    constexpr int kCentre = kTaps == 8 ? 4 : 2;        // the centre tap: difference 0, weight 4 (here 1)
    float P[4], F[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        float w0 = 1.0f, f0 = centre;
        if (m != kCentre) {
            const float t0 = tap(m);
            w0 = am_weight_q(t0, centre);
            f0 = __fmul_rn(t0, w0);
        }
        if (kTaps == 8) {
            float w1 = 1.0f, f1 = centre;
            if (m + 4 != kCentre) {
                const float t1 = tap(m + 4);
                w1 = am_weight_q(t1, centre);
                f1 = __fmul_rn(t1, w1);
            }
            P[m] = __fadd_rn(w0, w1);
            F[m] = __fadd_rn(f0, f1);
        } else {
            P[m] = w0;
            F[m] = f0;
        }
    }
    constexpr int a = (4 - kRot) & 3;
    const float wsum = __fadd_rn(__fadd_rn(__fadd_rn(P[a], P[(a + 1) & 3]), P[(a + 2) & 3]), P[(a + 3) & 3]);
    const float fsum = __fadd_rn(__fadd_rn(__fadd_rn(F[a], F[(a + 1) & 3]), F[(a + 2) & 3]), F[(a + 3) & 3]);
    if (wsum > 0) {
        const float dv = __fdiv_rn(fsum, wsum);
        if (dv >= 0) {
            *res = dv;
            return true;
        }
    }
    return false;
}

// LDS layout of k_mean_tile.  The input tile is stored column-swizzled: column c of a row sits at
// (c & 3) * kMeanSub + (c >> 2), so that a wave whose lanes take every fourth column (one residue of
// x mod 4) reads 16 consecutive words per row; four rows per wave, rows kMeanSA = 80 = 16 (mod 64)
// words apart, cover the 64 banks exactly once.  The row-pass result is stored plainly with rows 65
// words apart (the stride-4 writes of the row pass and the row-strided reads of the column pass are
// both conflict-free with it).
constexpr int kMeanSub = 20, kMeanSA = 80, kMeanSB = 65;

template <int kTaps, int kQ>
__device__ __forceinline__ void mean_row_pass(const float* sA, float* sB, int x0, int y0, int DW, int DH,
                                              int lane) {
    constexpr int back = kTaps == 8 ? 3 : 1, lead = kTaps - 1, HL = lead - back, HR = back;
    constexpr int AH = QY + HL + HR;
    const int l = lane & 15, rs = lane >> 4;
    const int gx = x0 + 4 * l + kQ;
    const bool xin = gx >= lead - back && gx <= DW - 1 - back;
    const float* a = sA + rs * kMeanSA + l;
    float* b = sB + rs * kMeanSB + 4 * l + kQ;
#pragma unroll 2
    for (int r4 = 0; r4 < AH; r4 += 4) {
        const int r = r4 + rs;
        if (r < AH) {
            const int gy = y0 - HL + r;
            const float* row = a + r4 * kMeanSA;
            const float centre = row[((kQ + HL) & 3) * kMeanSub + ((kQ + HL) >> 2)];
            float res = centre < 0 ? -10.f : 0.f;
            if (centre >= 0 && xin && gy >= 3 && gy < DH - 3)
                am_eval_rot<kTaps, (kQ - HL) & 3>(
                    [&](int k) { return row[((kQ + k) & 3) * kMeanSub + ((kQ + k) >> 2)]; }, centre, &res);
            b[r4 * kMeanSB] = res;
        }
    }
}

template <int kTaps, int kQ>
__device__ __forceinline__ void mean_col_pass(const float* sB, float* D, const float (&orig)[QY / 4], int x0,
                                              int y0, int DW, int DH, int lane) {
    constexpr int back = kTaps == 8 ? 3 : 1, lead = kTaps - 1, HL = lead - back;
    const int gx = x0 + lane;
    if (gx >= DW) return;
    const bool xin = gx >= 3 && gx < DW - 3;
#pragma unroll
    for (int j = 0; j < QY / 4; j++) {
        const int r = kQ + 4 * j, gy = y0 + r;
        if (gy >= DH) break;
        const float* col = sB + r * kMeanSB + lane;
        float res = orig[j];
        // (round 6 measured: reading the window's taps together with the centre, before the centre is looked at, costs
        // more than the saved round trip -- 61 -> 66 us per 32 pairs: invalid centres are many and skip their taps)
        const float centre = col[HL * kMeanSB];
        if (centre >= 0 && xin && gy >= lead - back && gy <= DH - 1 - back)
            am_eval_rot<kTaps, (kQ - HL) & 3>([&](int k) { return col[k * kMeanSB]; }, centre, &res);
        D[(size_t)gy * DW + gx] = res;
    }
}

template <int kTaps>
__global__ __launch_bounds__(256) void k_mean_tile(GroupDev G, DevMaps m, PostScratch S, int nside, int DW,
                                                   int DH, int nz) {
    constexpr int back = kTaps == 8 ? 3 : 1, lead = kTaps - 1;
    constexpr int HL = lead - back, HR = back;         // taps reach HL before and HR after the centre
    constexpr int AW = QX + HL + HR, AH = QY + HL + HR;
    static_assert((AW + 3) / 4 <= kMeanSub && 4 * kMeanSub <= kMeanSA && QX <= kMeanSB, "tile layout");
    __shared__ float sA[AH * kMeanSA];                 // input (negatives as -10) with halo, swizzled
    __shared__ float sB[AH * kMeanSB];                 // horizontal-pass result (D_tmp), rows with halo
    TileIdx tb;
    if (!tile_index(QX, QY, DW, DH, nz, &tb)) return;   // (block-uniform, before any barrier)
    int pair;
    float* D = post_map(m, tb.z, nside, &pair);
    if (!G.hdr->active[pair]) return;
    const float* in = S.tmp + (size_t)tb.z * DW * DH;   // written by k_gap_tile
    const int x0 = tb.x * QX, y0 = tb.y * QY;
    const int lane = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
    float orig[QY / 4];
    {
        // Rows wave, wave + 4, ... of the tile: 64 columns by the lanes, the HL + HR columns beyond
        // them as a second, short pass.  All of a thread's entries are requested before the first
        // one is used (a load / clamp / store loop waits for memory once per entry), and so are the
        // values the column pass falls back to where it does not fire.
        constexpr int NR = (AH + 3) / 4, XW = AW - 64, NX = (AH * XW + 255) / 256;
        float val[NR], xv[NX];
        const int gxm = x0 - HL + lane;
        const bool cin = gxm >= 0 && gxm < DW;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int r = wave + 4 * j, gy = y0 - HL + r;
            val[j] = (r < AH && gy >= 0 && gy < DH && cin) ? in[(size_t)gy * DW + gxm] : -10.f;
        }
        const int tid = wave * 64 + lane;
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const int e = tid + 256 * k, r = e / XW, c = 64 + (e - r * XW);
            const int gy = y0 - HL + r, gx = x0 - HL + c;
            xv[k] = (e < AH * XW && gy >= 0 && gy < DH && gx < DW) ? in[(size_t)gy * DW + gx] : -10.f;
        }
#pragma unroll
        for (int j = 0; j < QY / 4; j++) {
            const int gy = y0 + wave + 4 * j, gx = x0 + lane;
            orig[j] = (gy < DH && gx < DW) ? in[(size_t)gy * DW + gx] : 0.f;
        }
        const int sw = (lane & 3) * kMeanSub + (lane >> 2);
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int r = wave + 4 * j;
            if (r < AH) sA[r * kMeanSA + sw] = val[j] < 0 ? -10.f : val[j];   // D_copy (elas.cpp:1553-1560)
        }
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const int e = tid + 256 * k, r = e / XW, c = 64 + (e - r * XW);
            if (e < AH * XW) sA[r * kMeanSA + (c & 3) * kMeanSub + (c >> 2)] = xv[k] < 0 ? -10.f : xv[k];
        }
    }
    __syncthreads();
    // horizontal pass -> D_tmp: -10 where the input is invalid, 0 where valid but never written.
    // Wave q takes the columns with x = q (mod 4), four rows at a time.
    switch (wave) {
        case 0: mean_row_pass<kTaps, 0>(sA, sB, x0, y0, DW, DH, lane); break;
        case 1: mean_row_pass<kTaps, 1>(sA, sB, x0, y0, DW, DH, lane); break;
        case 2: mean_row_pass<kTaps, 2>(sA, sB, x0, y0, DW, DH, lane); break;
        default: mean_row_pass<kTaps, 3>(sA, sB, x0, y0, DW, DH, lane); break;
    }
    __syncthreads();
    // vertical pass on D_tmp, wave q on the rows with y = q (mod 4); where it does not fire the map
    // keeps the filter's input
    switch (wave) {
        case 0: mean_col_pass<kTaps, 0>(sB, D, orig, x0, y0, DW, DH, lane); break;
        case 1: mean_col_pass<kTaps, 1>(sB, D, orig, x0, y0, DW, DH, lane); break;
        case 2: mean_col_pass<kTaps, 2>(sB, D, orig, x0, y0, DW, DH, lane); break;
        default: mean_col_pass<kTaps, 3>(sB, D, orig, x0, y0, DW, DH, lane); break;
    }
}

// the two tile kernels replace k_gap_rows/cols + k_mean_h/v (four map round trips -> two)
bool post_tiles_ok(const svh_elas_params& p) {
    static const bool off = svh::env("SVH_NO_POST_TILES") != nullptr;
    int gap = p.ipol_gap_width;
    if (p.subsampling) gap = p.ipol_gap_width / 2 + 1;
    return !off && gap <= 4 && !p.add_corners && p.filter_adaptive_mean && !p.filter_median;
}

void launch_gap_mean_tiles(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                           int32_t nside, const GroupDev& G, const DevMaps& out, const PostScratch& S) {
    int gap = p.ipol_gap_width;
    if (p.subsampling) gap = p.ipol_gap_width / 2 + 1;  // elas.cpp:1340
    const dim3 gr((d.DW + QX - 1) / QX, (d.DH + QY - 1) / QY, g * nside), b(64, 4);
    int min_size = p.speckle_size;
    if (p.subsampling) min_size = (int)(sqrtf((float)p.speckle_size) * 2);  // elas.cpp:1218
    static const bool tile_xcd = !(svh::env("SVH_TILE_XCD") && atoi(svh::env("SVH_TILE_XCD")) == 0);
    const int nz = tile_xcd ? g * nside : 0;
    const dim3 gl = tile_xcd ? dim3(xcd_blocks((int)(gr.x * gr.y * gr.z))) : gr;
    LAUNCH("k_gap_tile", k_gap_tile, gl, b, G, out, S, nside, d.DW, d.DH, gap, min_size > 1 ? min_size : 0, nz);
    if (p.subsampling) LAUNCH("k_mean_tile", k_mean_tile<4>, gl, b, G, out, S, nside, d.DW, d.DH, nz);
    else               LAUNCH("k_mean_tile", k_mean_tile<8>, gl, b, G, out, S, nside, d.DW, d.DH, nz);
}

void launch_adaptive_mean(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                          int32_t nside, const GroupDev& G, const DevMaps& out, const PostScratch& S) {
    const dim3 gr = grid2d(d.DW, d.DH, g * nside), b(64, 4);
    if (p.subsampling) {
        LAUNCH("k_mean_h", (k_adaptive_mean<false, 4>), gr, b, G, out, S, nside, d.DW, d.DH);
        LAUNCH("k_mean_v", (k_adaptive_mean<true, 4>), gr, b, G, out, S, nside, d.DW, d.DH);
    } else {
        LAUNCH("k_mean_h", (k_adaptive_mean<false, 8>), gr, b, G, out, S, nside, d.DW, d.DH);
        LAUNCH("k_mean_v", (k_adaptive_mean<true, 8>), gr, b, G, out, S, nside, d.DW, d.DH);
    }
}

void launch_median(const LaunchCtx& cx, const Dims& d, int32_t g, int32_t nside, const GroupDev& G,
                   const DevMaps& out, const PostScratch& S) {
    const dim3 gr = grid2d(d.DW, d.DH, g * nside), b(64, 4);
    LAUNCH("k_median_h", k_median<false>, gr, b, G, out, S, nside, d.DW, d.DH);
    LAUNCH("k_median_v", k_median<true>, gr, b, G, out, S, nside, d.DW, d.DH);
}

}  // namespace svh

#if SVH_ML_PROBE == 7
// probe build only (tools/Makefile `probe`): read (and clear) the event counters of k_match_list
extern "C" int svh_probe_ml_counters(unsigned long long* out16, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(svh::g_ml_cnt), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (clear) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(svh::g_ml_cnt), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif
