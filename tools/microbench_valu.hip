// VALU issue-rate probe for gfx950: how many cycles does a wave64 VALU instruction occupy its
// SIMD, by opcode and by the number of resident waves per SIMD?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_valu.hip && /tmp/mb
//
// Every wave runs 8 independent chains of ONE opcode (inline asm, so the instruction is exactly
// the one named), 64 instructions per loop trip.  Two clocks are reported:
//   wall   = HIP-event time of the launch x the device clock / wave-instructions per SIMD
//   memtime= s_memtime ticks (shader cycles) a wave spent in its loop x waves per SIMD / its
//            instruction count (independent of the clock the chip actually holds)
// Rows with fp32 controls (v_fma_f32, v_pk_fma_f32) next to the integer ops the matching kernels
// are made of (v_sad_u8, v_min3_i32, v_lshl_add_u32, ...).  Waves per SIMD = 1, 2, 4, 8: blocks of
// 256 threads put one wave on each SIMD of a CU, the grid is 256 CUs x waves-per-SIMD blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

enum Op { FMA_F32, PK_FMA_F32, ADD_U32, LSHL_ADD_U32, MIN3_I32, MIN_U32, MED3_U32, SAD_U8, AND_B32, XOR_B32,
          CNDMASK, ADD3_U32, MAD_U32_U24, PK_ADD_U16, MIN_U32_DPP, MOV_DPP, BFE_U32, MUL_LO_U32, PERM_B32,
          CNDMASK_SGPR, CMP_CNDMASK, CMP_VCC, OR_B32, SUB_U32, LSHLREV_B32, LSHRREV_B32, MOV_B32, MAX_U32, MIN_I32, MUL_U32_U24,
          ADD_F32, MUL_F32, CVT_F32_U32, BFI_B32, OR3_B32, AND_OR_B32, LSHL_OR_B32, ADD_LSHL_U32, BFE_I32, ALIGNBIT,
          SAD_HI_U8, SAD_U16, SAD_U32, MSAD_U8, QSAD_PK, MQSAD_PK, MQSAD_U32, SUB_SDWA, NUM_OPS };
static const char* kNames[NUM_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_lshl_add_u32", "v_min3_i32",
    "v_min_u32", "v_med3_u32", "v_sad_u8", "v_and_b32", "v_xor_b32", "v_cndmask_b32", "v_add3_u32", "v_mad_u32_u24",
    "v_pk_add_u16", "v_min_u32_dpp", "v_mov_b32_dpp", "v_bfe_u32", "v_mul_lo_u32", "v_perm_b32",
    "v_cndmask(sgpr)", "v_cmp+v_cndmask", "v_cmp_lt_u32 vcc", "v_or_b32", "v_sub_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_mov_b32", "v_max_u32", "v_min_i32",
    "v_mul_u32_u24", "v_add_f32", "v_mul_f32", "v_cvt_f32_u32", "v_bfi_b32", "v_or3_b32", "v_and_or_b32", "v_lshl_or_b32", "v_add_lshl_u32", "v_bfe_i32", "v_alignbit_b32",
    "v_sad_hi_u8", "v_sad_u16", "v_sad_u32", "v_msad_u8", "v_qsad_pk_u16_u8", "v_mqsad_pk_u16_u8", "v_mqsad_u32_u8", "v_sub_u32_sdwa"};

#define CH8(STMT) STMT(a0) STMT(a1) STMT(a2) STMT(a3) STMT(a4) STMT(a5) STMT(a6) STMT(a7)

template <int kOp>
__global__ __launch_bounds__(256) void k_probe(uint32_t* out, unsigned long long* ticks, uint32_t seed, int iters) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed ^ 0x12345678u, c = seed * 77u + 1u;
    // 64-bit pairs for the packed fp32 op
    double p0 = a0, p1 = a1, p2 = a2, p3 = a3, p4 = a4, p5 = a5, p6 = a6, p7 = a7, pb = b, pc = c;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q0 = {a0, a1, a2, a3}, q1 = {a4, a5, a6, a7}, q2 = {a1, a3, a5, a7}, q3 = {a0, a2, a4, a6};   // 128-bit accumulators (v_mqsad_u32_u8)
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a0), "v"(b) : "vcc");
    const unsigned long long smask = __builtin_amdgcn_ballot_w64((threadIdx.x & 3) != 0) ^ (unsigned long long)seed;   // an SGPR pair
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define S_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pb), "v"(pc));
#define S_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
#define S_MIN3(x) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_MIN(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_MED3(x) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_SAD(x) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define S_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_CND(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define S_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_PKADD(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_MINDPP(x) asm volatile("v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define S_MOVDPP(x) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
#define S_BFE(x) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(x));
#define S_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_CNDS(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(smask));
#define S_CMPCND(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define S_CMP(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define S_OR(x) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_SUB(x) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_LSHL(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
#define S_LSHR(x) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x));
#define S_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
#define S_MAX(x) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_MINI(x) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_ADDF(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_MULF(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define S_CVT(x) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
#define S_BFI(x) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_OR3(x) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
#define S_ADDLSHL(x) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(x) : "v"(b));
#define S_BFEI(x) asm volatile("v_bfe_i32 %0, %0, 3, 29" : "+v"(x));
#define S_ALIGN(x) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(x) : "v"(b));
#define S_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define S_SADHI(x) asm volatile("v_sad_hi_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define S_SAD16(x) asm volatile("v_sad_u16 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define S_SAD32(x) asm volatile("v_sad_u32 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define S_MSAD(x) asm volatile("v_msad_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define S_QSAD(x) asm volatile("v_qsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define S_MQSADPK(x) asm volatile("v_mqsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));
#define S_MQSAD32(x, y) asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(b));
#define S_SUBSDWA(x) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(x) : "v"(b));
            if (kOp == FMA_F32) { CH8(S_FMA) }
            else if (kOp == PK_FMA_F32) { S_PKFMA(p0) S_PKFMA(p1) S_PKFMA(p2) S_PKFMA(p3) S_PKFMA(p4) S_PKFMA(p5) S_PKFMA(p6) S_PKFMA(p7) }
            else if (kOp == ADD_U32) { CH8(S_ADD) }
            else if (kOp == LSHL_ADD_U32) { CH8(S_LSHLADD) }
            else if (kOp == MIN3_I32) { CH8(S_MIN3) }
            else if (kOp == MIN_U32) { CH8(S_MIN) }
            else if (kOp == MED3_U32) { CH8(S_MED3) }
            else if (kOp == SAD_U8) { CH8(S_SAD) }
            else if (kOp == AND_B32) { CH8(S_AND) }
            else if (kOp == XOR_B32) { CH8(S_XOR) }
            else if (kOp == CNDMASK) { CH8(S_CND) }
            else if (kOp == ADD3_U32) { CH8(S_ADD3) }
            else if (kOp == MAD_U32_U24) { CH8(S_MAD24) }
            else if (kOp == PK_ADD_U16) { CH8(S_PKADD) }
            else if (kOp == MIN_U32_DPP) { CH8(S_MINDPP) }
            else if (kOp == MOV_DPP) { CH8(S_MOVDPP) }
            else if (kOp == BFE_U32) { CH8(S_BFE) }
            else if (kOp == MUL_LO_U32) { CH8(S_MULLO) }
            else if (kOp == PERM_B32) { CH8(S_PERM) }
            else if (kOp == CNDMASK_SGPR) { CH8(S_CNDS) }
            else if (kOp == CMP_CNDMASK) { CH8(S_CMPCND) }
            else if (kOp == CMP_VCC) { CH8(S_CMP) }
            else if (kOp == OR_B32) { CH8(S_OR) }
            else if (kOp == SUB_U32) { CH8(S_SUB) }
            else if (kOp == LSHLREV_B32) { CH8(S_LSHL) }
            else if (kOp == LSHRREV_B32) { CH8(S_LSHR) }
            else if (kOp == MOV_B32) { CH8(S_MOV) }
            else if (kOp == MAX_U32) { CH8(S_MAX) }
            else if (kOp == MIN_I32) { CH8(S_MINI) }
            else if (kOp == MUL_U32_U24) { CH8(S_MUL24) }
            else if (kOp == ADD_F32) { CH8(S_ADDF) }
            else if (kOp == MUL_F32) { CH8(S_MULF) }
            else if (kOp == CVT_F32_U32) { CH8(S_CVT) }
            else if (kOp == BFI_B32) { CH8(S_BFI) }
            else if (kOp == OR3_B32) { CH8(S_OR3) }
            else if (kOp == AND_OR_B32) { CH8(S_ANDOR) }
            else if (kOp == LSHL_OR_B32) { CH8(S_LSHLOR) }
            else if (kOp == ADD_LSHL_U32) { CH8(S_ADDLSHL) }
            else if (kOp == BFE_I32) { CH8(S_BFEI) }
            else if (kOp == ALIGNBIT) { CH8(S_ALIGN) }
            else if (kOp == SAD_HI_U8) { CH8(S_SADHI) }
            else if (kOp == SAD_U16) { CH8(S_SAD16) }
            else if (kOp == SAD_U32) { CH8(S_SAD32) }
            else if (kOp == MSAD_U8) { CH8(S_MSAD) }
            else if (kOp == QSAD_PK) { S_QSAD(p0) S_QSAD(p1) S_QSAD(p2) S_QSAD(p3) S_QSAD(p4) S_QSAD(p5) S_QSAD(p6) S_QSAD(p7) }
            else if (kOp == MQSAD_PK) { S_MQSADPK(p0) S_MQSADPK(p1) S_MQSADPK(p2) S_MQSADPK(p3) S_MQSADPK(p4) S_MQSADPK(p5) S_MQSADPK(p6) S_MQSADPK(p7) }
            else if (kOp == MQSAD_U32) { S_MQSAD32(q0, p0) S_MQSAD32(q1, p1) S_MQSAD32(q2, p2) S_MQSAD32(q3, p3) S_MQSAD32(q0, p4) S_MQSAD32(q1, p5) S_MQSAD32(q2, p6) S_MQSAD32(q3, p7) }
            else if (kOp == SUB_SDWA) { CH8(S_SUBSDWA) }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    acc += (uint32_t)(p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7);
    acc += q0.x + q1.y + q2.z + q3.w;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int kOp>
void run(int waves_per_simd, int clk_khz) {
    const int blocks = 256 * waves_per_simd, iters = 2048;
    uint32_t* out;
    unsigned long long* ticks;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&ticks, (size_t)blocks * 4 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_probe<kOp><<<blocks, 256>>>(out, ticks, 1, 64);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k_probe<kOp><<<blocks, 256>>>(out, ticks, 1, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    static unsigned long long host[256 * 8 * 4];
    hipMemcpy(host, ticks, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
    double tsum = 0;
    for (int i = 0; i < blocks * 4; i++) tsum += (double)host[i];
    const double instr_per_wave = (double)iters * 64;
    const double winstr = (double)blocks * 4 * instr_per_wave;
    const double per_simd = winstr / (256.0 * 4);
    // readcyclecounter lowers to s_memtime (shader cycles): a wave's loop time x the SIMD's share
    const double wave_ticks = tsum / (blocks * 4);
    printf("%-16s w/SIMD %d  %8.3f ms  wall: %5.2f cyc/wave-instr at %4d MHz   s_memtime: %5.2f cyc/wave-instr   (%.3f G wave-instr/s/SIMD)\n",
           kNames[kOp], waves_per_simd, best, best * 1e-3 * clk_khz * 1e3 / per_simd, clk_khz / 1000,
           wave_ticks / (instr_per_wave * waves_per_simd), per_simd / best / 1e6);
    hipFree(out); hipFree(ticks);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int kOp>
void sweep(int clk_khz) {
    for (int w : {1, 2, 4, 8}) run<kOp>(w, clk_khz);
}

int main(int argc, char** argv) {
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("# %s, %d CUs, clock attribute %d MHz; cyc = launch time x clock / (wave-instructions per SIMD)\n", p.gcnArchName,
           p.multiProcessorCount, clk_khz / 1000);
    printf("# a dependent-free stream of one opcode; w/SIMD = resident waves per SIMD (grid = 256 x w blocks of 256)\n");
    if (argc > 1 && !strcmp(argv[1], "sad")) {
        // round 4: the SAD family (the quad forms slide a 4-byte window over 8 bytes: four SADs per instruction)
        sweep<SAD_U8>(clk_khz);
        sweep<SAD_HI_U8>(clk_khz);
        sweep<SAD_U16>(clk_khz);
        sweep<SAD_U32>(clk_khz);
        sweep<MSAD_U8>(clk_khz);
        sweep<QSAD_PK>(clk_khz);
        sweep<MQSAD_PK>(clk_khz);
        sweep<MQSAD_U32>(clk_khz);
        sweep<SUB_SDWA>(clk_khz);
        sweep<SUB_U32>(clk_khz);
        return 0;
    }
    sweep<FMA_F32>(clk_khz);
    sweep<PK_FMA_F32>(clk_khz);
    sweep<ADD_U32>(clk_khz);
    sweep<LSHL_ADD_U32>(clk_khz);
    sweep<MIN3_I32>(clk_khz);
    sweep<MIN_U32>(clk_khz);
    sweep<MED3_U32>(clk_khz);
    sweep<SAD_U8>(clk_khz);
    sweep<AND_B32>(clk_khz);
    sweep<XOR_B32>(clk_khz);
    sweep<CNDMASK>(clk_khz);
    sweep<ADD3_U32>(clk_khz);
    sweep<MAD_U32_U24>(clk_khz);
    sweep<PK_ADD_U16>(clk_khz);
    sweep<MIN_U32_DPP>(clk_khz);
    sweep<MOV_DPP>(clk_khz);
    sweep<BFE_U32>(clk_khz);
    sweep<MUL_LO_U32>(clk_khz);
    sweep<PERM_B32>(clk_khz);
    sweep<CNDMASK_SGPR>(clk_khz);
    printf("# v_cmp+v_cndmask: 128 instructions per trip counted as 64 PAIRS (cyc per pair)\n");
    sweep<CMP_CNDMASK>(clk_khz);
    sweep<CMP_VCC>(clk_khz);
    sweep<OR_B32>(clk_khz);
    sweep<SUB_U32>(clk_khz);
    sweep<LSHLREV_B32>(clk_khz);
    sweep<LSHRREV_B32>(clk_khz);
    sweep<MOV_B32>(clk_khz);
    sweep<MAX_U32>(clk_khz);
    sweep<MIN_I32>(clk_khz);
    sweep<MUL_U32_U24>(clk_khz);
    sweep<ADD_F32>(clk_khz);
    sweep<MUL_F32>(clk_khz);
    sweep<CVT_F32_U32>(clk_khz);
    sweep<BFI_B32>(clk_khz);
    sweep<OR3_B32>(clk_khz);
    sweep<AND_OR_B32>(clk_khz);
    sweep<LSHL_OR_B32>(clk_khz);
    sweep<ADD_LSHL_U32>(clk_khz);
    sweep<BFE_I32>(clk_khz);
    sweep<ALIGNBIT>(clk_khz);
    return 0;
}
