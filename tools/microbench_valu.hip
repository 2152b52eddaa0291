// VALU issue-rate probe for gfx950: how many cycles does a wave64 integer VALU
// instruction occupy a SIMD?  Each wave runs N dependent-free chains of one op.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_valu.hip && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int kOp>
__global__ __launch_bounds__(256) void k_probe(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const uint32_t b = seed ^ 0x12345678u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (kOp == 0) {
                a0 = __builtin_amdgcn_sad_u8(a0, b, a0); a1 = __builtin_amdgcn_sad_u8(a1, b, a1);
                a2 = __builtin_amdgcn_sad_u8(a2, b, a2); a3 = __builtin_amdgcn_sad_u8(a3, b, a3);
                a4 = __builtin_amdgcn_sad_u8(a4, b, a4); a5 = __builtin_amdgcn_sad_u8(a5, b, a5);
                a6 = __builtin_amdgcn_sad_u8(a6, b, a6); a7 = __builtin_amdgcn_sad_u8(a7, b, a7);
            } else if (kOp == 1) {
                a0 = (a0 ^ b) + 1; a1 = (a1 ^ b) + 1; a2 = (a2 ^ b) + 1; a3 = (a3 ^ b) + 1;
                a4 = (a4 ^ b) + 1; a5 = (a5 ^ b) + 1; a6 = (a6 ^ b) + 1; a7 = (a7 ^ b) + 1;
            } else {
                a0 = min(a0 ^ 1u, b); a1 = min(a1 ^ 2u, b); a2 = min(a2 ^ 3u, b); a3 = min(a3 ^ 4u, b);
                a4 = min(a4 ^ 5u, b); a5 = min(a5 ^ 6u, b); a6 = min(a6 ^ 7u, b); a7 = min(a7 ^ 8u, b);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int kOp>
void run(const char* name, int instr_per_iter) {
    uint32_t* out;
    const int blocks = 256 * 8, iters = 4096;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_probe<kOp><<<blocks, 256>>>(out, 1, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe<kOp><<<blocks, 256>>>(out, 1, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * instr_per_iter;   // wave-instructions
    const double per_simd = winstr / (256.0 * 4);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("%-10s %.3f ms  %.3f G wave-instr/s/SIMD  -> %.2f cycles per wave-instr at %d MHz\n", name, ms,
           per_simd / ms / 1e6, ms * 1e-3 * clk_khz * 1e3 / per_simd, clk_khz / 1000);
    hipFree(out);
}

int main() {
    run<0>("v_sad_u8", 64);
    run<1>("xor+add", 128);
    run<2>("xor+min", 128);
    run<0>("v_sad_u8", 64);
    return 0;
}
