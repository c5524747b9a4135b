// mfma_power.cpp -- does the ORDER of v_mfma_f32_32x32x16_bf16 instructions change what the chip sustains at its power
// limit?  Whole-chip kernel, one wave per SIMD (1024 workgroups x 256 threads, 4 per CU in turn), operands random bf16
// held in registers, `iters` x 64 MFMAs per wave, nothing else in the loop:
//   chains = 1 : one accumulator, every MFMA depends on the one before it (C = D of the previous instruction)
//   chains = 2 : two accumulators alternating (the X phases of attn_bwd64.h)
//   chains = 8 : eight accumulators in turn (the Y phases)
//   same   = 1 : every MFMA reads the SAME A / B registers (no operand toggling at the pipe's inputs); 0: eight pairs in turn
//   same   = 2 : as 0, and the A operand of every MFMA comes from LDS (one ds_read_b128 per lane and MFMA, 1 KiB per wave:
//                what the attention kernels' fragment reads cost); 3: A from LDS for every SECOND MFMA (a fragment feeds two)
//   same   = 4 : as 0, plus two v_fma_f32 per MFMA on random data (the softmax / dS vector work beside the matrix pipe);
//   same   = 5 : four v_fma_f32 per MFMA
// Prints ms, executed PFLOP/s and the rocm-smi samples (clock, power) taken beside each run by scripts/gpu_mfma_power.sh.
// Build: hipcc -O2 --offload-arch=gfx950 -o scripts/micro/mfma_power scripts/micro/mfma_power.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ inline void mfma(f32x16& d, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

template <int CHAINS, int SAME>
__global__ __launch_bounds__(256) void burn(const uint32_t* seed, float* sink, int iters, float amp) {
    __shared__ bf16x8 frag[8 * 256];      // 8 fragments per thread, 32 KiB
    float v[4] = {1.0f, 2.0f, 3.0f, 4.0f};
    bf16x8 a[8], b[8];
    uint32_t h = seed[threadIdx.x & 63] ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            a[i][j] = (__bf16)(((int)((h >> 9) & 0xffff) - 32768) * (1.0f / 32768.0f) * amp);
            h = h * 1664525u + 1013904223u;
            b[i][j] = (__bf16)(((int)((h >> 9) & 0xffff) - 32768) * (1.0f / 32768.0f) * amp);
        }
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    for (int i = 0; i < 8; ++i) frag[i * 256 + threadIdx.x] = a[i];
    __syncthreads();
    const float m1 = 1.0f + amp * 1e-7f, m2 = amp * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        if (SAME == 2 || SAME == 3) {
            bf16x8 r[4];
#pragma unroll
            for (int g = 0; g < 3; ++g) r[g] = frag[g * 256 + threadIdx.x];
#pragma unroll
            for (int g = 0; g < 64; ++g) {
                const int step = SAME == 2 ? 1 : 2, f = g / step;          // fragment index
                if (g % step == 0) r[(f + 3) & 3] = frag[((f + 3) & 7) * 256 + threadIdx.x];
                __builtin_amdgcn_sched_barrier(0);
                mfma(acc[g % CHAINS], r[f & 3], b[(g >> 3) & 7]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 64; ++g) {
                mfma(acc[g % CHAINS], a[SAME == 1 ? 0 : (g & 7)], b[SAME == 1 ? 0 : ((g >> 3) & 7)]);
                if (SAME >= 4) {
#pragma unroll
                    for (int k = 0; k < (SAME == 4 ? 2 : 4); ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m1), "v"(m2));
                }
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int k = 0; k < 4; ++k) s += v[k];
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
    const int chains = argc > 1 ? atoi(argv[1]) : 8, same = argc > 2 ? atoi(argv[2]) : 0;
    const float amp = argc > 3 ? (float)atof(argv[3]) : 1.0f;
    const int iters = argc > 4 ? atoi(argv[4]) : 700000;      // ~3 s: 4 waves per SIMD in turn x iters x 64 MFMAs
    uint32_t hs[64];
    for (int i = 0; i < 64; ++i) hs[i] = 0x9e3779b9u * (i + 1);
    uint32_t* seed;
    float* sink;
    CK(hipMalloc(&seed, sizeof(hs)));
    CK(hipMalloc(&sink, 16));
    CK(hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&](int it) {
#define RUN(C, S) if (chains == C && same == S) hipLaunchKernelGGL((burn<C, S>), dim3(1024), dim3(256), 0, 0, seed, sink, it, amp);
        RUN(1, 0) RUN(2, 0) RUN(4, 0) RUN(8, 0) RUN(1, 1) RUN(8, 1) RUN(8, 2) RUN(8, 3) RUN(8, 4) RUN(8, 5)
    };
    launch(100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    launch(iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 1024.0 * 4 * (double)iters * 64 * 2.0 * 32 * 32 * 16;
    printf("chains %d  same-operands %d  amplitude %g : %8.1f ms  %6.3f PFLOP/s executed  (%.1f cycles per MFMA at 2.4 GHz)\n", chains, same, amp,
           ms, flop / ms * 1e-12, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * 4));
    return 0;
}
