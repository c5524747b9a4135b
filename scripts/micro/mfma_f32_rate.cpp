// mfma_f32_rate.cpp -- what does the chip sustain on NOTHING but v_mfma_f32_32x32x2_f32 (the instruction of the VQGAN
// convolutions, vqgan_conv.h, and of the f32 attention flavour, attn_f32.h)?  Whole-chip kernel, `wps` waves per SIMD
// (256 x wps workgroups of 256 threads... one workgroup = 4 waves = one per SIMD; grid = CUs x wps x rounds), operands
// random floats in registers, `iters` x 64 MFMAs per wave over `chains` accumulators.  Prints executed TFLOP/s against the
// nominal 157.3 (256 FLOP / cycle / CU x 256 CUs x 2.4 GHz).
// Build: hipcc -O2 --offload-arch=gfx950 -o /tmp/mfma_f32_rate scripts/micro/mfma_f32_rate.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ inline void mfma(f32x16& d, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

template <int CHAINS, int OCC>
__global__ __launch_bounds__(256, OCC) void burn(const uint32_t* seed, float* sink, int iters) {
    float a[8], b[8];
    uint32_t h = seed[threadIdx.x & 63] ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        a[i] = ((int)((h >> 9) & 0xffff) - 32768) * (1.0f / 32768.0f);
        h = h * 1664525u + 1013904223u;
        b[i] = ((int)((h >> 9) & 0xffff) - 32768) * (1.0f / 32768.0f);
    }
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 64; ++g) mfma(acc[g % CHAINS], a[g & 7], b[(g >> 3) & 7]);
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) sink[0] = s;
}

template <int CHAINS, int OCC>
static void run(int iters, const uint32_t* seed, float* sink, int cus) {
    const int grid = cus * OCC;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((burn<CHAINS, OCC>), dim3(grid), dim3(256), 0, 0, seed, sink, iters / 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((burn<CHAINS, OCC>), dim3(grid), dim3(256), 0, 0, seed, sink, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * 4 * iters * 64 * 4096.0;
    printf("chains %d  waves/SIMD %d  grid %d: %.2f ms  %.1f TFLOP/s  = %.3f of 157.3\n", CHAINS, OCC, grid, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 157.3);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    uint32_t hs[64];
    for (int i = 0; i < 64; ++i) hs[i] = 0x9e3779b9u * (i + 1);
    uint32_t* seed; float* sink;
    CK(hipMalloc(&seed, sizeof(hs))); CK(hipMalloc(&sink, 4));
    CK(hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice));
    run<1, 1>(iters, seed, sink, cus);
    run<2, 1>(iters, seed, sink, cus);
    run<4, 1>(iters, seed, sink, cus);
    run<8, 1>(iters, seed, sink, cus);
    run<1, 2>(iters, seed, sink, cus);
    run<4, 2>(iters, seed, sink, cus);
    run<4, 1>(iters, seed, sink, cus);
    return 0;
}
