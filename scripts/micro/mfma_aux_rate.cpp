// mfma_aux_rate.cpp -- what do the OTHER instructions of the VQGAN convolution's k-quad cost beside its four
// v_mfma_f32_32x32x2_f32?  One k-quad = 4 MFMAs on two accumulators (vqgan_conv.h::conv_patch_body), plus, by switch:
//   L  buffer_load_dword per quad (0 / 2 / 4) from a hot 16 KiB table into a ring of B registers used RING quads later
//   D  one ds_read_b128 per quad (the A fragment), consumed by
//   P  two v_bfi_b32 (the operand picks)
//   I  0 = all of them in front of the quad's MFMAs (rounds 3-6), 1 = in the MFMAs' shadows (behind the 1st / 2nd / 3rd)
// `wps` waves per SIMD.  Prints executed TFLOP/s against 157.3.
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/micro/mfma_aux_rate scripts/micro/mfma_aux_rate.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__device__ inline f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ inline void fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ inline float bload(const float* base, uint32_t voff, uint32_t soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
template <int L, int D, int P, int I, int OCC, int UNR = 8>
__global__ __launch_bounds__(256, OCC) void burn(const float* tab, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x4* lds = (f32x4*)smem;
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f32x4{1.0f + i, 2.0f, 3.0f, 4.0f};
    __syncthreads();
    uint32_t m = (lane >> 5) ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));
    const uint32_t voff = (uint32_t)lane * 4u;
    constexpr int RING = 4;
    float bq[RING][4];
    for (int g = 0; g < RING; ++g)
        for (int k = 0; k < 4; ++k) bq[g][k] = 0.001f * (g * 4 + k + lane);
    f32x4 ar[4] = {lds[lane], lds[lane + 64], lds[lane + 128], lds[lane + 192]};
    float af[2][2] = {{1.0f, 2.0f}, {3.0f, 4.0f}};
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
    for (int it = 0; it < iters * 8 / UNR; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int cs = u & 1, ns = cs ^ 1, slot = (u + 3) % RING;
            const uint32_t soff = (uint32_t)((it * UNR + u) & 15) * 1024u;
            auto loads = [&](int half) {
                if (L == 4) { bq[slot][2 * half] = bload(tab, voff, soff + half * 512u); bq[slot][2 * half + 1] = bload(tab, voff + 256u, soff + half * 512u); }
                if (L == 2) { bq[slot][2 * half] = bload(tab, voff, soff + half * 512u); }
            };
            auto dsr = [&]() {
                if (D == 1) ar[ns] = lds[(lane + 64 * ((it * UNR + u) & 7)) & 1023];
                if (D == 2) ar[(u + 2) % 4] = lds[(lane + 64 * ((it * UNR + u) & 7)) & 1023];      // two quads ahead of its picks
            };
            auto dsr2 = [&]() {       // D == 3: the A operands straight from LDS, no vector instruction: dwords (hi, hi + 2) of the lane's 16-byte slot
                const float* fl = (const float*)smem;
                const int base = (((lane & 31) + 32 * ((it * UNR + u) & 7)) & 255) * 4 + (lane >> 5);
                af[ns][0] = fl[base];
                af[ns][1] = fl[base + 2];
            };
            auto pick = [&]() {
                if (D == 3) { dsr2(); return; }
                const int a = D == 2 ? (u + 1) % 4 : ns;
                if (P) asm("v_bfi_b32 %0, %2, %4, %3\n\tv_bfi_b32 %1, %2, %6, %5\n\ts_nop 1" : "=&v"(af[ns][0]), "=&v"(af[ns][1]) : "v"(m), "v"(ar[a][0]), "v"(ar[a][1]), "v"(ar[a][2]), "v"(ar[a][3]));
            };
            if (I == 0) {
                loads(0); loads(1); dsr(); fence(); pick();
                acc0 = mfma(af[cs][0], bq[u % RING][0], acc0);
                acc1 = mfma(af[cs][0], bq[u % RING][1], acc1);
                acc0 = mfma(af[cs][1], bq[u % RING][2], acc0);
                acc1 = mfma(af[cs][1], bq[u % RING][3], acc1);
                fence();
            } else {
                acc0 = mfma(af[cs][0], bq[u % RING][0], acc0); fence();
                dsr(); loads(0); fence();
                acc1 = mfma(af[cs][0], bq[u % RING][1], acc1); fence();
                loads(1); fence();
                acc0 = mfma(af[cs][1], bq[u % RING][2], acc0); fence();
                pick(); fence();
                acc1 = mfma(af[cs][1], bq[u % RING][3], acc1); fence();
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) sink[0] = s;
}
// 64 pixels x 64 channels per wave (two pixel blocks x two channel blocks: 8 MFMAs per k-quad on four accumulators) -- the
// same four B loads, two fragment reads, four picks: half the other instructions per MFMA, one wave per SIMD
template <int OCC>
__global__ __launch_bounds__(256, OCC) void burn8(const float* tab, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x4* lds = (f32x4*)smem;
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f32x4{1.0f + i, 2.0f, 3.0f, 4.0f};
    __syncthreads();
    uint32_t m = (lane >> 5) ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));
    const uint32_t voff = (uint32_t)lane * 4u;
    constexpr int RING = 4;
    float bq[RING][4];
    for (int g = 0; g < RING; ++g)
        for (int k = 0; k < 4; ++k) bq[g][k] = 0.001f * (g * 4 + k + lane);
    f32x4 ar[2][2] = {{lds[lane], lds[lane + 64]}, {lds[lane + 128], lds[lane + 192]}};
    float af[2][2][2] = {{{1.0f, 2.0f}, {3.0f, 4.0f}}, {{1.5f, 2.5f}, {3.5f, 4.5f}}};      // [parity][t][i]
    f32x16 acc[2][2];
    for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0; acc[0][1][r] = 0; acc[1][0][r] = 0; acc[1][1][r] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cs = u & 1, ns = cs ^ 1, slot = (u + 3) % RING;
            const uint32_t soff = (uint32_t)((it * 8 + u) & 15) * 1024u;
            auto load1 = [&](int k) { bq[slot][k] = bload(tab, voff + (k & 1) * 256u, soff + (k >> 1) * 512u); };
            auto dsr = [&](int i) { ar[ns][i] = lds[(lane + 64 * ((it * 8 + u + 3 * i) & 7)) & 1023]; };
            auto pick = [&](int i) {
                asm("v_bfi_b32 %0, %2, %4, %3\n\tv_bfi_b32 %1, %2, %6, %5\n\ts_nop 1" : "=&v"(af[ns][0][i]), "=&v"(af[ns][1][i]) : "v"(m), "v"(ar[ns][i][0]), "v"(ar[ns][i][1]), "v"(ar[ns][i][2]), "v"(ar[ns][i][3]));
            };
            int k = 0;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j, ++k) {
                        acc[i][j] = mfma(af[cs][t][i], bq[u % RING][2 * t + j], acc[i][j]); fence();
                        if (k == 0) { dsr(0); load1(0); }
                        if (k == 1) { dsr(1); load1(1); }
                        if (k == 2) load1(2);
                        if (k == 3) load1(3);
                        if (k == 5) pick(0);
                        if (k == 6) pick(1);
                        fence();
                    }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    if (s == 12345.678f) sink[0] = s;
}
template <int OCC>
static void run8(int iters, const float* tab, float* sink, int cus) {
    const int grid = cus * OCC;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((burn8<OCC>), dim3(grid), dim3(256), 16384, 0, tab, sink, iters / 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((burn8<OCC>), dim3(grid), dim3(256), 16384, 0, tab, sink, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * 4 * iters * 64 * 4096.0;
    printf("8 MFMAs per quad (64 px x 64 ch per wave), 4 loads 2 ds_read 4 picks in the shadows, waves/SIMD %d: %7.2f ms  %6.1f TFLOP/s = %.3f of 157.3\n", OCC, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}
template <int L, int D, int P, int I, int OCC, int UNR = 8>
static void run(int iters, const float* tab, float* sink, int cus) {
    const int grid = cus * OCC;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((burn<L, D, P, I, OCC, UNR>), dim3(grid), dim3(256), 16384, 0, tab, sink, iters / 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((burn<L, D, P, I, OCC, UNR>), dim3(grid), dim3(256), 16384, 0, tab, sink, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * 4 * iters * 32 * 4096.0;
    printf("unroll %3d quads  loads/quad %d  ds_read %d  picks %d  %s  waves/SIMD %d: %7.2f ms  %6.1f TFLOP/s = %.3f of 157.3\n", UNR, L, D, P, I ? "in the shadows" : "in front      ", OCC, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float *tab, *sink;
    CK(hipMalloc(&tab, 65536)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(tab, 0, 65536));
#define RUN2(L, D, P, I) run<L, D, P, I, 1>(iters, tab, sink, cus); run<L, D, P, I, 2>(iters, tab, sink, cus);
    RUN2(0, 0, 0, 1)
    RUN2(4, 0, 0, 0) RUN2(4, 0, 0, 1) RUN2(2, 0, 0, 1)
    RUN2(0, 1, 0, 1) RUN2(0, 1, 1, 1) RUN2(0, 0, 1, 1)
    RUN2(4, 1, 1, 0) RUN2(4, 1, 1, 1)
    // the same k-quad unrolled as the convolution kernels have it (two taps of eight (tap, chunk)s: 128 quads, ~27 KB of code)
    run<4, 1, 1, 1, 2, 128>(iters, tab, sink, cus);
    run<4, 1, 1, 1, 1, 128>(iters, tab, sink, cus);
    run<4, 1, 1, 1, 2, 256>(iters, tab, sink, cus);
    RUN2(0, 2, 1, 1) RUN2(4, 2, 1, 1)
    RUN2(0, 3, 0, 1) RUN2(4, 3, 0, 1)
    run8<1>(iters, tab, sink, cus);
    run8<2>(iters, tab, sink, cus);
    return 0;
}
