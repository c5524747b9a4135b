// conv_bench.cpp -- times lwm_conv2d_nhwc_f32 on the VQGAN's dominant layer shapes through the C ABI, without
// Python:   conv_bench <liblwm_hip.so> [frames=32] [reps=3]
// Per shape: HIP-event ms, TFLOP/s (2*M*K*N), fraction of the 157.3 TF exact-f32 MFMA roof and an
// order-independent 64-bit checksum of the output bits (two libraries that agree bit for bit print the same).
// Build: hipcc -O2 --offload-arch=gfx950 -I include -o scripts/micro/conv_bench scripts/micro/conv_bench.cpp -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lwm_hip.h"

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(2);                                                \
        }                                                           \
    } while (0)

__global__ void fill_f32(float* p, size_t n, uint32_t seed, float amp) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = ((h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f) * amp;
    }
}
__global__ void bit_sum(const uint32_t* p, size_t n, unsigned long long* out) {
    unsigned long long s = 0;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) s += (unsigned long long)p[i] * (2 * (i % 1000003) + 1);
    atomicAdd(out, s);
}

struct Shape {
    const char* name;
    int H, W, Cin, Cout, up, res;
    int stride = 1;      // 2 = Downsample: pad 0 (the zero pad bottom / right is the out-of-image rule), Ho = H / 2
    int listed = 1;      // part of the "listed layers" total (the set of rounds 2-3)
};

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: conv_bench <lib> [frames] [reps]\n");
        return 2;
    }
    const int B = argc > 2 ? atoi(argv[2]) : 32, reps = argc > 3 ? atoi(argv[3]) : 3;
    // LWM_BENCH_AMP=0: all-zero operands (same instruction stream, no operand toggling: separates the clock from the kernel)
    const float amp = getenv("LWM_BENCH_AMP") ? (float)atof(getenv("LWM_BENCH_AMP")) : 1.0f;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    auto conv = (int (*)(const LwmConvArgs*, void*))dlsym(lib, "lwm_conv2d_nhwc_f32");
    auto last_error = (const char* (*)(void))dlsym(lib, "lwm_last_error");
    // 3x3 SAME layers of the default VQGAN (ch 128, ch_mult 1,1,2,2,4) that carry the FLOPs
    const Shape shapes[] = {
        {"enc/dec 256^2 128->128 (+res)", 256, 256, 128, 128, 0, 1},
        {"dec up 128^2->256^2 128->128", 128, 128, 128, 128, 1, 0},
        {"128^2 128->128 (+res)", 128, 128, 128, 128, 0, 1},
        {"dec 128^2 256->128", 128, 128, 256, 128, 0, 0},
        {"dec up 64^2->128^2 256->256", 64, 64, 256, 256, 1, 0},
        {"64^2 256->256 (+res)", 64, 64, 256, 256, 0, 1},
        {"enc 64^2 128->256", 64, 64, 128, 256, 0, 0},
        {"32^2 256->256 (+res)", 32, 32, 256, 256, 0, 1},
        {"32^2 512->512 (+res)", 32, 32, 512, 512, 0, 1},
        {"dec up 32^2->64^2 512->512", 32, 32, 512, 512, 1, 0},
        {"dec 64^2 512->256", 64, 64, 512, 256, 0, 0},
        {"16^2 768->768 (+res)", 16, 16, 768, 768, 0, 1},
        // the rest of the network's shapes (not in the total above)
        {"enc conv_in 256^2 3->128", 256, 256, 3, 128, 0, 0, 1, 0},
        {"dec conv_out 256^2 128->3", 256, 256, 128, 3, 0, 0, 1, 0},
        {"dec 256^2 256->128", 256, 256, 256, 128, 0, 0, 1, 0},
        {"enc down 256^2->128^2 128->128", 256, 256, 128, 128, 0, 0, 2, 0},
        {"enc down 128^2->64^2 256->256", 128, 128, 256, 256, 0, 0, 2, 0},
        {"dec up 16^2->32^2 768->768", 16, 16, 768, 768, 1, 0, 1, 0},
        {"dec 32^2 768->512", 32, 32, 768, 512, 0, 0, 1, 0},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned long long* dsum;
    CK(hipMalloc(&dsum, 8));
    double tot_flop = 0, tot_ms = 0;
    for (const Shape& s : shapes) {
        const int Ho = s.stride == 2 ? s.H / 2 : s.H << s.up, Wo = s.stride == 2 ? s.W / 2 : s.W << s.up;
        const size_t nx = (size_t)B * s.H * s.W * s.Cin, nw = (size_t)9 * s.Cin * s.Cout, ny = (size_t)B * Ho * Wo * s.Cout;
        float *x, *w, *bias, *res, *y;
        CK(hipMalloc(&x, nx * 4));
        CK(hipMalloc(&w, nw * 4));
        CK(hipMalloc(&bias, s.Cout * 4));
        CK(hipMalloc(&res, ny * 4));
        CK(hipMalloc(&y, ny * 4));
        fill_f32<<<2048, 256>>>(x, nx, 1u, 1.0f * amp);
        fill_f32<<<256, 256>>>(w, nw, 2u, 0.03f * amp);
        fill_f32<<<1, 256>>>(bias, s.Cout, 3u, 0.1f);
        fill_f32<<<2048, 256>>>(res, ny, 4u, 1.0f * amp);
        LwmConvArgs a;
        memset(&a, 0, sizeof(a));
        a.x = x; a.w = w; a.bias = bias; a.residual = s.res ? res : nullptr; a.y = y;
        a.B = B; a.Hin = s.H; a.Win = s.W; a.Cin = s.Cin; a.Cout = s.Cout; a.KH = 3; a.KW = 3;
        a.stride = s.stride; a.pad = s.stride == 2 ? 0 : 1; a.up_shift = s.up; a.Ho = Ho; a.Wo = Wo;
        if (conv(&a, nullptr) != 0) {
            fprintf(stderr, "conv: %s\n", last_error());
            return 2;
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) conv(&a, nullptr);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        CK(hipMemset(dsum, 0, 8));
        bit_sum<<<1024, 256>>>((const uint32_t*)y, ny, dsum);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, dsum, 8, hipMemcpyDeviceToHost));
        const double flop = 2.0 * B * Ho * Wo * 9.0 * s.Cin * s.Cout;
        if (s.listed) {
            tot_flop += flop;
            tot_ms += ms;
        }
        printf("  %-34s %8.3f ms  %6.1f TF/s  %.3f of roof  bits %016llx\n", s.name, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3, h);
        CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(bias)); CK(hipFree(res)); CK(hipFree(y));
    }
    printf("%s  frames %d: %.1f TF/s over the listed layers (%.3f of roof)\n", argv[1], B, tot_flop / tot_ms * 1e-9,
           tot_flop / tot_ms * 1e-9 / 157.3);
    // GroupNorm + SiLU on the two largest activation shapes (HBM bound: 4 B read twice + 4 B written per element)
    auto gn = (int (*)(const float*, const float*, const float*, float*, void*, int32_t, int64_t, int32_t, int32_t, float, int32_t,
                       void*))dlsym(lib, "lwm_groupnorm_silu_f32");
    auto gn_ws = (int64_t (*)(int32_t, int64_t, int32_t, int32_t))dlsym(lib, "lwm_groupnorm_workspace_bytes");
    const int gshapes[][2] = {{256 * 256, 128}, {128 * 128, 128}, {128 * 128, 256}, {64 * 64, 256}};
    for (auto& g : gshapes) {
        const int64_t HW = g[0];
        const int C = g[1];
        const size_t n = (size_t)B * HW * C;
        float *x, *y, *gam, *bet;
        void* ws;
        CK(hipMalloc(&x, n * 4));
        CK(hipMalloc(&y, n * 4));
        CK(hipMalloc(&gam, C * 4));
        CK(hipMalloc(&bet, C * 4));
        CK(hipMalloc(&ws, gn_ws(B, HW, C, 32)));
        fill_f32<<<2048, 256>>>(x, n, 7u, 2.0f);
        fill_f32<<<1, 256>>>(gam, C, 8u, 1.0f);
        fill_f32<<<1, 256>>>(bet, C, 9u, 0.2f);
        if (gn(x, gam, bet, y, ws, B, HW, C, 32, 1e-6f, 1, nullptr) != 0) {
            fprintf(stderr, "gn: %s\n", last_error());
            return 2;
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) gn(x, gam, bet, y, ws, B, HW, C, 32, 1e-6f, 1, nullptr);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        CK(hipMemset(dsum, 0, 8));
        bit_sum<<<1024, 256>>>((const uint32_t*)y, n, dsum);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, dsum, 8, hipMemcpyDeviceToHost));
        printf("  groupnorm+silu HW=%-6lld C=%-3d  %8.3f ms  %5.2f TB/s (12 B/element)  bits %016llx\n", (long long)HW, C, ms,
               12.0 * n / ms * 1e-9, h);
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(gam)); CK(hipFree(bet)); CK(hipFree(ws));
    }
    return 0;
}
