// gemv_bench.cpp -- the decode-step projections of LWM-7B through lwm_gemv_multi_bf16, without Python:
//   gemv_bench <liblwm_hip.so> [rows=1] [reps=50]
// Per call shape: HIP-event us per call (both launches) and weight bytes per second.
// Build: hipcc -O2 --offload-arch=gfx950 -I include -o scripts/micro/gemv_bench scripts/micro/gemv_bench.cpp -ldl
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

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        float v = ((h & 0xffff) * (1.0f / 32768.0f) - 1.0f) * 0.05f;
        uint32_t b = __builtin_bit_cast(uint32_t, v);
        p[i] = (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
    }
}

struct Shape {
    const char* name;
    int K, nmat, N[3];
};

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: gemv_bench <lib> [rows] [reps]\n");
        return 2;
    }
    const int rows = argc > 2 ? atoi(argv[2]) : 1, reps = argc > 3 ? atoi(argv[3]) : 50;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    auto gemv = (int (*)(const void*, int64_t, int32_t, const void* const*, void* const*, const int64_t*, float* const*,
                         const int32_t*, void*, int32_t, int32_t, void*))dlsym(lib, "lwm_gemv_multi_bf16");
    auto wsb = (int64_t (*)(int32_t, int32_t, int32_t))dlsym(lib, "lwm_gemv_workspace_bytes");
    auto last_error = (const char* (*)(void))dlsym(lib, "lwm_last_error");
    const Shape shapes[] = {{"wq | wk | wv", 4096, 3, {4096, 4096, 4096}}, {"wo", 4096, 1, {4096}},
                            {"w1 | w3", 4096, 2, {11008, 11008}},         {"w2", 11008, 1, {4096}},
                            {"lm_head", 4096, 1, {32000}}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double tot_b = 0, tot_us = 0;
    for (const Shape& s : shapes) {
        uint16_t* x;
        CK(hipMalloc(&x, (size_t)rows * s.K * 2));
        fill_bf16<<<8, 256>>>(x, (size_t)rows * s.K, 1u);
        const void* w[3];
        void* y[3];
        int64_t ldy[3];
        int64_t ws_bytes = 0;
        double bytes = 0;
        for (int i = 0; i < s.nmat; ++i) {
            uint16_t *wi, *yi;
            CK(hipMalloc(&wi, (size_t)s.K * s.N[i] * 2));
            CK(hipMalloc(&yi, (size_t)rows * s.N[i] * 2));
            fill_bf16<<<1024, 256>>>(wi, (size_t)s.K * s.N[i], 7u + i);
            w[i] = wi; y[i] = yi; ldy[i] = s.N[i];
            ws_bytes += wsb(rows, s.K, s.N[i]);
            bytes += 2.0 * s.K * s.N[i];
        }
        void* ws;
        CK(hipMalloc(&ws, ws_bytes));
        auto call = [&] {
            if (gemv(x, s.K, s.nmat, w, y, ldy, nullptr, s.N, ws, rows, s.K, nullptr) != 0) {
                fprintf(stderr, "gemv: %s\n", last_error());
                exit(2);
            }
        };
        call();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) call();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        uint16_t h4[4];
        CK(hipMemcpy(h4, y[0], 8, hipMemcpyDeviceToHost));
        printf("  %-14s K=%-5d N=%-17s %7.1f us  %5.2f TB/s  y %04x %04x %04x %04x\n", s.name, s.K,
               s.nmat == 3 ? "3 x 4096" : (s.nmat == 2 ? "2 x 11008" : (s.N[0] == 32000 ? "32000" : "4096")), us, bytes / us * 1e-6,
               h4[0], h4[1], h4[2], h4[3]);
        tot_b += bytes;
        tot_us += us;
    }
    printf("%s rows=%d: one layer's projections + lm_head %.1f us, %.2f TB/s\n", argv[1], rows, tot_us, tot_b / tot_us * 1e-6);
    return 0;
}
