// decode_bench.cpp -- one cached-decode attention step (q_len = 1, 32 heads x 128) over a K-token cache through the
// C ABI, without Python:   decode_bench <liblwm_hip.so> [K=131072] [splits=512] [reps=20]
// Prints HIP-event us per step for lwm_attn_fwd (decode kernel) and for lwm_attn_combine, and the cache bytes
// read per second (2 * K * 4096 * 2 B per step).
// Build: hipcc -O2 --offload-arch=gfx950 -I include -o scripts/micro/decode_bench scripts/micro/decode_bench.cpp -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
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
        float v = ((h & 0xffff) * (1.0f / 32768.0f) - 1.0f);
        uint32_t b = __builtin_bit_cast(uint32_t, v);
        p[i] = (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
    }
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: decode_bench <lib> [K] [splits] [reps]\n");
        return 2;
    }
    const int K = argc > 2 ? atoi(argv[2]) : 131072, splits = argc > 3 ? atoi(argv[3]) : 512, reps = argc > 4 ? atoi(argv[4]) : 20;
    const int H = 32, D = 128;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    auto fwd = (int (*)(const LwmAttnArgs*, void*))dlsym(lib, "lwm_attn_fwd");
    auto combine = (int (*)(const float*, const float*, int32_t, LwmTensor4, float*, float*, int32_t, int32_t, int32_t, int32_t,
                            void*))dlsym(lib, "lwm_attn_combine");
    auto last_error = (const char* (*)(void))dlsym(lib, "lwm_last_error");
    const size_t n = (size_t)K * H * D;
    uint16_t *q, *k, *v, *out;
    float *o_parts, *l_parts, *lse;
    uint8_t* mask;
    CK(hipMalloc(&q, H * D * 2));
    CK(hipMalloc(&out, H * D * 2));
    CK(hipMalloc(&k, n * 2));
    CK(hipMalloc(&v, n * 2));
    CK(hipMalloc(&o_parts, (size_t)splits * H * D * 4));
    CK(hipMalloc(&l_parts, (size_t)splits * H * 4));
    CK(hipMalloc(&lse, H * 4));
    CK(hipMalloc(&mask, K));
    CK(hipMemset(mask, 1, K));
    fill_bf16<<<1, 256>>>(q, H * D, 1u);
    fill_bf16<<<2048, 256>>>(k, n, 2u);
    fill_bf16<<<2048, 256>>>(v, n, 3u);
    LwmAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = LwmTensor4{q, (int64_t)H * D, (int64_t)H * D, D};
    a.k = LwmTensor4{k, (int64_t)n, (int64_t)H * D, D};
    a.v = LwmTensor4{v, (int64_t)n, (int64_t)H * D, D};
    a.out_acc = o_parts; a.lse_acc = l_parts;
    a.B = 1; a.H = H; a.Sq = 1; a.Sk = K; a.D = D;
    a.scale = 1.0f / sqrtf((float)D);
    a.dense_mask = mask; a.mask_stride_b = K; a.mask_stride_q = K;
    a.k_splits = splits;
    const LwmTensor4 o4 = {out, (int64_t)H * D, (int64_t)H * D, D};
    auto step = [&](bool do_fwd, bool do_comb) {
        if (do_fwd && fwd(&a, nullptr) != 0) {
            fprintf(stderr, "fwd: %s\n", last_error());
            exit(2);
        }
        if (do_comb && combine(o_parts, l_parts, splits, o4, nullptr, lse, 1, 1, H, D, nullptr) != 0) {
            fprintf(stderr, "combine: %s\n", last_error());
            exit(2);
        }
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_us = [&](bool f, bool c) {
        step(f, c);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) step(f, c);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / reps;
    };
    const float us_f = time_us(true, false), us_c = time_us(false, true), us_b = time_us(true, true);
    uint16_t ho[4];
    CK(hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost));
    printf("%-36s K=%d splits=%d  decode %.1f us (%.2f TB/s)  combine %.1f us  both %.1f us (%.2f TB/s)  out %04x %04x %04x %04x\n", argv[1],
           K, splits, us_f, 4.0 * n / us_f * 1e-6, us_c, us_b, 4.0 * n / us_b * 1e-6, ho[0], ho[1], ho[2], ho[3]);
    return 0;
}
