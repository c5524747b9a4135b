// fused_bench.cpp -- times the attention backward of one layer through the C ABI, without Python:
//   fused_bench <liblwm_hip.so> [S=32768] [H=32] [reps=3] [what=all|fused|two] [heads per fused launch = all]
// One line per library: HIP-event ms per launch of lwm_attn_fwd, lwm_attn_bwd_fused and of delta + dkdv + dq,
// checksums of dq / dk (fused vs two-kernel) and the largest element-wise dq difference, so that a timing
// variant that breaks the result is visible.  Used by scripts/gpu_fused_ab.sh to sweep variant builds in one GPU call.
// Build: hipcc -O2 --offload-arch=gfx950 -I include -o scripts/micro/fused_bench scripts/micro/fused_bench.cpp -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "lwm_hip.h"

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            exit(2);                                                          \
        }                                                                     \
    } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float amp) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        uint32_t g = h * 747796405u + 2891336453u;
        g ^= g >> 16;
        // sum of two uniforms: triangular, unit-ish variance after scaling
        float u = ((h & 0xffff) + (g & 0xffff)) * (1.0f / 65536.0f) - 1.0f;
        float v = u * amp;
        uint32_t b = __builtin_bit_cast(uint32_t, v);
        b += 0x7fffu + ((b >> 16) & 1u);
        p[i] = (uint16_t)(b >> 16);
    }
}

__global__ void abs_sum_bf16(const uint16_t* p, size_t n, double* out) {
    double s = 0;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) s += fabsf(__builtin_bit_cast(float, (uint32_t)p[i] << 16));
    atomicAdd(out, s);
}

// max |a - b| and max |b| over bf16 arrays (non-negative floats order like their bit patterns)
__global__ void max_diff_bf16(const uint16_t* a, const uint16_t* b, size_t n, unsigned long long* out) {
    float md = 0, mr = 0;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t)a[i] << 16), y = __builtin_bit_cast(float, (uint32_t)b[i] << 16);
        md = fmaxf(md, fabsf(x - y));
        mr = fmaxf(mr, fabsf(y));
    }
    atomicMax(out, (unsigned long long)__builtin_bit_cast(uint32_t, md));
    atomicMax(out + 1, (unsigned long long)__builtin_bit_cast(uint32_t, mr));
}

template <class F>
static F sym(void* lib, const char* name) {
    void* p = dlsym(lib, name);
    if (!p) {
        fprintf(stderr, "missing symbol %s\n", name);
        exit(2);
    }
    return (F)p;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: fused_bench <lib> [S] [H] [reps] [all|fused|two]\n");
        return 2;
    }
    const int S = argc > 2 ? atoi(argv[2]) : 32768, H = argc > 3 ? atoi(argv[3]) : 32, reps = argc > 4 ? atoi(argv[4]) : 3;
    const char* what = argc > 5 ? argv[5] : "all";
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    typedef int (*attn_fn)(const LwmAttnArgs*, void*);
    attn_fn fwd = sym<attn_fn>(lib, "lwm_attn_fwd"), bdelta = sym<attn_fn>(lib, "lwm_attn_bwd_delta"),
            bdq = sym<attn_fn>(lib, "lwm_attn_bwd_dq"), bdkdv = sym<attn_fn>(lib, "lwm_attn_bwd_dkdv"),
            bfused = sym<attn_fn>(lib, "lwm_attn_bwd_fused");
    auto ws_bytes = sym<int64_t (*)(int32_t, int32_t, int32_t, int32_t, int64_t, int64_t, int32_t, int32_t)>(lib, "lwm_attn_bwd_fused_workspace_bytes");
    auto last_error = sym<const char* (*)(void)>(lib, "lwm_last_error");

    const int D = 128;
    const size_t n = (size_t)S * H * D;
    uint16_t *q, *k, *v, *dout, *out, *dq, *dk, *dv;
    float *lse, *delta, *dq_acc;
    void* ws;
    double* sums;
    for (uint16_t** p : {&q, &k, &v, &dout, &out, &dq, &dk, &dv}) CK(hipMalloc(p, n * 2));
    CK(hipMalloc(&lse, (size_t)H * S * 4));
    CK(hipMalloc(&delta, (size_t)H * S * 4));
    CK(hipMalloc(&dq_acc, n * 4));
    const int group = argc > 6 ? atoi(argv[6]) : 0;      // (batch*head) slices per fused launch (0 = all)
    const int64_t wsb = ws_bytes(1, H, S, S, 0, 0, 1, group);
    CK(hipMalloc(&ws, wsb));
    CK(hipMalloc(&sums, 64));
    fill_bf16<<<2048, 256>>>(q, n, 1u, 1.7f);
    fill_bf16<<<2048, 256>>>(k, n, 2u, 1.7f);
    fill_bf16<<<2048, 256>>>(v, n, 3u, 1.7f);
    fill_bf16<<<2048, 256>>>(dout, n, 4u, 1.7f);
    CK(hipDeviceSynchronize());

    LwmAttnArgs a;
    memset(&a, 0, sizeof(a));
    auto t4 = [&](void* p) { return LwmTensor4{p, (int64_t)n, (int64_t)H * D, (int64_t)D}; };
    a.q = t4(q); a.k = t4(k); a.v = t4(v); a.out = t4(out); a.dout = t4(dout);
    a.dq = t4(dq); a.dk = t4(dk); a.dv = t4(dv);
    a.lse = lse; a.delta = delta; a.dq_acc = dq_acc;
    a.B = 1; a.H = H; a.Sq = S; a.Sk = S; a.D = D;
    a.scale = 1.0f / sqrtf((float)D);
    a.causal = 1; a.final_out = 1; a.dq_final_out = 1;
    a.bwd_workspace = ws;
    a.bwd_workspace_bytes = wsb;
    if (fwd(&a, nullptr) != 0) {
        fprintf(stderr, "fwd: %s\n", last_error());
        return 2;
    }
    CK(hipDeviceSynchronize());

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_ms = [&](auto&& fn) {
        fn();  // warm-up
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) fn();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    auto checksum = [&](const uint16_t* p) {
        CK(hipMemset(sums, 0, 8));
        abs_sum_bf16<<<1024, 256>>>(p, n, sums);
        double h = 0;
        CK(hipMemcpy(&h, sums, 8, hipMemcpyDeviceToHost));
        return h / (double)n;
    };
    const bool do_fused = strcmp(what, "two") != 0, do_two = strcmp(what, "fused") != 0;
    float ms_fused = -1, ms_dkdv = -1, ms_dq = -1, ms_delta = -1;
    const float ms_fwd = time_ms([&] { fwd(&a, nullptr); });
    if (getenv("LWM_PROF_DUMP")) {      // -DLWM_PROF builds of the library report s_memtime laps of the forward through out_acc
        unsigned long long* prof = nullptr;
        CK(hipMalloc(&prof, 4 * 10 * 8));
        CK(hipMemset(prof, 0, 4 * 10 * 8));
        a.out_acc = (float*)prof;
        fwd(&a, nullptr);
        CK(hipDeviceSynchronize());
        unsigned long long h[40];
        CK(hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost));
        a.out_acc = nullptr;
        printf("forward, last q tile of head 0: cycles per tile iteration (phase1a, mask+toggle, phase2a, phase1b, mask, phase2b, dma wait, barrier | iterations, total)\n");
        for (int w = 0; w < 4; ++w) {
            const double n = h[w * 10 + 8] ? (double)h[w * 10 + 8] : 1.0;
            printf("  wave %d:", w);
            for (int i = 0; i < 8; ++i) printf(" %7.1f", (double)h[w * 10 + i] / n);
            printf(" | %llu %llu\n", h[w * 10 + 8], h[w * 10 + 9]);
        }
    }
    bdelta(&a, nullptr);           // delta is an input of every backward flavour
    double cs_fused = -1, cs_two = -1, cs_dk_f = -1, cs_dk_t = -1;
    uint16_t* dq_keep = nullptr;       // the fused dq, kept for the element-wise comparison with the two-kernel dq
    if (do_fused) {
        ms_fused = time_ms([&] {
            if (bfused(&a, nullptr) != 0) {
                fprintf(stderr, "fused: %s\n", last_error());
                exit(2);
            }
        });
        cs_fused = checksum(dq);
        cs_dk_f = checksum(dk);
        CK(hipMalloc(&dq_keep, n * 2));
        CK(hipMemcpy(dq_keep, dq, n * 2, hipMemcpyDeviceToDevice));
    }
    double max_diff = -1, max_ref = -1;
    if (do_two) {
        a.dq_acc_head_major = 0;
        ms_delta = time_ms([&] { bdelta(&a, nullptr); });
        ms_dkdv = time_ms([&] { bdkdv(&a, nullptr); });
        ms_dq = time_ms([&] { bdq(&a, nullptr); });
        cs_two = checksum(dq);
        cs_dk_t = checksum(dk);
        if (dq_keep) {
            CK(hipMemset(sums, 0, 16));
            max_diff_bf16<<<1024, 256>>>(dq_keep, dq, n, (unsigned long long*)sums);
            unsigned long long h[2];
            CK(hipMemcpy(h, sums, 16, hipMemcpyDeviceToHost));
            max_diff = __builtin_bit_cast(float, (uint32_t)h[0]);
            max_ref = __builtin_bit_cast(float, (uint32_t)h[1]);
        }
    }
    printf("%-40s S=%d H=%d fwd %.3f  fused %.3f ms  delta %.3f dkdv %.3f dq %.3f  |dq| fused %.6f two %.6f  "
           "max|dq_f - dq_2| %.3e of %.3e  |dk| %.6f %.6f\n",
           argv[1], S, H, ms_fwd, ms_fused, ms_delta, ms_dkdv, ms_dq, cs_fused, cs_two, max_diff, max_ref, cs_dk_f, cs_dk_t);
    return 0;
}
