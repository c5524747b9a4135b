// attn_bench.cpp -- times the attention kernels of one layer through the C ABI, without Python:
//   attn_bench <liblwm_hip.so> [S=32768] [H=32] [reps=3]
// One line per library: HIP-event ms per launch of lwm_attn_fwd, lwm_attn_bwd_delta, _dkdv and _dq, and mean |x| checksums
// of dq / dk / dv, so that a timing variant that breaks a result is visible.  scripts/gpu_ab.sh sweeps variant builds
// (scripts/ab_build.sh) with it in one GPU call.  LWM_PROF_DUMP=1 with a -DLWM_PROF build prints the s_memtime laps of
// the dK/dV kernel (attn_bwd64.h), LWM_PROF_DUMP=-1 those of the forward (attn_fwd64.h).
// Build: hipcc -O2 --offload-arch=gfx950 -I include -o scripts/micro/attn_bench scripts/micro/attn_bench.cpp -ldl
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
        fprintf(stderr, "usage: attn_bench <lib> [S] [H] [reps]\n");
        return 2;
    }
    const int S = argc > 2 ? atoi(argv[2]) : 32768, H = argc > 3 ? atoi(argv[3]) : 32, reps = argc > 4 ? atoi(argv[4]) : 3;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    typedef int (*attn_fn)(const LwmAttnArgs*, void*);
    attn_fn fwd = sym<attn_fn>(lib, "lwm_attn_fwd"), bdelta = sym<attn_fn>(lib, "lwm_attn_bwd_delta"),
            bdq = sym<attn_fn>(lib, "lwm_attn_bwd_dq"), bdkdv = sym<attn_fn>(lib, "lwm_attn_bwd_dkdv");
    auto stat_bytes = sym<int64_t (*)(int32_t, int32_t, int32_t)>(lib, "lwm_attn_bwd_delta_bytes");
    auto last_error = sym<const char* (*)(void)>(lib, "lwm_last_error");

    const int D = 128;
    const size_t n = (size_t)S * H * D;
    uint16_t *q, *k, *v, *dout, *out, *dq, *dk, *dv;
    float *lse, *delta;
    double* sums;
    for (uint16_t** p : {&q, &k, &v, &dout, &out, &dq, &dk, &dv}) CK(hipMalloc(p, n * 2));
    CK(hipMalloc(&lse, (size_t)H * S * 4));
    CK(hipMalloc(&delta, (size_t)stat_bytes(1, H, S)));
    CK(hipMalloc(&sums, 64));
    // LWM_BENCH_AMP=0: all-zero operands (the matrix pipe then draws less power and the chip clocks higher: the gap to
    // the random-data time is what DVFS costs, MI355X_MICROARCH.md "DVFS give-back")
    const float amp = getenv("LWM_BENCH_AMP") ? (float)atof(getenv("LWM_BENCH_AMP")) : 1.7f;
    fill_bf16<<<2048, 256>>>(q, n, 1u, amp);
    fill_bf16<<<2048, 256>>>(k, n, 2u, amp);
    fill_bf16<<<2048, 256>>>(v, n, 3u, amp);
    fill_bf16<<<2048, 256>>>(dout, n, 4u, amp);
    CK(hipDeviceSynchronize());

    LwmAttnArgs a;
    memset(&a, 0, sizeof(a));
    auto t4 = [&](void* p) { return LwmTensor4{p, (int64_t)n, (int64_t)H * D, (int64_t)D}; };
    a.q = t4(q); a.k = t4(k); a.v = t4(v); a.out = t4(out); a.dout = t4(dout);
    a.dq = t4(dq); a.dk = t4(dk); a.dv = t4(dv);
    a.lse = lse; a.delta = delta;
    a.B = 1; a.H = H; a.Sq = S; a.Sk = S; a.D = D;
    a.scale = 1.0f / sqrtf((float)D);
    a.causal = 1; a.final_out = 1;
    auto must = [&](int rc, const char* what) {
        if (rc != 0) {
            fprintf(stderr, "%s: %s\n", what, last_error());
            exit(2);
        }
    };
    must(fwd(&a, nullptr), "fwd");
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
    // -DLWM_PROF builds of the library report s_memtime laps through out_acc (unused by these launches)
    const int dump = getenv("LWM_PROF_DUMP") ? atoi(getenv("LWM_PROF_DUMP")) : 0;
    auto laps = [&](attn_fn fn, const char* title, int slots, int n_lap, int it_slot) {
        unsigned long long* prof = nullptr;
        CK(hipMalloc(&prof, 4 * slots * 8));
        CK(hipMemset(prof, 0, 4 * slots * 8));
        a.out_acc = (float*)prof;
        must(fn(&a, nullptr), title);
        CK(hipDeviceSynchronize());
        unsigned long long h[64];
        CK(hipMemcpy(h, prof, 4 * slots * 8, hipMemcpyDeviceToHost));
        a.out_acc = nullptr;
        printf("%s\n", title);
        for (int w = 0; w < 4; ++w) {
            const double cnt = h[w * slots + it_slot] ? (double)h[w * slots + it_slot] : 1.0;
            double sum = 0;
            printf("  wave %d:", w);
            for (int i = 0; i < n_lap; ++i) {
                printf(" %7.1f", (double)h[w * slots + i] / cnt);
                sum += (double)h[w * slots + i] / cnt;
            }
            printf(" = %7.1f | %llu %llu", sum, h[w * slots + it_slot], h[w * slots + it_slot + 1]);
            if (it_slot + 4 < slots)      // (dK/dV: the whole workgroup)
                printf(" | %llu %llu %llu", h[w * slots + it_slot + 2], h[w * slots + it_slot + 3], h[w * slots + it_slot + 4]);
            printf("\n");
        }
        CK(hipFree(prof));
    };
    const float ms_fwd = time_ms([&] { must(fwd(&a, nullptr), "fwd"); });
    if (dump < 0)
        laps(fwd, "forward, last q tile of head 0: cycles per tile iteration (phase1a, mask+toggle, phase2a, phase1b, mask, phase2b, dma wait, barrier | iterations, total)", 10, 8, 8);
    const float ms_delta = time_ms([&] { must(bdelta(&a, nullptr), "delta"); });
    if (dump > 0)
        laps(bdkdv, "dK/dV, key block 0 of head 0: cycles per step (X0, mask, Y0, X1, mask, Y1, wait, barrier, addresses | steps, cycles of the pipelined steps | entry->stores done, entry->first step, entry->epilogue)", 16, 9, 9);
    const float ms_dkdv = time_ms([&] { must(bdkdv(&a, nullptr), "dkdv"); });
    const float ms_dq = time_ms([&] { must(bdq(&a, nullptr), "dq"); });
    printf("%-34s S=%d H=%d  fwd %.3f  delta %.3f  dkdv %.3f  dq %.3f ms   |dq| %.6f |dk| %.6f |dv| %.6f\n", argv[1], S, H, ms_fwd,
           ms_delta, ms_dkdv, ms_dq, checksum(dq), checksum(dk), checksum(dv));
    return 0;
}
