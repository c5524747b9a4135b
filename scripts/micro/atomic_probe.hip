// atomic_probe.hip -- what a fire-and-forget global_atomic_add_f32 costs on MI355X and where it is performed.
//   atomic_probe [tiles=2048]
// Leg A (semantics): every workgroup of a 1024-block grid adds 1.0 to the same 64 KiB of floats, with no scope
//   bits and with sc1.  With per-XCD L2s that are not coherent with each other, adds performed in the issuing
//   XCD's L2 lose updates when several XCDs hit the same word; adds performed at the memory side do not.
// Leg B (throughput, the attn_bwd_fused pattern): 256 persistent workgroups of 512 threads; the workgroups of
//   XCD x (claimed by XCC id, as the kernel does) add 16 KiB tiles (32 rows x 512 B; a 16-lane group = 64
//   contiguous bytes) into region x only, walking the region's tiles downwards, 32 workgroups per region at
//   staggered positions.  Reports GB/s of adds and checks every word.
// Build: hipcc -O2 --offload-arch=gfx950 -o scripts/micro/atomic_probe scripts/micro/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            exit(2);                                                          \
        }                                                                     \
    } while (0)

template <int SC1>
__device__ __forceinline__ void add_f32(float* base, uint32_t voff, float v) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    // (s_nop: the "VALU writes SGPR -> VMEM reads it" hazard is invisible to hipcc through an asm statement)
    if (SC1) asm volatile("s_nop 4\n\tglobal_atomic_add_f32 %0, %1, %2 sc1" ::"v"(voff), "v"(v), "s"(u) : "memory");
    else asm volatile("s_nop 4\n\tglobal_atomic_add_f32 %0, %1, %2" ::"v"(voff), "v"(v), "s"(u) : "memory");
}

// MODE 0: global_atomic_add_x2 (u64 integer add: two packed 32-bit fixed-point lanes), 1: global_atomic_add_f64,
// 2: global_atomic_pk_add_bf16 (4 bytes per op)
template <int MODE>
__device__ __forceinline__ void add_wide(void* base, uint32_t voff, uint64_t v) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    if (MODE == 0) asm volatile("s_nop 4\n\tglobal_atomic_add_x2 %0, %1, %2" ::"v"(voff), "v"(v), "s"(u) : "memory");
    else if (MODE == 1) asm volatile("s_nop 4\n\tglobal_atomic_add_f64 %0, %1, %2" ::"v"(voff), "v"(v), "s"(u) : "memory");
    else asm volatile("s_nop 4\n\tglobal_atomic_pk_add_bf16 %0, %1, %2" ::"v"(voff), "v"((uint32_t)v), "s"(u) : "memory");
}

// leg C: leg B's walk with 8-byte operations: a tile is still 16 KiB (32 rows x 512 B) but a lane issues 4 adds of
// 8 bytes (a 16-lane group = 128 contiguous bytes) instead of 8 adds of 4 bytes.  MODE 2 (packed bf16, 4-byte ops)
// issues 8 per lane like leg B.
template <int MODE, int WORK>
__global__ __launch_bounds__(512) void leg_c(uint64_t* buf, int tiles, int* rank_ctr, float* sink) {
    __shared__ int s_rank, s_x;
    if (threadIdx.x == 0) {
        int x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        s_x = x & 7;
        s_rank = atomicAdd(rank_ctr + s_x, 1);
    }
    __syncthreads();
    const int x = s_x, rank = s_rank;
    uint64_t* region = buf + (size_t)x * tiles * 2048;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kg = lane >> 4, db = wave & 3, qh = wave >> 2;
    int t = (tiles - 1 - 2 * rank) % tiles;
    if (t < 0) t += tiles;
    float acc = 0.f;
    const uint64_t one = MODE == 0 ? ((1ull << 32) | 1ull) : MODE == 1 ? 0x3ff0000000000000ull : 0x3f803f80ull;
    for (int s = 0; s < tiles; ++s) {
        uint64_t* tile = region + (size_t)t * 2048;
        if (MODE == 2) {
            uint32_t voff = (uint32_t)(16 * qh + 4 * kg) * 512u + (uint32_t)(32 * db + i) * 4u;
            for (int r = 0; r < 4; ++r) {
                add_wide<2>(tile, voff, one);
                add_wide<2>(tile, voff + 64, one);
                voff += 512u;
            }
        } else {
            uint32_t voff = (uint32_t)(16 * qh + 4 * kg) * 512u + (uint32_t)(16 * db + i) * 8u;
            for (int r = 0; r < 4; ++r) {
                add_wide<MODE>(tile, voff, one);
                voff += 512u;
            }
        }
        for (int w = 0; w < WORK; ++w) acc = __builtin_fmaf(acc, 1.000001f, 0.5f);
        t = t == 0 ? tiles - 1 : t - 1;
    }
    if (acc == 12345.f) *sink = acc;
}

template <int SC1>
__global__ __launch_bounds__(256) void leg_a(float* buf, int n) {
    for (int i = threadIdx.x; i < n; i += 256) add_f32<SC1>(buf, (uint32_t)i * 4u, 1.0f);
}

__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7;
}

// region r: `tiles` tiles of 4096 floats.  Workgroup k of an XCD (rank by arrival) starts at tile (tiles-1-stagger*k)
// and walks down `steps` tiles (wrapping), adding 1.0 everywhere: every word of region r ends at (#wg of that XCD
// that covered it).  To keep the check simple every workgroup covers ALL tiles once (steps = tiles).
template <int SC1, int WORK>
__global__ __launch_bounds__(512) void leg_b(float* buf, int tiles, int* rank_ctr, int* wg_count, float* sink) {
    __shared__ int s_rank, s_x;
    if (threadIdx.x == 0) {
        s_x = xcc_id();
        s_rank = atomicAdd(rank_ctr + s_x, 1);
    }
    __syncthreads();
    const int x = s_x, rank = s_rank;
    if (threadIdx.x == 0) atomicAdd(wg_count + x, 0);
    float* region = buf + (size_t)x * tiles * 4096;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kg = lane >> 4, db = wave & 3, qh = wave >> 2;
    int t = (tiles - 1 - 2 * rank) % tiles;
    if (t < 0) t += tiles;
    float acc = 0.f;
    for (int s = 0; s < tiles; ++s) {
        float* tile = region + (size_t)t * 4096;
        uint32_t voff = (uint32_t)(16 * qh + 4 * kg) * 512u + (uint32_t)(32 * db + i) * 4u;
        for (int r = 0; r < 4; ++r) {
            add_f32<SC1>(tile, voff, 1.0f);
            add_f32<SC1>(tile, voff + 64, 1.0f);
            voff += 512u;
        }
        // stand-in for the MFMA work of a step: WORK dependent FMAs per thread
        for (int w = 0; w < WORK; ++w) acc = __builtin_fmaf(acc, 1.000001f, 0.5f);
        t = t == 0 ? tiles - 1 : t - 1;
    }
    if (acc == 12345.f) *sink = acc;
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 2048;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // ---- leg A
    {
        const int n = 16384;
        float* buf;
        CK(hipMalloc(&buf, n * 4));
        std::vector<float> h(n);
        for (int sc1 = 0; sc1 < 2; ++sc1) {
            CK(hipMemset(buf, 0, n * 4));
            if (sc1) leg_a<1><<<1024, 256>>>(buf, n);
            else leg_a<0><<<1024, 256>>>(buf, n);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), buf, n * 4, hipMemcpyDeviceToHost));
            float mn = 1e30f, mx = -1e30f;
            for (float v : h) {
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
            }
            printf("leg A  %-8s 1024 workgroups x +1.0 on the same %d floats: min %.0f max %.0f (1024 = every add landed)\n",
                   sc1 ? "sc1" : "no-scope", n, mn, mx);
        }
        CK(hipFree(buf));
    }
    // ---- leg B
    {
        const size_t n = (size_t)8 * tiles * 4096;
        float *buf, *sink;
        int *ctr;
        CK(hipMalloc(&buf, n * 4));
        CK(hipMalloc(&sink, 4));
        CK(hipMalloc(&ctr, 64));
        std::vector<float> h(n);
        auto run = [&](int sc1, int work) {
            CK(hipMemset(buf, 0, n * 4));
            CK(hipMemset(ctr, 0, 64));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (sc1 && work) leg_b<1, 1500><<<256, 512>>>(buf, tiles, ctr, ctr + 8, sink);
            else if (sc1) leg_b<1, 0><<<256, 512>>>(buf, tiles, ctr, ctr + 8, sink);
            else if (work) leg_b<0, 1500><<<256, 512>>>(buf, tiles, ctr, ctr + 8, sink);
            else leg_b<0, 0><<<256, 512>>>(buf, tiles, ctr, ctr + 8, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            int ranks[8];
            CK(hipMemcpy(ranks, ctr, 32, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h.data(), buf, n * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (int x = 0; x < 8; ++x)
                for (size_t j = 0; j < (size_t)tiles * 4096; ++j) bad += h[(size_t)x * tiles * 4096 + j] != (float)ranks[x];
            const double bytes = 256.0 * tiles * 16384.0;
            printf("leg B  %-8s %s  %d tiles x 16 KiB x 256 workgroups = %.2f GB of adds in %.3f ms = %.2f TB/s; wrong words %zu; wg per XCD %d %d %d %d %d %d %d %d\n",
                   sc1 ? "sc1" : "no-scope", work ? "with ~6k-cycle filler per tile" : "adds only              ", tiles, bytes / 1e9, ms,
                   bytes / ms / 1e9, bad, ranks[0], ranks[1], ranks[2], ranks[3], ranks[4], ranks[5], ranks[6], ranks[7]);
        };
        for (int work = 0; work < 2; ++work)
            for (int sc1 = 0; sc1 < 2; ++sc1) run(sc1, work);
    }
    // ---- leg C: 8-byte operations on the same walk
    {
        const size_t n = (size_t)8 * tiles * 2048;
        uint64_t* buf;
        float* sink;
        int* ctr;
        CK(hipMalloc(&buf, n * 8));
        CK(hipMalloc(&sink, 4));
        CK(hipMalloc(&ctr, 64));
        std::vector<uint64_t> h(n);
        auto run = [&](int mode, int work) {
            CK(hipMemset(buf, 0, n * 8));
            CK(hipMemset(ctr, 0, 64));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (mode == 0 && !work) leg_c<0, 0><<<256, 512>>>(buf, tiles, ctr, sink);
            if (mode == 0 && work) leg_c<0, 1500><<<256, 512>>>(buf, tiles, ctr, sink);
            if (mode == 1 && !work) leg_c<1, 0><<<256, 512>>>(buf, tiles, ctr, sink);
            if (mode == 1 && work) leg_c<1, 1500><<<256, 512>>>(buf, tiles, ctr, sink);
            if (mode == 2 && !work) leg_c<2, 0><<<256, 512>>>(buf, tiles, ctr, sink);
            if (mode == 2 && work) leg_c<2, 1500><<<256, 512>>>(buf, tiles, ctr, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            int ranks[8];
            CK(hipMemcpy(ranks, ctr, 32, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h.data(), buf, n * 8, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (int x = 0; x < 8; ++x)
                for (size_t j = 0; j < (size_t)tiles * 2048; ++j) {
                    const uint64_t w = h[(size_t)x * tiles * 2048 + j];
                    uint64_t want;
                    if (mode == 0) want = ((uint64_t)ranks[x] << 32) | (uint64_t)ranks[x];
                    else if (mode == 1) { double d = ranks[x]; memcpy(&want, &d, 8); }
                    else { float f = (float)ranks[x]; uint32_t b; memcpy(&b, &f, 4); b >>= 16; want = ((uint64_t)((b << 16) | b) << 32) | ((b << 16) | b); }
                    bad += w != want;
                }
            const double bytes = 256.0 * tiles * 16384.0;
            const char* nm = mode == 0 ? "add_x2 (2 x i32 fixed point)" : mode == 1 ? "add_f64" : "pk_add_bf16 (4-byte ops)";
            printf("leg C  %-30s %s  %.2f GB of adds in %.3f ms = %.2f TB/s; wrong words %zu\n", nm,
                   work ? "with ~6k-cycle filler per tile" : "adds only              ", bytes / 1e9, ms, bytes / ms / 1e9, bad);
        };
        for (int work = 0; work < 2; ++work)
            for (int mode = 0; mode < 3; ++mode) run(mode, work);
    }
    return 0;
}
