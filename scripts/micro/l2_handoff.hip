// l2_handoff.hip -- does a read-modify-write stream handed from workgroup to workgroup inside ONE XCD
// stay in that XCD's L2?  (The question behind the fused backward's dq accumulation)
//
// 256 workgroups (one per CU).  The workgroups of an XCD form a chain in arrival order; link i adds 1.0
// to every float of tiles 0..T-1 (16 KiB each) of the XCD's region, tile t only after link i-1 has
// published "t done" -- the same protocol as attn_bwd_fused_kernel.  Every variant is its own kernel
// name, so one rocprofv3 --pmc pass gives FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS per variant.
//
// Build: hipcc -O3 --offload-arch=gfx950 -o l2_handoff l2_handoff.hip        Run: ./l2_handoff [tiles]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define DEV __device__ __forceinline__

DEV int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}
template <int AUX>
DEV f32x4 buf_load(const float* base, uint32_t off) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX));
}
DEV void store_plain(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
DEV void store_sc1(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
DEV void store_nt(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }
DEV void store_flag(int* p, int v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
DEV int poll_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV int poll_inv(const int* p) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: this CU's L1 only
    int v;
    asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

constexpr int kTileFloats = 4096;  // 16 KiB
constexpr int kSpin = 1 << 22;

// LOAD: 0 plain, 1 sc1 (aux 16), 2 nt (aux 2).  STORE: 0 plain, 1 sc1, 2 nt.  POLL: 0 sc1 load, 1 L1 invalidate + plain.
// FILL: s_sleep units (64 clocks each) of idle time per tile (the product's step is ~2 us = ~75 units).
// STREAM: every link also reads a 32 KiB slice per tile of a stream shared by the XCD (the Q/dO tiles).
// CHAIN: 1 = links wait for each other; 0 = every link runs free (a race by construction; traffic only).
template <int LOAD, int STORE, int POLL, int FILL, int STREAM, int CHAIN, int COUNTED = 0>
__global__ __launch_bounds__(256) void handoff(float* data, const float* stream, int* ctl, int tiles, float* sink) {
    __shared__ int s_pos;
    const int tid = threadIdx.x;
    const int x = __builtin_amdgcn_readfirstlane(xcc_id());
    if (tid == 0) s_pos = atomicAdd(ctl + x, 1);
    __syncthreads();
    const int i = s_pos;                           // position in this XCD's chain
    int* prog = ctl + 64 + x * 64;                 // prog[j] = tiles finished by link j
    float* region = data + (size_t)x * tiles * kTileFloats;
    const float* sreg = stream + (size_t)x * tiles * 2 * kTileFloats;
    f32x4 keep = {0, 0, 0, 0};
    for (int t = 0; t < tiles; ++t) {
        if (CHAIN && i > 0) {
            int n = 0;
            if (tid == 0) {
                while ((POLL ? poll_inv(prog + i - 1) : poll_sc1(prog + i - 1)) <= t && ++n < kSpin) __builtin_amdgcn_s_sleep(2);
                if (n >= kSpin) ctl[8] = 1;
            }
            __syncthreads();
        }
        float* tile = region + (size_t)t * kTileFloats;
        f32x4 v[4];
        for (int j = 0; j < 4; ++j) v[j] = buf_load<LOAD == 0 ? 0 : (LOAD == 1 ? 16 : 2)>(tile, (uint32_t)(j * 256 + tid) * 16);
        if (STREAM) {
            for (int j = 0; j < 8; ++j) keep += buf_load<0>(sreg + (size_t)t * 2 * kTileFloats, (uint32_t)(j * 256 + tid) * 16);
        }
        for (int j = 0; j < 4; ++j) v[j] += 1.0f;
        for (int j = 0; j < 4; ++j) {
            float* dst = tile + (j * 256 + tid) * 4;
            if (STORE == 0) store_plain(dst, v[j]);
            else if (STORE == 1) store_sc1(dst, v[j]);
            else store_nt(dst, v[j]);
        }
        if (FILL) __builtin_amdgcn_s_sleep(FILL);
        if (COUNTED) {
            // do vector-memory operations retire in issue order across stores and loads?  Four younger loads
            // (cold lines: the stream region, never touched before) follow the stores; the link publishes as
            // soon as at most those four are outstanding.  A store overtaken by them would be published early
            // and the next link would add to a stale tile ("wrong" > 0 in the report).
            f32x4 y[4];
            for (int j = 0; j < 4; ++j)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(y[j]) : "v"(sreg + ((size_t)t * 2 * kTileFloats + (j * 256 + tid) * 4)) : "memory");
            asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");   // (not __syncthreads: its fence waits vmcnt(0))
            if (tid == 0) store_flag(prog + i, t + 1);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3])::"memory");
            keep += y[0] + y[1] + y[2] + y[3];
            continue;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) store_flag(prog + i, t + 1);
    }
    if (keep[0] == 123.456f) sink[0] = keep[1];
}

struct Variant {
    const char* name;
    void (*fn)(float*, const float*, int*, int, float*);
    bool chain;
};

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 512;
    const size_t nfl = (size_t)8 * tiles * kTileFloats;
    float *data, *stream, *sink;
    int* ctl;
    hipMalloc(&data, nfl * 4);
    hipMalloc(&stream, nfl * 2 * 4);
    hipMalloc(&sink, 64);
    hipMalloc(&ctl, 4096);
    hipMemset(stream, 0, nfl * 2 * 4);
    const Variant vs[] = {
        {"plain_ld plain_st sc1_poll", handoff<0, 0, 0, 0, 0, 1>, true},
        {"sc1_ld   plain_st sc1_poll", handoff<1, 0, 0, 0, 0, 1>, true},
        {"nt_ld    plain_st sc1_poll", handoff<2, 0, 0, 0, 0, 1>, true},
        {"plain_ld plain_st inv_poll", handoff<0, 0, 1, 0, 0, 1>, true},
        {"sc1_ld   sc1_st   sc1_poll", handoff<1, 1, 0, 0, 0, 1>, true},
        {"sc1_ld   nt_st    sc1_poll", handoff<1, 2, 0, 0, 0, 1>, true},
        {"plain_ld plain_st sc1_poll fill75", handoff<0, 0, 0, 75, 0, 1>, true},
        {"sc1_ld   plain_st sc1_poll fill75", handoff<1, 0, 0, 75, 0, 1>, true},
        {"plain_ld plain_st sc1_poll fill75 stream", handoff<0, 0, 0, 75, 1, 1>, true},
        {"sc1_ld   plain_st sc1_poll fill75 stream", handoff<1, 0, 0, 75, 1, 1>, true},
        {"plain_ld plain_st free-running", handoff<0, 0, 0, 0, 0, 0>, false},
        {"sc1_ld   plain_st publish at vmcnt(4)", handoff<1, 0, 0, 0, 0, 1, 1>, true},
        {"sc1_ld   plain_st publish at vmcnt(4) fill75", handoff<1, 0, 0, 75, 0, 1, 1>, true},
    };
    std::vector<float> host(nfl);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("tiles per XCD %d (%.1f MiB per XCD region); RMW bytes per launch: %.2f GB read + the same written\n", tiles,
           tiles * 16384 / 1048576.0, 256.0 * tiles * 16384 / 1e9);
    for (const Variant& v : vs) {
        hipMemset(data, 0, nfl * 4);
        hipMemset(ctl, 0, 4096);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(v.fn, dim3(256), dim3(256), 0, 0, data, stream, ctl, tiles, sink);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        int ctlh[16];
        hipMemcpy(ctlh, ctl, sizeof(ctlh), hipMemcpyDeviceToHost);
        hipMemcpy(host.data(), data, nfl * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int x = 0; x < 8; ++x)
            for (size_t k = 0; k < (size_t)tiles * kTileFloats; ++k)
                bad += host[(size_t)x * tiles * kTileFloats + k] != (float)ctlh[x];
        printf("%-44s %8.3f ms  %6.2f us/tile  links/XCD %d..%d  timeout %d  wrong %zu%s\n", v.name, ms,
               ms * 1e3 / tiles, ctlh[0], ctlh[7], ctlh[8], bad, v.chain ? "" : " (race expected)");
    }
    return 0;
}
