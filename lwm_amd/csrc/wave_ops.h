// wave_ops.h -- the gfx950 (CDNA4) device vocabulary every kernel in this
// directory is written against: wave64 lane ids, the 32x32x16 bf16 MFMA, the
// LDS transpose read, half-wave exchange, block barrier, fast exp2.
//
// Kernels never spell a __builtin_amdgcn_* themselves; they call these
// wrappers.  tests/emu/wave_ops.h provides the same names on the host (fibers,
// one per lane) so kernel index math can be exercised without a GPU.  That
// header is test infrastructure; this one is the product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lwm {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define LWM_DEVICE __device__ __forceinline__
#define LWM_HD __host__ __device__ inline
#define LWM_GLOBAL __global__
#define LWM_KERNEL(max_threads) __global__ __launch_bounds__(max_threads)
// second argument = minimum waves per SIMD the register allocation must allow
#define LWM_KERNEL_OCC(max_threads, waves_per_simd) __global__ __launch_bounds__(max_threads, waves_per_simd)

LWM_DEVICE int thread_idx() { return (int)threadIdx.x; }
LWM_DEVICE int block_idx_x() { return (int)blockIdx.x; }
LWM_DEVICE int block_idx_y() { return (int)blockIdx.y; }
LWM_DEVICE int block_idx_z() { return (int)blockIdx.z; }
LWM_DEVICE int grid_dim_x() { return (int)gridDim.x; }

// LDS is addressed with 32-bit byte addresses (lds_t), never generic 64-bit
// pointers: loop-invariant fragment addresses then cost one VGPR each and the
// per-tile constants fold into the ds_read/ds_write immediate offset field.
typedef uint32_t lds_t;
#define LWM_LDS(T, a) ((__attribute__((address_space(3))) T*)(uintptr_t)(a))

// Dynamic LDS base (16-byte aligned: no static __shared__ anywhere, see
// cdna_hip_programming.md Guideline 17).
LWM_DEVICE lds_t dyn_lds() {
    extern __shared__ __attribute__((aligned(16))) char lwm_smem[];
    return (lds_t)(uintptr_t)((__attribute__((address_space(3))) char*)lwm_smem);
}

LWM_DEVICE void block_sync() { __syncthreads(); }
// Workgroup barrier that orders LDS traffic ONLY: __syncthreads() carries a workgroup-scope release fence,
// which on gfx950 is an s_waitcnt vmcnt(0) -- every vector-memory operation still in flight is waited for at
// every barrier.  A kernel that keeps LDS-DMA pieces in flight across its step barrier behind a counted vmcnt
// (attn_bwd64.h) uses this one: LDS operations retired (lgkmcnt(0)), s_barrier.
LWM_DEVICE void block_sync_lds() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0); vmcnt and expcnt left alone
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Orders one wave's LDS accesses among its own lanes (a wave's private staging area: written by some lanes, read by
// others).  The LDS queue of a wave is in order, so this is a compiler fence only.
LWM_DEVICE void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// D = A(32x16) * B(16x32) + C(32x32), bf16 in / f32 accumulate.
//   A: lane l holds A[row = l&31][k = 8*(l>>5) + j], j = 0..7
//   B: lane l holds B[k = 8*(l>>5) + j][col = l&31]
//   C/D: lane l, reg r holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
LWM_DEVICE f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ds_read_b64_tr_b16.  Per 16-lane group: lanes 4j..4j+3 each point at 4
// consecutive bf16 of "row j" (so a group addresses a 4x16 block, rows
// anywhere); lane i receives column i of that block: {row0[i], row1[i],
// row2[i], row3[i]}.  `a` must be an 8-byte aligned LDS address.
LWM_DEVICE bf16x4 lds_read_tr16(lds_t a) {
    typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4;
    v4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LWM_LDS(v4, a));
    bf16x4 o;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
    return o;
}

// D = A(32x2) * B(2x32) + C(32x32), f32 in / f32 accumulate, EXACT f32: per
// element D = fmaf(A[i][1], B[1][j], fmaf(A[i][0], B[0][j], C)) (one rounding per
// product, k in order -- MI355X_MICROARCH.md "FP32-input MFMA"; confirmed on
// hardware by tests/test_gpu_probe.py).  64 cycles per instruction per SIMD.
//   A: lane l holds A[row = l&31][k = l>>5];  B: lane l holds B[k = l>>5][col = l&31]
//   C/D: as the bf16 32x32 form.
LWM_DEVICE f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

LWM_DEVICE bf16x8 lds_read_b128(lds_t a) { return *LWM_LDS(const bf16x8, a); }
LWM_DEVICE f32x4 lds_read_f32x4(lds_t a) { return *LWM_LDS(const f32x4, a); }
LWM_DEVICE u32x4 lds_read_u32x4(lds_t a) { return *LWM_LDS(const u32x4, a); }
LWM_DEVICE void lds_write_b128(lds_t a, u32x4 v) { *LWM_LDS(u32x4, a) = v; }
LWM_DEVICE void lds_write_b64(lds_t a, u32x2 v) { *LWM_LDS(u32x2, a) = v; }
LWM_DEVICE void lds_write_f32x4(lds_t a, f32x4 v) { *LWM_LDS(f32x4, a) = v; }
LWM_DEVICE void lds_write_i32(lds_t a, int32_t v) { *LWM_LDS(int32_t, a) = v; }
LWM_DEVICE void lds_write_f32(lds_t a, float v) { *LWM_LDS(float, a) = v; }
LWM_DEVICE float lds_read_f32(lds_t a) { return *LWM_LDS(const float, a); }
LWM_DEVICE int32_t lds_read_i32(lds_t a) { return *LWM_LDS(const int32_t, a); }
LWM_DEVICE void lds_write_f64(lds_t a, double v) { *LWM_LDS(double, a) = v; }
LWM_DEVICE double lds_read_f64(lds_t a) { return *LWM_LDS(const double, a); }

// Direct global -> LDS copy (global_load_lds_dwordx4): lane l's 16 bytes at `g`
// land at LDS address wave_base + 16*l.  `wave_base` must be wave-uniform (an
// SGPR: kernel args, blockIdx, wave_uniform()).  No VGPR round trip.
// Issued from inline asm ON PURPOSE: hipcc does not count it, so it inserts no
// s_waitcnt vmcnt(0) in front of later ds_reads of the OTHER tile buffer (with the
// builtin it drains the DMA at the first transposed read of every tile).  The
// kernel owns the wait: glds_wait_all() by every wave, then block_sync(), before
// anyone reads the destination (cdna_hip_programming.md section 5.7 item 1).
LWM_DEVICE void glds_load_b128(const void* g, lds_t wave_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(wave_base)
        : "memory");
}
// 4-byte form: lane l's dword at `g` lands at wave_base + 4*l (row statistics: no VGPR, no ds_write,
// and -- being asm -- no compiler-visible load whose retirement hipcc would schedule for us).
LWM_DEVICE void glds_load_b32(const void* g, lds_t wave_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(wave_base)
        : "memory");
}
// wait until at most N of this wave's vector-memory operations are outstanding (they retire in issue order)
template <int N>
LWM_DEVICE void wait_vmem_le() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");   // asm: hipcc neither merges nor drops it
}
LWM_DEVICE void glds_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Make a value that IS the same in every lane provably uniform (SGPR).
LWM_DEVICE int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Make a value opaque to the optimiser at this point (no instruction is emitted):
// address arithmetic derived from it cannot be hoisted out of the enclosing loop
// (hipcc hoists such loop invariants and then spills them under register pressure).
LWM_DEVICE uint32_t opaque(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}
// Scheduling fence: the compiler may not move instructions across it.
LWM_DEVICE void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// Raise/lower this wave's issue priority around an MFMA cluster (T5).
#ifndef LWM_PRIO_MODE
#define LWM_PRIO_MODE 1      // 1: matrix phases at priority 1; 0: no priorities; 2: vector phases at priority 1
#endif
LWM_DEVICE void prio_hi() {
    if (LWM_PRIO_MODE == 1) __builtin_amdgcn_s_setprio(1);
    if (LWM_PRIO_MODE == 2) __builtin_amdgcn_s_setprio(0);
}
LWM_DEVICE void prio_lo() {
    if (LWM_PRIO_MODE == 1) __builtin_amdgcn_s_setprio(0);
    if (LWM_PRIO_MODE == 2) __builtin_amdgcn_s_setprio(1);
}

// value held by lane (l ^ 32)
LWM_DEVICE float xhalf(float x) {
    return __shfl_xor(x, 32, 64);
}
// lane `src_lane`'s x for every lane, as a scalar operand (v_readlane_b32; src_lane wave-uniform)
LWM_DEVICE float lane_value(float x, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
}
LWM_DEVICE float shfl_xor_f(float x, int m) { return __shfl_xor(x, m, 64); }
LWM_DEVICE int shfl_xor_i(int x, int m) { return __shfl_xor(x, m, 64); }

// true if the predicate holds in ANY lane of the wave (wave-uniform result)
LWM_DEVICE bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0; }

LWM_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
LWM_DEVICE float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

LWM_DEVICE u32x4 global_load_b128(const void* p) { return *(const u32x4*)p; }
LWM_DEVICE void global_store_b128(void* p, u32x4 v) { *(u32x4*)p = v; }
LWM_DEVICE void global_store_b64(void* p, u32x2 v) { *(u32x2*)p = v; }
LWM_DEVICE f32x4 global_load_f32x4(const float* p) { return *(const f32x4*)p; }
LWM_DEVICE void global_store_f32x4(float* p, f32x4 v) { *(f32x4*)p = v; }

// XCC id of the executing CU (0..7): the truth about placement, not blockIdx % 8 (tests/test_gpu_probe.py checks that
// the two agree closely enough for the kernels' "one head per XCD" block mappings to share an L2)
LWM_DEVICE int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}
LWM_DEVICE int atomic_add_i32(int32_t* p, int32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one float at base + voff + soff bytes (base and soff wave-uniform: soff rides in an SGPR, the address costs
// no VALU); plain cache policy
LWM_DEVICE float global_load_f32_at(const float* base, uint32_t voff, uint32_t soff) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// a float every lane of the wave reads from the same address of read-only memory: a scalar load (s_load_dword, the
// value lands in an SGPR and can be an operand of a vector instruction directly)
LWM_DEVICE float uniform_load_f32(const float* p, int idx) {
    return ((const __attribute__((address_space(4))) float*)p)[idx];
}

LWM_DEVICE uint32_t pack_bf16x2(float lo, float hi) {
    union { bf16_t h[2]; uint32_t u; } x;
    x.h[0] = (bf16_t)lo;
    x.h[1] = (bf16_t)hi;
    return x.u;
}

}  // namespace lwm
