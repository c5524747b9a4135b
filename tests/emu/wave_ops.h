// tests/emu/wave_ops.h -- HOST emulation of lwm_amd/csrc/wave_ops.h.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under lwm_amd/ includes this file.  It
// lets the kernel headers (attn_fwd.h, attn_bwd.h, ...) be compiled with the
// host clang++ and executed one fiber per lane, so that tile/fragment index
// math, masks and the online-softmax bookkeeping can be checked on a machine
// with no GPU.  The cross-lane instructions are emulated from their documented
// lane maps (see the comments in the product header); the first GPU run of the
// round confirms those maps with tests/test_gpu_probe.py.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <ucontext.h>
#include <vector>
#include <thread>
#include <atomic>
#include <functional>

namespace lwm {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define LWM_DEVICE static inline __attribute__((always_inline))
#define LWM_HD static inline
#define LWM_GLOBAL static
#define LWM_KERNEL(max_threads) static
#define LWM_KERNEL_OCC(max_threads, waves_per_simd) static
#define LWM_EMU 1
typedef uint32_t lds_t;

namespace emu {

struct Wave {
    bf16x8 a[64];
    bf16x8 b[64];
    const char* addr[64];
    float f[64];
    float fa[64], fb[64];
    int i[64];
    int arrived = 0;
    unsigned gen = 0;
};

struct Lane {
    ucontext_t ctx;
    char* stack = nullptr;
    int tid = 0;
    bool done = false;
};

struct Block {
    int nthreads = 0;
    int bx = 0, by = 0, bz = 0, gx = 0;
    char* lds = nullptr;
    size_t lds_bytes = 0;
    std::vector<Wave> waves;
    std::vector<Lane> lanes;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    long progress = 0;
    const std::function<void()>* body = nullptr;
};

inline thread_local Block* g_blk = nullptr;
inline thread_local Lane* g_lane = nullptr;
inline thread_local ucontext_t g_sched;

inline void yield() { swapcontext(&g_lane->ctx, &g_sched); }

inline void die(const char* msg) {
    fprintf(stderr, "[lwm-emu] FATAL: %s\n", msg);
    abort();
}

inline void wave_sync() {
    Wave& w = g_blk->waves[g_lane->tid >> 6];
    unsigned g = w.gen;
    if (++w.arrived == 64) {
        w.arrived = 0;
        w.gen++;
        g_blk->progress++;
    } else {
        while (w.gen == g) yield();
    }
}

inline void block_barrier() {
    Block& b = *g_blk;
    unsigned g = b.bar_gen;
    if (++b.bar_arrived == b.nthreads) {
        b.bar_arrived = 0;
        b.bar_gen++;
        b.progress++;
    } else {
        while (b.bar_gen == g) yield();
    }
}

inline void lane_entry() {
    Block* b = g_blk;
    (*b->body)();
    g_lane->done = true;
    b->progress++;
    swapcontext(&g_lane->ctx, &g_sched);
}

// LDS byte address -> host pointer, with bounds and alignment checks.  The
// emulated LDS base is a non-zero address (4096) so that code forgetting to add
// the dynamic-LDS base is caught.
constexpr uint32_t kLdsBase = 4096;
inline char* lds_ptr(lds_t a, size_t n, size_t align) {
    Block& b = *g_blk;
    if (a < kLdsBase || (size_t)(a - kLdsBase) + n > b.lds_bytes) die("LDS access out of bounds");
    if ((a - kLdsBase) % align) die("LDS access misaligned");
    return b.lds + (a - kLdsBase);
}

// Runs one block to completion on the calling OS thread.
inline void run_block(int bx, int by, int bz, int gx, int nthreads, size_t lds_bytes,
                      const std::function<void()>& body) {
    if (nthreads % 64) die("block size must be a multiple of 64");
    Block blk;
    blk.nthreads = nthreads;
    blk.bx = bx; blk.by = by; blk.bz = bz; blk.gx = gx;
    blk.lds_bytes = lds_bytes;
    blk.lds = (char*)aligned_alloc(256, lds_bytes + 256);
    memset(blk.lds, 0xCD, lds_bytes + 256);  // poison: uninitialised reads show up
    blk.waves.resize(nthreads / 64);
    blk.lanes.resize(nthreads);
    blk.body = &body;
    const size_t STK = 256 * 1024;
    g_blk = &blk;
    for (int t = 0; t < nthreads; ++t) {
        Lane& ln = blk.lanes[t];
        ln.tid = t;
        ln.stack = (char*)malloc(STK);
        getcontext(&ln.ctx);
        ln.ctx.uc_stack.ss_sp = ln.stack;
        ln.ctx.uc_stack.ss_size = STK;
        ln.ctx.uc_link = &g_sched;
        makecontext(&ln.ctx, (void (*)())lane_entry, 0);
    }
    int remaining = nthreads;
    long stall_rounds = 0;
    while (remaining > 0) {
        long before = blk.progress;
        remaining = 0;
        for (int t = 0; t < nthreads; ++t) {
            Lane& ln = blk.lanes[t];
            if (ln.done) continue;
            g_lane = &ln;
            swapcontext(&g_sched, &ln.ctx);
            if (!ln.done) remaining++;
        }
        if (blk.progress == before) {
            if (++stall_rounds > 4) die("deadlock: lanes waiting at a barrier nobody else reaches");
        } else {
            stall_rounds = 0;
        }
    }
    for (auto& ln : blk.lanes) free(ln.stack);
    free(blk.lds);
    g_blk = nullptr;
    g_lane = nullptr;
}

struct Dim3 { int x, y, z; };

// Runs the whole grid; blocks are distributed over host threads.
inline void launch(Dim3 grid, int nthreads, size_t lds_bytes, const std::function<void()>& body) {
    long total = (long)grid.x * grid.y * grid.z;
    std::atomic<long> next{0};
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if ((long)nt > total) nt = (unsigned)total;
    auto worker = [&]() {
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= total) break;
            int bx = (int)(i % grid.x);
            int by = (int)((i / grid.x) % grid.y);
            int bz = (int)(i / ((long)grid.x * grid.y));
            run_block(bx, by, bz, grid.x, nthreads, lds_bytes, body);
        }
    };
    if (nt <= 1) { worker(); return; }
    std::vector<std::thread> ths;
    for (unsigned t = 0; t < nt; ++t) ths.emplace_back(worker);
    for (auto& t : ths) t.join();
}

}  // namespace emu

LWM_DEVICE int thread_idx() { return emu::g_lane->tid; }
LWM_DEVICE int block_idx_x() { return emu::g_blk->bx; }
LWM_DEVICE int block_idx_y() { return emu::g_blk->by; }
LWM_DEVICE int block_idx_z() { return emu::g_blk->bz; }
LWM_DEVICE int grid_dim_x() { return emu::g_blk->gx; }
LWM_DEVICE lds_t dyn_lds() { return emu::kLdsBase; }
LWM_DEVICE void block_sync() { emu::block_barrier(); }
LWM_DEVICE void block_sync_lds() { emu::block_barrier(); }
LWM_DEVICE void wave_lds_fence() { emu::wave_sync(); }

LWM_DEVICE f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    emu::Wave& w = emu::g_blk->waves[emu::g_lane->tid >> 6];
    int l = emu::g_lane->tid & 63;
    w.a[l] = a;
    w.b[l] = b;
    emu::wave_sync();
    f32x16 d;
    int col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float s = c[r];
        for (int k = 0; k < 16; ++k) {
            float av = (float)w.a[(k >> 3) * 32 + row][k & 7];
            float bv = (float)w.b[(k >> 3) * 32 + col][k & 7];
            s += av * bv;
        }
        d[r] = s;
    }
    emu::wave_sync();
    return d;
}

// v_mfma_f32_32x32x2_f32: exact f32, k = 0 (lanes 0-31) then k = 1 (lanes 32-63).
LWM_DEVICE f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
    emu::Wave& w = emu::g_blk->waves[emu::g_lane->tid >> 6];
    int l = emu::g_lane->tid & 63;
    w.fa[l] = a;
    w.fb[l] = b;
    emu::wave_sync();
    f32x16 d;
    int col = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float s = fmaf(w.fa[row], w.fb[col], c[r]);
        d[r] = fmaf(w.fa[32 + row], w.fb[32 + col], s);
    }
    emu::wave_sync();
    return d;
}

LWM_DEVICE bf16x4 lds_read_tr16(lds_t a) {
    const char* p = emu::lds_ptr(a, 8, 8);
    emu::Wave& w = emu::g_blk->waves[emu::g_lane->tid >> 6];
    int l = emu::g_lane->tid & 63;
    w.addr[l] = p;
    emu::wave_sync();
    int g = l >> 4, i = l & 15;
    bf16x4 o;
    for (int j = 0; j < 4; ++j) {
        const bf16_t* src = (const bf16_t*)w.addr[16 * g + 4 * j + (i >> 2)];
        o[j] = src[i & 3];
    }
    emu::wave_sync();
    return o;
}

LWM_DEVICE bf16x8 lds_read_b128(lds_t a) { bf16x8 v; memcpy(&v, emu::lds_ptr(a, 16, 16), 16); return v; }
LWM_DEVICE f32x4 lds_read_f32x4(lds_t a) { f32x4 v; memcpy(&v, emu::lds_ptr(a, 16, 16), 16); return v; }
LWM_DEVICE u32x4 lds_read_u32x4(lds_t a) { u32x4 v; memcpy(&v, emu::lds_ptr(a, 16, 16), 16); return v; }
LWM_DEVICE void lds_write_b128(lds_t a, u32x4 v) { memcpy(emu::lds_ptr(a, 16, 16), &v, 16); }
LWM_DEVICE void lds_write_b64(lds_t a, u32x2 v) { memcpy(emu::lds_ptr(a, 8, 8), &v, 8); }
LWM_DEVICE void lds_write_f32x4(lds_t a, f32x4 v) { memcpy(emu::lds_ptr(a, 16, 16), &v, 16); }
LWM_DEVICE void lds_write_i32(lds_t a, int32_t v) { memcpy(emu::lds_ptr(a, 4, 4), &v, 4); }
LWM_DEVICE void lds_write_bf16(lds_t a, bf16_t v) { memcpy(emu::lds_ptr(a, 2, 2), &v, 2); }
LWM_DEVICE void lds_write_f32(lds_t a, float v) { memcpy(emu::lds_ptr(a, 4, 4), &v, 4); }
LWM_DEVICE float lds_read_f32(lds_t a) { float v; memcpy(&v, emu::lds_ptr(a, 4, 4), 4); return v; }
LWM_DEVICE int32_t lds_read_i32(lds_t a) { int32_t v; memcpy(&v, emu::lds_ptr(a, 4, 4), 4); return v; }
LWM_DEVICE void lds_write_f64(lds_t a, double v) { memcpy(emu::lds_ptr(a, 8, 8), &v, 8); }
LWM_DEVICE double lds_read_f64(lds_t a) { double v; memcpy(&v, emu::lds_ptr(a, 8, 8), 8); return v; }
LWM_DEVICE void glds_load_b128(const void* g, lds_t wave_base) {
    int l = emu::g_lane->tid & 63;
    memcpy(emu::lds_ptr(wave_base + 16 * l, 16, 16), g, 16);
}
LWM_DEVICE void glds_load_b32(const void* g, lds_t wave_base) {
    int l = emu::g_lane->tid & 63;
    memcpy(emu::lds_ptr(wave_base + 4 * l, 4, 4), g, 4);
}
template <int N>
LWM_DEVICE void wait_vmem_le() {}
LWM_DEVICE void glds_wait_all() {}
LWM_DEVICE int wave_uniform(int x) { return x; }
LWM_DEVICE void sched_fence() {}
LWM_DEVICE uint32_t opaque(uint32_t x) { return x; }
LWM_DEVICE void prio_hi() {}
LWM_DEVICE void prio_lo() {}

LWM_DEVICE float shfl_xor_f(float x, int m) {
    emu::Wave& w = emu::g_blk->waves[emu::g_lane->tid >> 6];
    int l = emu::g_lane->tid & 63;
    w.f[l] = x;
    emu::wave_sync();
    float r = w.f[l ^ m];
    emu::wave_sync();
    return r;
}
LWM_DEVICE int shfl_xor_i(int x, int m) {
    emu::Wave& w = emu::g_blk->waves[emu::g_lane->tid >> 6];
    int l = emu::g_lane->tid & 63;
    w.i[l] = x;
    emu::wave_sync();
    int r = w.i[l ^ m];
    emu::wave_sync();
    return r;
}
LWM_DEVICE float xhalf(float x) { return shfl_xor_f(x, 32); }
LWM_DEVICE float lane_value(float x, int src_lane) {       // v_readlane_b32: every lane gets lane `src_lane`'s x
    emu::Wave& w = emu::g_blk->waves[emu::g_lane->tid >> 6];
    int l = emu::g_lane->tid & 63;
    w.f[l] = x;
    emu::wave_sync();
    float r = w.f[src_lane & 63];
    emu::wave_sync();
    return r;
}
LWM_DEVICE bool wave_any(bool x) {
    emu::Wave& w = emu::g_blk->waves[emu::g_lane->tid >> 6];
    int l = emu::g_lane->tid & 63;
    w.i[l] = x ? 1 : 0;
    emu::wave_sync();
    int r = 0;
    for (int k = 0; k < 64; ++k) r |= w.i[k];
    emu::wave_sync();
    return r != 0;
}

LWM_DEVICE float fast_exp2(float x) { return exp2f(x); }
LWM_DEVICE float fast_log2(float x) { return log2f(x); }

LWM_DEVICE u32x4 global_load_b128(const void* p) {
    u32x4 v;
    memcpy(&v, p, 16);
    return v;
}
LWM_DEVICE void global_store_b128(void* p, u32x4 v) { memcpy(p, &v, 16); }
LWM_DEVICE void global_store_b64(void* p, u32x2 v) { memcpy(p, &v, 8); }
LWM_DEVICE f32x4 global_load_f32x4(const float* p) { f32x4 v; memcpy(&v, p, 16); return v; }
LWM_DEVICE void global_store_f32x4(float* p, f32x4 v) { memcpy(p, &v, 16); }

LWM_DEVICE int xcc_id() { return emu::g_blk->bx & 7; }
LWM_DEVICE int atomic_add_i32(int32_t* p, int32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
LWM_DEVICE float global_load_f32_at(const float* base, uint32_t voff, uint32_t soff) {
    float v;
    memcpy(&v, (const char*)base + voff + soff, 4);
    return v;
}
LWM_DEVICE void global_store_f32_at(float* base, uint32_t voff, uint32_t soff, float v) {
    memcpy((char*)base + voff + soff, &v, 4);
}
LWM_DEVICE void wave_priority(int) {}
struct ranged_t { const char* base; uint32_t bytes; };      // zeros past the range, as a buffer descriptor does
LWM_DEVICE ranged_t ranged_make(const float* base, uint32_t bytes) { return ranged_t{(const char*)base, bytes}; }
LWM_DEVICE f32x4 ranged_load_f32x4(ranged_t r, uint32_t voff) {
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if ((uint64_t)voff + 16 <= r.bytes) memcpy(&v, r.base + voff, 16);
    return v;
}
LWM_DEVICE float uniform_load_f32(const float* p, int idx) { return p[idx]; }
LWM_DEVICE uint32_t pack_bf16x2(float lo, float hi) {
    union { bf16_t h[2]; uint32_t u; } x;
    x.h[0] = (bf16_t)lo;
    x.h[1] = (bf16_t)hi;
    return x.u;
}

}  // namespace lwm
