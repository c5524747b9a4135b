// misc_kernels.h -- small HBM-bound helpers of the ring driver.
// Requires wave_ops.h.
#pragma once

namespace lwm {

constexpr int kCastThreads = 256;

// dst[i] = bf16(src[i]); 8 elements (32 B in, 16 B out) per thread per step.
LWM_KERNEL(kCastThreads) void cast_f32_to_bf16_kernel(const float* src, bf16_t* dst, int64_t n) {
    const int64_t nvec = n >> 3;
    int64_t i = (int64_t)block_idx_x() * kCastThreads + thread_idx();
    const int64_t step = (int64_t)grid_dim_x() * kCastThreads;
    for (; i < nvec; i += step) {
        // f32x4 loads: __builtin_bit_cast(float, v[k]) on an element of a u32 ext-vector
        // reads element 0 for every k (host clang and hipcc alike) -- never bit-cast a
        // vector ELEMENT lvalue; cast whole vectors or rvalue expressions.
        f32x4 a = global_load_f32x4(src + i * 8);
        f32x4 b = global_load_f32x4(src + i * 8 + 4);
        u32x4 o = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]),
                   pack_bf16x2(b[2], b[3])};
        global_store_b128(dst + i * 8, o);
    }
    // tail (n not a multiple of 8)
    if (block_idx_x() == 0 && thread_idx() == 0)
        for (int64_t j = nvec << 3; j < n; ++j) dst[j] = (bf16_t)src[j];
}

// dst[i] = bf16(((src0[i] + src1[i]) + src2[i]) + ...): the owner-side reduction of the
// dK/dV partials that the other ranks return under the mesh schedule (lwm_amd/ring.py).
// The order is the argument order, so a rank's result does not depend on arrival times.
constexpr int kSumMaxSrc = 16;
struct SumSrcs { const float* p[kSumMaxSrc]; };
LWM_KERNEL(kCastThreads) void sum_f32_to_bf16_kernel(SumSrcs srcs, int n_src, bf16_t* dst, int64_t n) {
    const int64_t nvec = n >> 3;
    int64_t i = (int64_t)block_idx_x() * kCastThreads + thread_idx();
    const int64_t step = (int64_t)grid_dim_x() * kCastThreads;
    for (; i < nvec; i += step) {
        f32x4 a = global_load_f32x4(srcs.p[0] + i * 8);
        f32x4 b = global_load_f32x4(srcs.p[0] + i * 8 + 4);
        for (int s = 1; s < n_src; ++s) {
            a += global_load_f32x4(srcs.p[s] + i * 8);
            b += global_load_f32x4(srcs.p[s] + i * 8 + 4);
        }
        u32x4 o = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]),
                   pack_bf16x2(b[2], b[3])};
        global_store_b128(dst + i * 8, o);
    }
    if (block_idx_x() == 0 && thread_idx() == 0)
        for (int64_t j = nvec << 3; j < n; ++j) {
            float t = srcs.p[0][j];
            for (int s = 1; s < n_src; ++s) t += srcs.p[s][j];
            dst[j] = (bf16_t)t;
        }
}

// (min, max) segment id per block of 32 rows (include/lwm_hip.h, lwm_attn_segment_blocks).
LWM_KERNEL(256) void seg_blocks_kernel(const int32_t* seg, const uint8_t* valid, int32_t* out, int B, int S) {
    const int nblk = (S + 31) >> 5;
    const int64_t i = (int64_t)block_idx_x() * 256 + thread_idx();
    if (i >= (int64_t)B * nblk) return;
    const int b = (int)(i / nblk), blk = (int)(i % nblk);
    int mn = 0x7fffffff, mx = (int)0x80000000;
    for (int r = 0; r < 32; ++r) {
        const int row = blk * 32 + r;
        if (row < S && (!valid || valid[(int64_t)b * S + row] != 0)) {
            const int v = seg[(int64_t)b * S + row];
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
    }
    out[2 * i] = mn;
    out[2 * i + 1] = mx;
}

// Merge P normalised partial attention results (include/lwm_hip.h, lwm_attn_combine).
// One thread per (b, q, h, 4 consecutive d); HBM-bound.
struct CombineParams {
    const float* o_parts;    // [P][B,Sq,H,D]
    const float* lse_parts;  // [P][B,H,Sq]
    bf16_t* out;             // strided bf16 or null
    int64_t o_sb, o_ss, o_sh;
    float* out_f32;          // dense [B,Sq,H,D] or null
    float* lse;              // [B,H,Sq] or null
    int32_t P, B, Sq, H, D;
};

// Workgroup = one (b, q, h) row: thread t owns d-quad t & 31 and partials
// t >> 5, t >> 5 + kCombineSubs, ... (kCombineSubs subsets walked concurrently, each an online-softmax merge
// in (m, l, acc) form), then the subset states are merged through LDS.  A thread per
// output element with a serial loop over P was latency-bound for decode (P ~ 512 pieces); with 8 subsets the
// 64 dependent steps still cost 26 us beside a 356 us decode launch, with 32 they cost a quarter.  D = 128.
constexpr int kCombineSubs = 32;
constexpr int kCombineThreads = 32 * kCombineSubs;
LWM_KERNEL(kCombineThreads) void attn_combine_kernel(CombineParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int dq = tid & 31, sub = tid >> 5;
    const int64_t part_o = (int64_t)p.B * p.Sq * p.H * p.D, part_l = (int64_t)p.B * p.H * p.Sq;
    for (int64_t row = block_idx_x(); row < (int64_t)p.B * p.Sq * p.H; row += grid_dim_x()) {
        const int h = (int)(row % p.H);
        const int q = (int)((row / p.H) % p.Sq);
        const int b = (int)(row / ((int64_t)p.H * p.Sq));
        const int64_t li = ((int64_t)b * p.H + h) * p.Sq + q;
        const int64_t oi = (((int64_t)b * p.Sq + q) * p.H + h) * p.D + dq * 4;
        float m = -INFINITY, l = 0.0f;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        // four partials in flight per thread (the pieces were written by other XCDs a moment ago: every load is a
        // trip to the fabric, and one at a time they cost ~1 us each); loads are unconditional, pieces past P
        // re-read the last one and are skipped like empty pieces (lse = -inf)
        for (int s0 = sub; s0 < p.P; s0 += 4 * kCombineSubs) {
            float ls[4];
            f32x4 o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u * kCombineSubs;
                const int sc = s < p.P ? s : p.P - 1;
                ls[u] = p.lse_parts[sc * part_l + li];
                o[u] = global_load_f32x4(p.o_parts + sc * part_o + oi);
                if (s >= p.P) ls[u] = -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ls[u] == -INFINITY) continue;
                const float mn = fmaxf(m, ls[u]);
                const float wa = expf(m - mn), wb = expf(ls[u] - mn);   // m = -inf: wa = 0
                l = l * wa + wb;
                for (int j = 0; j < 4; ++j) acc[j] = acc[j] * wa + o[u][j] * wb;
                m = mn;
            }
        }
        // merge the subset states: [sub][dq] -> (m, l, acc[4]) = 24 B
        const lds_t slot = lds + (uint32_t)(sub * 32 + dq) * 32;
        lds_write_f32(slot, m);
        lds_write_f32(slot + 4, l);
        lds_write_f32x4(slot + 16, acc);
        block_sync();
        if (sub == 0) {
            float mx = -INFINITY;
            for (int s = 0; s < kCombineSubs; ++s) mx = fmaxf(mx, lds_read_f32(lds + (uint32_t)(s * 32 + dq) * 32));
            float den = 0.0f;
            f32x4 r = {0.0f, 0.0f, 0.0f, 0.0f};
            if (mx != -INFINITY) {
                for (int s = 0; s < kCombineSubs; ++s) {
                    const lds_t sl = lds + (uint32_t)(s * 32 + dq) * 32;
                    const float ms = lds_read_f32(sl);
                    if (ms == -INFINITY) continue;
                    const float w = expf(ms - mx);
                    den += w * lds_read_f32(sl + 4);
                    f32x4 a = lds_read_f32x4(sl + 16);
                    for (int j = 0; j < 4; ++j) r[j] += w * a[j];
                }
                const float inv = 1.0f / den;
                for (int j = 0; j < 4; ++j) r[j] *= inv;
            }
            if (p.out) {
                u32x2 pk = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
                global_store_b64(p.out + (int64_t)b * p.o_sb + (int64_t)q * p.o_ss + (int64_t)h * p.o_sh + dq * 4, pk);
            }
            if (p.out_f32) global_store_f32x4(p.out_f32 + oi, r);
            if (p.lse && dq == 0) p.lse[li] = mx == -INFINITY ? -INFINITY : mx + logf(den);
        }
        block_sync();
    }
}

// cache[b, dst_row0 + i, :] = src[b, src_row0 + i, :]; rows of row_elems bf16, 16 B per thread.
LWM_KERNEL(256) void kv_cache_write_kernel(bf16_t* cache, const bf16_t* src, int B, int64_t cache_sb,
                                           int64_t src_sb, int64_t dst_row0, int64_t src_row0,
                                           int64_t nrows, int row_elems) {
    const int vec = row_elems >> 3;
    const int64_t total = (int64_t)B * nrows * vec;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < total;
         i += (int64_t)grid_dim_x() * 256) {
        const int c = (int)(i % vec) * 8;
        const int64_t r = (i / vec) % nrows;
        const int64_t b = i / ((int64_t)vec * nrows);
        u32x4 v = global_load_b128(src + b * src_sb + (src_row0 + r) * row_elems + c);
        global_store_b128(cache + b * cache_sb + (dst_row0 + r) * row_elems + c, v);
    }
}

// The same copy with the destination row taken from DEVICE memory: dst row = *row0_dev + row_offset + i,
// rows outside [0, cache_rows) are skipped ("only the owning shard writes", lwm/llama.py:454-467).
// Nothing about the step depends on a host value, so a decode step can be captured in a hipGraph and
// replayed while the index advances on the device.
LWM_KERNEL(256) void kv_cache_write_at_kernel(bf16_t* cache, const bf16_t* src, int B, int64_t cache_sb,
                                              int64_t src_sb, const int32_t* row0_dev, int64_t row_offset,
                                              int64_t cache_rows, int64_t src_row0, int64_t nrows,
                                              int row_elems) {
    const int vec = row_elems >> 3;
    const int64_t total = (int64_t)B * nrows * vec;
    const int64_t dst_row0 = (int64_t)row0_dev[0] + row_offset;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < total;
         i += (int64_t)grid_dim_x() * 256) {
        const int c = (int)(i % vec) * 8;
        const int64_t r = (i / vec) % nrows;
        const int64_t b = i / ((int64_t)vec * nrows);
        const int64_t dr = dst_row0 + r;
        if (dr < 0 || dr >= cache_rows) continue;
        u32x4 v = global_load_b128(src + b * src_sb + (src_row0 + r) * row_elems + c);
        global_store_b128(cache + b * cache_sb + dr * row_elems + c, v);
    }
}

}  // namespace lwm
