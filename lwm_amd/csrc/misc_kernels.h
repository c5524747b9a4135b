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
        u32x4 a = global_load_b128(src + i * 8);
        u32x4 b = global_load_b128(src + i * 8 + 4);
        u32x4 o = {pack_bf16x2(__builtin_bit_cast(float, a[0]), __builtin_bit_cast(float, a[1])),
                   pack_bf16x2(__builtin_bit_cast(float, a[2]), __builtin_bit_cast(float, a[3])),
                   pack_bf16x2(__builtin_bit_cast(float, b[0]), __builtin_bit_cast(float, b[1])),
                   pack_bf16x2(__builtin_bit_cast(float, b[2]), __builtin_bit_cast(float, b[3]))};
        global_store_b128(dst + i * 8, o);
    }
    // tail (n not a multiple of 8)
    if (block_idx_x() == 0 && thread_idx() == 0)
        for (int64_t j = nvec << 3; j < n; ++j) dst[j] = (bf16_t)src[j];
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

LWM_KERNEL(256) void attn_combine_kernel(CombineParams p) {
    const int dq = p.D >> 2;
    const int64_t total = (int64_t)p.B * p.Sq * p.H * dq;
    const int64_t part_o = (int64_t)p.B * p.Sq * p.H * p.D, part_l = (int64_t)p.B * p.H * p.Sq;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < total;
         i += (int64_t)grid_dim_x() * 256) {
        const int c = (int)(i % dq) * 4;
        int64_t r = i / dq;
        const int h = (int)(r % p.H);
        r /= p.H;
        const int q = (int)(r % p.Sq);
        const int b = (int)(r / p.Sq);
        const int64_t li = ((int64_t)b * p.H + h) * p.Sq + q;
        const int64_t oi = (((int64_t)b * p.Sq + q) * p.H + h) * p.D + c;
        float mx = -INFINITY;
        for (int s = 0; s < p.P; ++s) mx = fmaxf(mx, p.lse_parts[s * part_l + li]);
        float den = 0.0f;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (mx != -INFINITY) {
            for (int s = 0; s < p.P; ++s) {
                const float l = p.lse_parts[s * part_l + li];
                if (l == -INFINITY) continue;
                const float w = expf(l - mx);
                den += w;
                f32x4 o = global_load_f32x4(p.o_parts + s * part_o + oi);
                for (int j = 0; j < 4; ++j) acc[j] += w * o[j];
            }
            const float inv = 1.0f / den;
            for (int j = 0; j < 4; ++j) acc[j] *= inv;
        }
        if (p.out) {
            u32x2 pk = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])};
            global_store_b64(p.out + (int64_t)b * p.o_sb + (int64_t)q * p.o_ss + (int64_t)h * p.o_sh + c, pk);
        }
        if (p.out_f32) global_store_f32x4(p.out_f32 + oi, acc);
        if (p.lse && c == 0) p.lse[li] = mx == -INFINITY ? -INFINITY : mx + logf(den);
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

}  // namespace lwm
