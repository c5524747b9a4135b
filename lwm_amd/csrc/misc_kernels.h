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

}  // namespace lwm
