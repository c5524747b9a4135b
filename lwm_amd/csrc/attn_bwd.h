// attn_bwd.h -- blockwise attention backward for one (q block, kv block) ring
// step on gfx950.  Requires wave_ops.h + attn_common.h.
//
// Replaces the custom-VJP backward of `ringattention` (call site
// lwm/llama.py:539-569): recompute p from the saved
// LSE, dv += p^T do, dp = do v^T, ds = p*(dp - rowsum(do*o)), dq += ds k,
// dk += ds^T q, all scaled by 1/sqrt(D) where the reference does.
//
// Three kernels, no atomics, deterministic:
//   attn_bwd_delta_kernel : the row statistics of the backward           (HBM bound)
//   attn_bwd_dq4_kernel   : workgroup owns 128 queries, streams K/V      (3 GEMMs; attn_bwd64.h)
//   attn_bwd_dkdv4_kernel : workgroup owns 128 keys, streams Q/dO        (4 GEMMs; attn_bwd64.h)
// The f32 *_acc carries let a ring driver accumulate dq locally and dk/dv in
// buffers that travel with the K/V block.
//
// Row statistics (AttnParams::delta, written by the delta kernel, read by the other two): per (batch, head) two rows
// of Sqp = Sq rounded up to 64 floats,
//   nl2[q] = -lse[q] * log2(e)      (-inf for a row that saw no key and for the padding q >= Sq: p = exp2(s c + nl2) = 0)
//   nd[q]  = -rowsum(dO * O)[q]     (0 in the padding)
// already in the form the kernels consume (one fma per score; -delta as the initial value of the dP accumulator), so
// that they can travel to LDS by DMA with no arithmetic on the way.
#pragma once

namespace lwm {

// ------------------------------------------------------------------ row statistics
constexpr int kDeltaThreads = 256;
LWM_HD int64_t bwd_stat_pad(int64_t Sq) { return (Sq + 63) & ~(int64_t)63; }
// element offset of the nl2 row of (b,h); the nd row follows at + Sqp
LWM_HD int64_t bwd_stat_row(int64_t bh, int64_t Sqp) { return bh * 2 * Sqp; }

LWM_KERNEL(kDeltaThreads) void attn_bwd_delta_kernel(AttnParams p, float* stats) {
    const int tid = thread_idx();
    const int64_t Sqp = bwd_stat_pad(p.Sq);
    const int64_t rows = (int64_t)p.B * p.H * Sqp;
    const int part = tid & 15;
    int64_t row = (int64_t)block_idx_x() * (kDeltaThreads / 16) + (tid >> 4);
    const int64_t row_step = (int64_t)grid_dim_x() * (kDeltaThreads / 16);
    // every lane runs the same number of iterations (shuffles need full waves)
    const int64_t iters = (rows + row_step - 1) / row_step;
    for (int64_t it = 0; it < iters; ++it, row += row_step) {
        float s = 0.0f;
        const int64_t q = row % Sqp, bh = row / Sqp;
        const bool ok = row < rows && q < p.Sq;
        if (ok) {
            const int64_t h = bh % p.H, b = bh / p.H;
            u32x4 ov = global_load_b128(p.out + b * p.o_sb + q * p.o_ss + h * p.o_sh + part * 8);
            u32x4 dv = global_load_b128(p.dout + b * p.do_sb + q * p.do_ss + h * p.do_sh + part * 8);
            bf16x8 o8 = __builtin_bit_cast(bf16x8, ov);
            bf16x8 d8 = __builtin_bit_cast(bf16x8, dv);
            for (int j = 0; j < 8; ++j) s += (float)o8[j] * (float)d8[j];
        }
        s += shfl_xor_f(s, 1);
        s += shfl_xor_f(s, 2);
        s += shfl_xor_f(s, 4);
        s += shfl_xor_f(s, 8);
        if (row < rows && part == 0) {
            float nl2 = -INFINITY;
            if (ok) {
                const float l = p.lse[bh * p.Sq + q];
                nl2 = (l == -INFINITY) ? -INFINITY : -l * kLog2e;
            }
            stats[bwd_stat_row(bh, Sqp) + q] = nl2;
            stats[bwd_stat_row(bh, Sqp) + Sqp + q] = ok ? -s : 0.0f;
        }
    }
}

}  // namespace lwm
