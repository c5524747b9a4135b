// llama_elem.h -- the HBM-bound steps either side of the attention op in
// FlaxLLaMAAttention / FlaxLLaMABlock: rotary position
// embedding and RMSNorm, forward and backward.  Requires wave_ops.h.
//
//   rope_kernel     apply_rotary_emb (lwm/llama.py:353-375): interleaved pairs
//                   (x[2i], x[2i+1]) as complex numbers times cis(angle[pos][i]), f32
//                   math, cast to the model dtype.  The (cos, sin) table is precomputed on
//                   the host exactly as precompute_freqs_cis does (lwm/llama.py:344-350):
//                   no trigonometry on the device.  conj = 1 multiplies by the conjugate
//                   = the backward pass (the rotation is orthogonal).
//   rmsnorm_*       RMSNorm (lwm/llama.py:320-341): f32 upcast, x * rsqrt(mean(x^2) +
//                   eps) cast to dtype, times weight (dtype).
// Roofline: HBM.  Algorithmic bytes: RoPE 2*n*2 (read + write bf16) + table;
// RMSNorm fwd 2*n*2, bwd 3*n*2 + the dW partials.
#pragma once

namespace lwm {

// ---------------------------------------------------------------- RoPE
// x, y: [B,S,H,D] bf16 (D contiguous, strided), table: [max_pos][D/2][2] f32 (cos, sin),
// pos: [B,S] int32.  One thread = 8 bf16 = 4 complex pairs.
struct RopeParams {
    const bf16_t* x;
    bf16_t* y;
    const float* table;
    const int32_t* pos;
    int64_t x_sb, x_ss, x_sh, y_sb, y_ss, y_sh;
    int32_t B, S, H, D, max_pos, conj;
};

LWM_KERNEL(256) void rope_kernel(RopeParams p) {
    const int vec = p.D >> 3;                         // threads per (b,s,h) row
    const int64_t total = (int64_t)p.B * p.S * p.H * vec;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < total;
         i += (int64_t)grid_dim_x() * 256) {
        const int c = (int)(i % vec);
        int64_t r = i / vec;
        const int h = (int)(r % p.H);
        r /= p.H;
        const int s = (int)(r % p.S);
        const int b = (int)(r / p.S);
        int ps = p.pos[(int64_t)b * p.S + s];
        ps = ps < 0 ? 0 : (ps >= p.max_pos ? p.max_pos - 1 : ps);
        const float* t = p.table + ((int64_t)ps * (p.D >> 1) + c * 4) * 2;
        f32x4 t0 = global_load_f32x4(t), t1 = global_load_f32x4(t + 4);   // (c,s,c,s) x 2
        u32x4 raw = global_load_b128(p.x + (int64_t)b * p.x_sb + (int64_t)s * p.x_ss + (int64_t)h * p.x_sh + c * 8);
        u32x4 o;
        for (int j = 0; j < 4; ++j) {
            const float x0 = __builtin_bit_cast(float, raw[j] << 16);
            const float x1 = __builtin_bit_cast(float, raw[j] & 0xffff0000u);
            const float cs = j < 2 ? t0[2 * j] : t1[2 * j - 4];
            float sn = j < 2 ? t0[2 * j + 1] : t1[2 * j - 3];
            sn = p.conj ? -sn : sn;
            // complex multiply as jnp does: re = x0*c - x1*s, im = x0*s + x1*c
            o[j] = pack_bf16x2(x0 * cs - x1 * sn, x0 * sn + x1 * cs);
        }
        global_store_b128(p.y + (int64_t)b * p.y_sb + (int64_t)s * p.y_ss + (int64_t)h * p.y_sh + c * 8, o);
    }
}

// ---------------------------------------------------------------- RMSNorm
// rows of C bf16 (C % 8 == 0, C <= 8192), one workgroup (256 threads) per row.
struct RmsParams {
    const bf16_t* x;
    const bf16_t* w;      // [C]
    const bf16_t* g;      // upstream gradient (bwd)
    bf16_t* y;            // fwd output / dx
    float* rstd;          // [rows]: saved by fwd (may be null), read by bwd
    float* dw_part;       // [gridDim][C] f32 partial weight gradients (bwd)
    int64_t rows;
    int32_t C;
    float eps;
};

template <int NT>
LWM_DEVICE float block_sum(float v, lds_t scratch, int tid) {
    v += shfl_xor_f(v, 1);
    v += shfl_xor_f(v, 2);
    v += shfl_xor_f(v, 4);
    v += shfl_xor_f(v, 8);
    v += shfl_xor_f(v, 16);
    v += shfl_xor_f(v, 32);
    if ((tid & 63) == 0) lds_write_f32(scratch + (tid >> 6) * 4, v);
    block_sync();
    float t = 0.0f;
    for (int w = 0; w < NT / 64; ++w) t += lds_read_f32(scratch + w * 4);
    block_sync();
    return t;
}

LWM_DEVICE float block_sum_256(float v, lds_t scratch, int tid) {
    v += shfl_xor_f(v, 1);
    v += shfl_xor_f(v, 2);
    v += shfl_xor_f(v, 4);
    v += shfl_xor_f(v, 8);
    v += shfl_xor_f(v, 16);
    v += shfl_xor_f(v, 32);
    if ((tid & 63) == 0) lds_write_f32(scratch + (tid >> 6) * 4, v);
    block_sync();
    const float t = lds_read_f32(scratch) + lds_read_f32(scratch + 4) + lds_read_f32(scratch + 8) +
                    lds_read_f32(scratch + 12);
    block_sync();
    return t;
}

LWM_KERNEL(256) void rmsnorm_fwd_kernel(RmsParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int nv = p.C >> 3;   // 16-byte vectors per row
    for (int64_t row = block_idx_x(); row < p.rows; row += grid_dim_x()) {
        const bf16_t* xr = p.x + row * p.C;
        float xs[4][8];
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = tid + 256 * k;
            if (v < nv) {
                u32x4 raw = global_load_b128(xr + v * 8);
                for (int j = 0; j < 4; ++j) {
                    xs[k][2 * j] = __builtin_bit_cast(float, raw[j] << 16);
                    xs[k][2 * j + 1] = __builtin_bit_cast(float, raw[j] & 0xffff0000u);
                }
                for (int j = 0; j < 8; ++j) ss = fmaf(xs[k][j], xs[k][j], ss);
            }
        }
        const float tot = block_sum_256(ss, lds, tid);
        const float r = 1.0f / sqrtf(tot / (float)p.C + p.eps);
        if (tid == 0 && p.rstd) p.rstd[row] = r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = tid + 256 * k;
            if (v < nv) {
                u32x4 wr = global_load_b128(p.w + v * 8);
                u32x4 o;
                for (int j = 0; j < 4; ++j) {
                    // output = bf16(bf16(x*r) * w): the reference casts the normalised value
                    // to dtype BEFORE the weight product (lwm/llama.py:339-341)
                    const float y0 = (float)(bf16_t)(xs[k][2 * j] * r), y1 = (float)(bf16_t)(xs[k][2 * j + 1] * r);
                    const float w0 = __builtin_bit_cast(float, wr[j] << 16);
                    const float w1 = __builtin_bit_cast(float, wr[j] & 0xffff0000u);
                    o[j] = pack_bf16x2(y0 * w0, y1 * w1);
                }
                global_store_b128(p.y + row * p.C + v * 8, o);
            }
        }
    }
}

// dx = r * (dy - xhat * mean(dy * xhat)),  dy = g * w,  xhat = x * r;
// dW[c] = sum_rows g * xhat  (this workgroup's rows -> dw_part[block][c]).
LWM_KERNEL(256) void rmsnorm_bwd_kernel(RmsParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int nv = p.C >> 3;
    float dw[4][8];
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 8; ++j) dw[k][j] = 0.0f;
    for (int64_t row = block_idx_x(); row < p.rows; row += grid_dim_x()) {
        const float r = p.rstd[row];
        float xh[4][8], dy[4][8];
        float dot = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = tid + 256 * k;
            if (v < nv) {
                u32x4 xr = global_load_b128(p.x + row * p.C + v * 8);
                u32x4 gr = global_load_b128(p.g + row * p.C + v * 8);
                u32x4 wr = global_load_b128(p.w + v * 8);
                for (int j = 0; j < 4; ++j) {
                    const float x0 = __builtin_bit_cast(float, xr[j] << 16), x1 = __builtin_bit_cast(float, xr[j] & 0xffff0000u);
                    const float g0 = __builtin_bit_cast(float, gr[j] << 16), g1 = __builtin_bit_cast(float, gr[j] & 0xffff0000u);
                    const float w0 = __builtin_bit_cast(float, wr[j] << 16), w1 = __builtin_bit_cast(float, wr[j] & 0xffff0000u);
                    xh[k][2 * j] = x0 * r;
                    xh[k][2 * j + 1] = x1 * r;
                    dy[k][2 * j] = g0 * w0;
                    dy[k][2 * j + 1] = g1 * w1;
                    dw[k][2 * j] = fmaf(g0, xh[k][2 * j], dw[k][2 * j]);
                    dw[k][2 * j + 1] = fmaf(g1, xh[k][2 * j + 1], dw[k][2 * j + 1]);
                }
                for (int j = 0; j < 8; ++j) dot = fmaf(dy[k][j], xh[k][j], dot);
            }
        }
        const float mean = block_sum_256(dot, lds, tid) / (float)p.C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = tid + 256 * k;
            if (v < nv) {
                u32x4 o;
                for (int j = 0; j < 4; ++j)
                    o[j] = pack_bf16x2(r * (dy[k][2 * j] - xh[k][2 * j] * mean),
                                       r * (dy[k][2 * j + 1] - xh[k][2 * j + 1] * mean));
                global_store_b128(p.y + row * p.C + v * 8, o);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int v = tid + 256 * k;
        if (v < nv) {
            float* o = p.dw_part + (int64_t)block_idx_x() * p.C + v * 8;
            f32x4 a = {dw[k][0], dw[k][1], dw[k][2], dw[k][3]}, b2 = {dw[k][4], dw[k][5], dw[k][6], dw[k][7]};
            global_store_f32x4(o, a);
            global_store_f32x4(o + 4, b2);
        }
    }
}

// dw[c] = bf16(sum_blocks dw_part[block][c]) in block order.
LWM_KERNEL(256) void rmsnorm_dw_reduce_kernel(const float* part, bf16_t* dw, int nblk, int C) {
    const int c = block_idx_x() * 256 + thread_idx();
    if (c >= C) return;
    float s = 0.0f;
    for (int i = 0; i < nblk; ++i) s += part[(int64_t)i * C + c];
    dw[c] = (bf16_t)s;
}

// ---------------------------------------------------------------- softmax cross-entropy
// tux.cross_entropy_loss_and_accuracy as called at lwm/train.py:177-181, :192-201: f32
// log-softmax over the vocabulary, the target's log-probability, argmax == target, and (fused,
// same pass) the logits gradient (softmax - onehot) * w[row].  One workgroup per row; the row
// (V <= 32768 bf16) is read from HBM ONCE into registers.  HBM-bound: V*2 B read (+ V*2 B
// gradient write) per row.
struct CeParams {
    const bf16_t* logits;   // [rows, V]
    const int32_t* target;  // [rows]
    const float* weight;    // [rows] gradient weight (valid / normaliser) or null (= 1)
    float* nll;             // [rows]  -log p(target)
    int32_t* correct;       // [rows]  argmax == target (first maximum), or null
    bf16_t* dlogits;        // [rows, V] or null
    int64_t rows;
    int32_t V;
};

constexpr int kCeThreads = 512;   // 8 vectors of 8 logits per thread: V <= 32768
LWM_KERNEL(kCeThreads) void softmax_ce_kernel(CeParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int nv = p.V >> 3;
    for (int64_t row = block_idx_x(); row < p.rows; row += grid_dim_x()) {
        const bf16_t* lr = p.logits + row * p.V;
        float x[8][8];
        float mx = -INFINITY;
        int amax = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = tid + kCeThreads * k;
            if (v < nv) {
                u32x4 raw = global_load_b128(lr + v * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[k][2 * j] = __builtin_bit_cast(float, raw[j] << 16);
                    x[k][2 * j + 1] = __builtin_bit_cast(float, raw[j] & 0xffff0000u);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (x[k][j] > mx) {          // strict: keeps the first maximum of this thread
                        mx = x[k][j];
                        amax = v * 8 + j;
                    }
            }
        }
        // block argmax: larger value wins, ties -> smaller index
        for (int m = 1; m < 64; m <<= 1) {
            const float om = shfl_xor_f(mx, m);
            const int oi = shfl_xor_i(amax, m);
            if (om > mx || (om == mx && oi < amax)) {
                mx = om;
                amax = oi;
            }
        }
        if ((tid & 63) == 0) {
            lds_write_f32(lds + (tid >> 6) * 8, mx);
            lds_write_i32(lds + (tid >> 6) * 8 + 4, amax);
        }
        block_sync();
        for (int w = 0; w < kCeThreads / 64; ++w) {
            const float om = lds_read_f32(lds + w * 8);
            const int oi = lds_read_i32(lds + w * 8 + 4);
            if (om > mx || (om == mx && oi < amax)) {
                mx = om;
                amax = oi;
            }
        }
        block_sync();
        float se = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (tid + kCeThreads * k < nv)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    x[k][j] = fast_exp2((x[k][j] - mx) * 1.4426950408889634f);
                    se += x[k][j];
                }
        const float tot = block_sum<kCeThreads>(se, lds + 64, tid);
        const int tg = p.target[row];
        const float wgt = p.weight ? p.weight[row] : 1.0f;
        if (tid == 0) {
            const float lt = (tg >= 0 && tg < p.V) ? (float)lr[tg] : 0.0f;
            p.nll[row] = (mx + logf(tot)) - lt;
            if (p.correct) p.correct[row] = (amax == tg) ? 1 : 0;
        }
        if (p.dlogits) {
            const float s = wgt / tot;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int v = tid + kCeThreads * k;
                if (v < nv) {
                    u32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i0 = v * 8 + 2 * j;
                        const float g0 = x[k][2 * j] * s - (i0 == tg ? wgt : 0.0f);
                        const float g1 = x[k][2 * j + 1] * s - (i0 + 1 == tg ? wgt : 0.0f);
                        o[j] = pack_bf16x2(g0, g1);
                    }
                    global_store_b128(p.dlogits + row * p.V + v * 8, o);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- SwiGLU gate
// FlaxLLaMAMLP (lwm/llama.py:659): w2(silu(w1 x) * w3 x) -- the two GEMMs are library
// GEMMs; this is the elementwise gate between them, forward and backward, bf16, 8 elements
// per thread.  HBM-bound: fwd 3*n*2 B, bwd 5*n*2 B.
LWM_DEVICE float sigmoid_fast(float a) { return 1.0f / (1.0f + fast_exp2(-a * 1.4426950408889634f)); }

LWM_KERNEL(256) void swiglu_fwd_kernel(const bf16_t* a, const bf16_t* b, bf16_t* y, int64_t n) {
    const int64_t nvec = n >> 3;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < nvec; i += (int64_t)grid_dim_x() * 256) {
        u32x4 ra = global_load_b128(a + i * 8), rb = global_load_b128(b + i * 8), o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = __builtin_bit_cast(float, ra[j] << 16), a1 = __builtin_bit_cast(float, ra[j] & 0xffff0000u);
            const float b0 = __builtin_bit_cast(float, rb[j] << 16), b1 = __builtin_bit_cast(float, rb[j] & 0xffff0000u);
            o[j] = pack_bf16x2(a0 * sigmoid_fast(a0) * b0, a1 * sigmoid_fast(a1) * b1);
        }
        global_store_b128(y + i * 8, o);
    }
}

LWM_KERNEL(256) void swiglu_bwd_kernel(const bf16_t* a, const bf16_t* b, const bf16_t* g, bf16_t* da,
                                       bf16_t* db, int64_t n) {
    const int64_t nvec = n >> 3;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < nvec; i += (int64_t)grid_dim_x() * 256) {
        u32x4 ra = global_load_b128(a + i * 8), rb = global_load_b128(b + i * 8), rg = global_load_b128(g + i * 8);
        u32x4 oa, ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float av[2] = {__builtin_bit_cast(float, ra[j] << 16), __builtin_bit_cast(float, ra[j] & 0xffff0000u)};
            float bv[2] = {__builtin_bit_cast(float, rb[j] << 16), __builtin_bit_cast(float, rb[j] & 0xffff0000u)};
            float gv[2] = {__builtin_bit_cast(float, rg[j] << 16), __builtin_bit_cast(float, rg[j] & 0xffff0000u)};
            float dav[2], dbv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float sg = sigmoid_fast(av[e]);
                dbv[e] = gv[e] * av[e] * sg;
                dav[e] = gv[e] * bv[e] * sg * (1.0f + av[e] * (1.0f - sg));
            }
            oa[j] = pack_bf16x2(dav[0], dav[1]);
            ob[j] = pack_bf16x2(dbv[0], dbv[1]);
        }
        global_store_b128(da + i * 8, oa);
        global_store_b128(db + i * 8, ob);
    }
}

}  // namespace lwm
