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
    const bf16_t* res;    // bwd: gradient of the residual branch that by-passes the norm (or null): dx += res, one pass
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
                if (p.res) {
                    // the block's residual branch: x feeds the norm AND the add behind it, so its gradient is
                    // bf16(dx_norm) + res, rounded as autograd's separate bf16 add would round it
                    u32x4 rr = global_load_b128(p.res + row * p.C + v * 8);
                    for (int j = 0; j < 4; ++j) {
                        const float d0 = (float)(bf16_t)(r * (dy[k][2 * j] - xh[k][2 * j] * mean));
                        const float d1 = (float)(bf16_t)(r * (dy[k][2 * j + 1] - xh[k][2 * j + 1] * mean));
                        o[j] = pack_bf16x2(d0 + __builtin_bit_cast(float, rr[j] << 16),
                                           d1 + __builtin_bit_cast(float, rr[j] & 0xffff0000u));
                    }
                } else {
                    for (int j = 0; j < 4; ++j)
                        o[j] = pack_bf16x2(r * (dy[k][2 * j] - xh[k][2 * j] * mean),
                                           r * (dy[k][2 * j + 1] - xh[k][2 * j + 1] * mean));
                }
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

// dw[c] = bf16(sum_blocks dw_part[block][c]), a fixed tree: one workgroup = 32 columns x 8 row groups (row group j sums the
// blocks j, j + 8, ... in order -- four independent chains per thread keep loads in flight), then the 8 group sums in
// group order.  (Round 5's kernel walked all blocks of a column in ONE thread: 2048 dependent loads, 0.61 ms per call
// -- 40 ms of the 32-layer LWM-7B step, profiles/r06_model_full.md.)
LWM_KERNEL(256) void rmsnorm_dw_reduce_kernel(const float* part, bf16_t* dw, int nblk, int C) {
    const lds_t lds = dyn_lds();          // 8 x 32 floats
    const int tid = thread_idx();
    const int col = block_idx_x() * 32 + (tid & 31), grp = tid >> 5;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (col < C) {
        int i = grp;
        for (; i + 24 < nblk; i += 32) {
            s0 += part[(int64_t)i * C + col];
            s1 += part[(int64_t)(i + 8) * C + col];
            s2 += part[(int64_t)(i + 16) * C + col];
            s3 += part[(int64_t)(i + 24) * C + col];
        }
        for (; i < nblk; i += 8) s0 += part[(int64_t)i * C + col];
    }
    lds_write_f32(lds + tid * 4, (s0 + s1) + (s2 + s3));
    block_sync();
    if (tid < 32 && col < C) {
        float t = 0.0f;
        for (int g2 = 0; g2 < 8; ++g2) t += lds_read_f32(lds + (g2 * 32 + tid) * 4);
        dw[col] = (bf16_t)t;
    }
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

// The same two kernels on ROWS of a wider buffer: gate and up are the two halves of ONE (rows, 2F) GEMM output (w1 | w3 as
// one library GEMM, lwm_amd/llama_ops.py), and the backward writes d gate | d up into the halves of one (rows, 2F) buffer
// that the fused dgrad / wgrad GEMMs read -- no concatenation pass on either side.  cols % 8 == 0, every ld % 8 == 0.
struct SwigluLdParams {
    const bf16_t* a; const bf16_t* b; const bf16_t* g;
    bf16_t* y; bf16_t* da; bf16_t* db;
    int64_t lda, ldb, ldg, ldy, ldda, lddb, rows;
    int32_t cols;
};

LWM_KERNEL(256) void swiglu_fwd_ld_kernel(SwigluLdParams p) {
    const int cv = p.cols >> 3;
    const int64_t nvec = p.rows * cv;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < nvec; i += (int64_t)grid_dim_x() * 256) {
        const int64_t row = i / cv;
        const int c = (int)(i - row * cv) * 8;
        u32x4 ra = global_load_b128(p.a + row * p.lda + c), rb = global_load_b128(p.b + row * p.ldb + c), o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = __builtin_bit_cast(float, ra[j] << 16), a1 = __builtin_bit_cast(float, ra[j] & 0xffff0000u);
            const float b0 = __builtin_bit_cast(float, rb[j] << 16), b1 = __builtin_bit_cast(float, rb[j] & 0xffff0000u);
            o[j] = pack_bf16x2(a0 * sigmoid_fast(a0) * b0, a1 * sigmoid_fast(a1) * b1);
        }
        global_store_b128(p.y + row * p.ldy + c, o);
    }
}

LWM_KERNEL(256) void swiglu_bwd_ld_kernel(SwigluLdParams p) {
    const int cv = p.cols >> 3;
    const int64_t nvec = p.rows * cv;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < nvec; i += (int64_t)grid_dim_x() * 256) {
        const int64_t row = i / cv;
        const int c = (int)(i - row * cv) * 8;
        u32x4 ra = global_load_b128(p.a + row * p.lda + c), rb = global_load_b128(p.b + row * p.ldb + c),
              rg = global_load_b128(p.g + row * p.ldg + c);
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
        global_store_b128(p.da + row * p.ldda + c, oa);
        global_store_b128(p.db + row * p.lddb + c, ob);
    }
}

// ---------------------------------------------------------------- 2-D transpose (bf16)
// dst[c][r] = src[r][c] over 64 x 64 tiles: 16-byte global loads along the source rows, the tile parked in LDS at a pitch
// of 33 words (column walks hit 32 different banks, two lanes per word), 16-byte global stores along the destination rows.
// What it is for: hipBLASLt runs a GEMM fastest when BOTH operands have the reduction dimension contiguous
// (profiles/r06_model_full.md: 1.36-1.58 PF/s against 0.90-1.05 for the weight-gradient layout x^T g).  The (in, out)
// flax kernels (lwm/llama.py:390-421) are re-laid as (out, in) once per step for the forward GEMMs, and the narrow
// operand of every weight gradient is transposed so that S is contiguous.  HBM-bound: 2 x rows x cols x 2 B.
struct TransposeParams {
    const bf16_t* src; bf16_t* dst;
    int64_t ld_src, ld_dst;
    int32_t tiles_r, tiles_c;     // rows / 64, cols / 64
};

constexpr int kTrPitchWords = 33;
LWM_KERNEL(256) void transpose_bf16_kernel(TransposeParams p) {
    const lds_t lds = dyn_lds();          // 64 x 33 words
    const int tid = thread_idx();
    const int64_t ntiles = (int64_t)p.tiles_r * p.tiles_c;
    for (int64_t t = block_idx_x(); t < ntiles; t += grid_dim_x()) {
        // consecutive workgroups walk down the source ROWS of one column band: their stores fill whole destination rows
        const int tc = (int)(t / p.tiles_r), tr = (int)(t - (int64_t)tc * p.tiles_r);
        {
            const int r = tid >> 2, seg = tid & 3;
            const bf16_t* s = p.src + ((int64_t)tr * 64 + r) * p.ld_src + (int64_t)tc * 64 + seg * 16;
            u32x4 v0 = global_load_b128(s), v1 = global_load_b128(s + 8);
            const lds_t w = lds + (r * kTrPitchWords + seg * 8) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lds_write_i32(w + j * 4, (int32_t)v0[j]);
                lds_write_i32(w + (4 + j) * 4, (int32_t)v1[j]);
            }
        }
        block_sync();
        {
            const int c = tid >> 2, rseg = tid & 3;
            const int sh = (c & 1) * 16;
            u32x4 o0, o1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t a0 = (uint32_t)lds_read_i32(lds + ((rseg * 16 + 2 * j) * kTrPitchWords + (c >> 1)) * 4);
                const uint32_t a1 = (uint32_t)lds_read_i32(lds + ((rseg * 16 + 2 * j + 1) * kTrPitchWords + (c >> 1)) * 4);
                const uint32_t b0 = (uint32_t)lds_read_i32(lds + ((rseg * 16 + 8 + 2 * j) * kTrPitchWords + (c >> 1)) * 4);
                const uint32_t b1 = (uint32_t)lds_read_i32(lds + ((rseg * 16 + 8 + 2 * j + 1) * kTrPitchWords + (c >> 1)) * 4);
                o0[j] = ((a0 >> sh) & 0xffffu) | (((a1 >> sh) & 0xffffu) << 16);
                o1[j] = ((b0 >> sh) & 0xffffu) | (((b1 >> sh) & 0xffffu) << 16);
            }
            bf16_t* d = p.dst + ((int64_t)tc * 64 + c) * p.ld_dst + (int64_t)tr * 64 + rseg * 16;
            global_store_b128(d, o0);
            global_store_b128(d + 8, o1);
        }
        block_sync();
    }
}

}  // namespace lwm
