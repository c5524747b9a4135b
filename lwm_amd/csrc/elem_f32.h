// elem_f32.h -- the HBM-bound steps either side of the attention op with FLOAT32 tensors: the `--dtype=fp32` flavour
// (the reference's default, lwm/train.py:36; BASELINE configs[0]) of llama_elem.h's kernels and of the owner-side sum of
// misc_kernels.h.  Requires wave_ops.h + llama_elem.h (block_sum_256, block_sum, sigmoid_fast, kCeThreads).
//
//   rope_f32_kernel            apply_rotary_emb (lwm/llama.py:353-375), table as precompute_freqs_cis (:344-350)
//   rmsnorm_{fwd,bwd}_f32      RMSNorm (lwm/llama.py:320-341); at dtype = f32 the reference's two casts are identities
//   swiglu_{fwd,bwd}_f32       the gate of FlaxLLaMAMLP (lwm/llama.py:659)
//   softmax_ce_f32_kernel      tux.cross_entropy_loss_and_accuracy as called at lwm/train.py:177-181 (+ fused gradient)
//   sum_f32_kernel             dst = ((src0 + src1) + src2) + ... : a rank's dK / dV from the partials its peers return
// Same arithmetic as the bf16 kernels, statement for statement, minus the roundings to bf16; one thread = 4 floats = 16 B.
// Roofline: HBM; algorithmic bytes = the bf16 kernels' element counts x 4 B.
#pragma once

namespace lwm {

struct RopeF32Params {
    const float* x;
    float* y;
    const float* table;
    const int32_t* pos;
    int64_t x_sb, x_ss, x_sh, y_sb, y_ss, y_sh;
    int32_t B, S, H, D, max_pos, conj;
};

LWM_KERNEL(256) void rope_f32_kernel(RopeF32Params p) {
    const int vec = p.D >> 2;                         // threads per (b,s,h) row: 4 floats = 2 complex pairs each
    const int64_t total = (int64_t)p.B * p.S * p.H * vec;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < total; i += (int64_t)grid_dim_x() * 256) {
        const int c = (int)(i % vec);
        int64_t r = i / vec;
        const int h = (int)(r % p.H);
        r /= p.H;
        const int s = (int)(r % p.S);
        const int b = (int)(r / p.S);
        int ps = p.pos[(int64_t)b * p.S + s];
        ps = ps < 0 ? 0 : (ps >= p.max_pos ? p.max_pos - 1 : ps);
        const f32x4 t = global_load_f32x4(p.table + ((int64_t)ps * (p.D >> 1) + c * 2) * 2);      // (cos, sin) x 2
        const f32x4 x = global_load_f32x4(p.x + (int64_t)b * p.x_sb + (int64_t)s * p.x_ss + (int64_t)h * p.x_sh + c * 4);
        const float s0 = p.conj ? -t[1] : t[1], s1 = p.conj ? -t[3] : t[3];
        // complex multiply as jnp does: re = x0*c - x1*s, im = x0*s + x1*c
        const f32x4 o = {x[0] * t[0] - x[1] * s0, x[0] * s0 + x[1] * t[0], x[2] * t[2] - x[3] * s1, x[2] * s1 + x[3] * t[2]};
        global_store_f32x4(p.y + (int64_t)b * p.y_sb + (int64_t)s * p.y_ss + (int64_t)h * p.y_sh + c * 4, o);
    }
}

// rows of C floats (C % 4 == 0, C <= 8192: 8 vectors per thread), one workgroup (256 threads) per row
struct RmsF32Params {
    const float* x;
    const float* w;
    const float* g;       // upstream gradient (bwd)
    float* y;             // fwd output / dx
    float* rstd;          // [rows]
    float* dw_part;       // [gridDim][C] partial weight gradients (bwd)
    int64_t rows;
    int32_t C;
    float eps;
};

LWM_KERNEL(256) void rmsnorm_fwd_f32_kernel(RmsF32Params p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int nv = p.C >> 2;
    for (int64_t row = block_idx_x(); row < p.rows; row += grid_dim_x()) {
        const float* xr = p.x + row * p.C;
        f32x4 xs[8];
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = tid + 256 * k;
            if (v < nv) {
                xs[k] = global_load_f32x4(xr + v * 4);
                for (int j = 0; j < 4; ++j) ss = fmaf(xs[k][j], xs[k][j], ss);
            }
        }
        const float tot = block_sum_256(ss, lds, tid);
        const float r = 1.0f / sqrtf(tot / (float)p.C + p.eps);
        if (tid == 0 && p.rstd) p.rstd[row] = r;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = tid + 256 * k;
            if (v < nv) {
                const f32x4 w = global_load_f32x4(p.w + v * 4);
                f32x4 o;
                for (int j = 0; j < 4; ++j) o[j] = (xs[k][j] * r) * w[j];
                global_store_f32x4(p.y + row * p.C + v * 4, o);
            }
        }
    }
}

// dx = r * (dy - xhat * mean(dy * xhat)),  dy = g * w,  xhat = x * r;  dW[c] = sum_rows g * xhat (per workgroup -> dw_part)
LWM_KERNEL(256) void rmsnorm_bwd_f32_kernel(RmsF32Params p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int nv = p.C >> 2;
    f32x4 dw[8];
    for (int k = 0; k < 8; ++k) dw[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t row = block_idx_x(); row < p.rows; row += grid_dim_x()) {
        const float r = p.rstd[row];
        f32x4 xh[8], dy[8];
        float dot = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = tid + 256 * k;
            if (v < nv) {
                const f32x4 x = global_load_f32x4(p.x + row * p.C + v * 4);
                const f32x4 g = global_load_f32x4(p.g + row * p.C + v * 4);
                const f32x4 w = global_load_f32x4(p.w + v * 4);
                for (int j = 0; j < 4; ++j) {
                    xh[k][j] = x[j] * r;
                    dy[k][j] = g[j] * w[j];
                    dw[k][j] = fmaf(g[j], xh[k][j], dw[k][j]);
                    dot = fmaf(dy[k][j], xh[k][j], dot);
                }
            }
        }
        const float mean = block_sum_256(dot, lds, tid) / (float)p.C;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = tid + 256 * k;
            if (v < nv) {
                f32x4 o;
                for (int j = 0; j < 4; ++j) o[j] = r * (dy[k][j] - xh[k][j] * mean);
                global_store_f32x4(p.y + row * p.C + v * 4, o);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int v = tid + 256 * k;
        if (v < nv) global_store_f32x4(p.dw_part + (int64_t)block_idx_x() * p.C + v * 4, dw[k]);
    }
}

// dw[c] = sum_blocks dw_part[block][c]: the fixed tree of rmsnorm_dw_reduce_kernel (32 columns x 8 row groups), f32 out
LWM_KERNEL(256) void rmsnorm_dw_reduce_f32_kernel(const float* part, float* dw, int nblk, int C) {
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
        dw[col] = t;
    }
}

LWM_KERNEL(256) void swiglu_fwd_f32_kernel(const float* a, const float* b, float* y, int64_t n) {
    const int64_t nvec = n >> 2;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < nvec; i += (int64_t)grid_dim_x() * 256) {
        const f32x4 av = global_load_f32x4(a + i * 4), bv = global_load_f32x4(b + i * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = av[j] * sigmoid_fast(av[j]) * bv[j];
        global_store_f32x4(y + i * 4, o);
    }
}

LWM_KERNEL(256) void swiglu_bwd_f32_kernel(const float* a, const float* b, const float* g, float* da, float* db, int64_t n) {
    const int64_t nvec = n >> 2;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < nvec; i += (int64_t)grid_dim_x() * 256) {
        const f32x4 av = global_load_f32x4(a + i * 4), bv = global_load_f32x4(b + i * 4), gv = global_load_f32x4(g + i * 4);
        f32x4 oa, ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = sigmoid_fast(av[j]);
            ob[j] = gv[j] * av[j] * sg;
            oa[j] = gv[j] * bv[j] * sg * (1.0f + av[j] * (1.0f - sg));
        }
        global_store_f32x4(da + i * 4, oa);
        global_store_f32x4(db + i * 4, ob);
    }
}

// one workgroup (512 threads) per row of V <= 32768 floats, read from HBM once into registers (16 vectors of 4 per thread)
struct CeF32Params {
    const float* logits;    // [rows, V]
    const int32_t* target;  // [rows]
    const float* weight;    // [rows] gradient weight or null (= 1)
    float* nll;             // [rows]
    int32_t* correct;       // [rows] or null
    float* dlogits;         // [rows, V] or null
    int64_t rows;
    int32_t V;
};

LWM_KERNEL(kCeThreads) void softmax_ce_f32_kernel(CeF32Params p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int nv = p.V >> 2;
    for (int64_t row = block_idx_x(); row < p.rows; row += grid_dim_x()) {
        const float* lr = p.logits + row * p.V;
        f32x4 x[16];
        float mx = -INFINITY;
        int amax = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int v = tid + kCeThreads * k;
            if (v < nv) {
                x[k] = global_load_f32x4(lr + v * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x[k][j] > mx) {          // strict: keeps the first maximum of this thread
                        mx = x[k][j];
                        amax = v * 4 + j;
                    }
            }
        }
        for (int m = 1; m < 64; m <<= 1) {       // larger value wins, ties -> smaller index
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
        for (int k = 0; k < 16; ++k)
            if (tid + kCeThreads * k < nv)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[k][j] = fast_exp2((x[k][j] - mx) * 1.4426950408889634f);
                    se += x[k][j];
                }
        const float tot = block_sum<kCeThreads>(se, lds + 64, tid);
        const int tg = p.target[row];
        const float wgt = p.weight ? p.weight[row] : 1.0f;
        if (tid == 0) {
            const float lt = (tg >= 0 && tg < p.V) ? lr[tg] : 0.0f;
            p.nll[row] = (mx + logf(tot)) - lt;
            if (p.correct) p.correct[row] = (amax == tg) ? 1 : 0;
        }
        if (p.dlogits) {
            const float s = wgt / tot;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int v = tid + kCeThreads * k;
                if (v < nv) {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = x[k][j] * s - (v * 4 + j == tg ? wgt : 0.0f);
                    global_store_f32x4(p.dlogits + row * p.V + v * 4, o);
                }
            }
        }
        block_sync();       // the reduction scratch is reused by the next row
    }
}

// dst[i] = ((src0[i] + src1[i]) + src2[i]) + ... in argument order (n % 4 == 0)
LWM_KERNEL(kCastThreads) void sum_f32_kernel(SumSrcs srcs, int n_src, float* dst, int64_t n) {
    const int64_t nvec = n >> 2;
    int64_t i = (int64_t)block_idx_x() * kCastThreads + thread_idx();
    const int64_t step = (int64_t)grid_dim_x() * kCastThreads;
    for (; i < nvec; i += step) {
        f32x4 a = global_load_f32x4(srcs.p[0] + i * 4);
        for (int s = 1; s < n_src; ++s) a += global_load_f32x4(srcs.p[s] + i * 4);
        global_store_f32x4(dst + i * 4, a);
    }
}

}  // namespace lwm
