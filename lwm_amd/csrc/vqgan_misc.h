// vqgan_misc.h -- GroupNorm(+SiLU), vector-quantiser argmin and codebook gather
// for the VQGAN tokeniser.  Requires wave_ops.h + attn_common.h (fragment helpers).
//
// Arithmetic contracts are those of oracle/vqgan_ref.c (restating flax
// nn.GroupNorm / nn.silu as used at lwm/vqgan.py:161-162,:181-182,:251-255 and
// VectorQuantizer, lwm/vqgan.py:187-221), so results are bit-exact wherever the
// contract is order-free (everything except the f64 statistics sums, which are
// rounded to f32 after the reduction).
#pragma once

namespace lwm {

// ---------------------------------------------------------------- exp / SiLU
// exp(x) on [-87, 88] in f32: Cody-Waite reduction + degree-6 Horner, fmaf only
// (same constants and operation order as oracle/vqgan_ref.c: no libm/ocml call).
LWM_DEVICE float det_expf(float x) {
    x = x > 88.0f ? 88.0f : x;
    x = x < -87.0f ? -87.0f : x;
    const float k = rintf(x * 1.44269504088896341f);
    float r = fmaf(k, -0.693145751953125f, x);
    r = fmaf(k, -1.42860682030941723212e-6f, r);
    float q = 1.0f / 720.0f;
    q = fmaf(q, r, 1.0f / 120.0f);
    q = fmaf(q, r, 1.0f / 24.0f);
    q = fmaf(q, r, 1.0f / 6.0f);
    q = fmaf(q, r, 0.5f);
    q = fmaf(q, r, 1.0f);
    q = fmaf(q, r, 1.0f);
    const uint32_t bits = (uint32_t)((int32_t)k + 127) << 23;
    return q * __builtin_bit_cast(float, bits);
}

LWM_DEVICE float det_silu(float y) {
    const float e = det_expf(-y);
    const float sig = 1.0f / (1.0f + e);
    return y * sig;
}

// ---------------------------------------------------------------- GroupNorm
// x, y: [B, HW, C] f32; G groups of cg = C/G contiguous channels (cg % 4 == 0).
// Pass 1 (gn_stats): each workgroup reduces a slice of pixels to per-group
//   f64 (sum, sum of squares) partials  part[b][slice][g][2].
// Pass 1b (gn_finalize, one thread per (b, group)): sums the slice partials in
//   slice order (f64), derives mean / rstd and rounds them to f32.
// Pass 2 (gn_apply): streams a pixel slice:
//   y = fmaf(x - mean, rstd*gamma, beta) [-> SiLU].
// Passes 1 and 2 are HBM-bound: 16-byte loads/stores, one channel quad per thread,
// kGnUnroll pixel rows in flight per thread (one load per iteration leaves a wave
// with 1 KiB outstanding: 1.9 TB/s on the 1 GiB tensors, measured).
struct GnParams {
    const float* x;
    const float* gamma;
    const float* beta;
    float* y;
    double* part;   // [B, NS, G, 2]
    float* stats;   // [B, G, 2]: mean, rstd (gn_finalize -> gn_apply)
    int32_t B, C, G, silu, NS;
    int64_t HW, slice;  // pixels per slice (stats) -- NS = ceil(HW/slice)
    int64_t aslice;     // pixels per workgroup in gn_apply
    float eps;
};

constexpr int kGnThreads = 256;
constexpr int kGnUnroll = 4;

LWM_KERNEL(kGnThreads) void gn_stats_kernel(GnParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int tpr = p.C >> 2;            // threads (channel quads) per pixel row
    const int rpp = kGnThreads / tpr;    // pixel rows per pass
    const int cg4 = (p.C / p.G) >> 2;    // quads per group
    const int q = tid % tpr, prow = tid / tpr;
    const int b = block_idx_x() / p.NS, sl = block_idx_x() % p.NS;
    const int64_t p0 = (int64_t)sl * p.slice;
    const int64_t p1 = p0 + p.slice < p.HW ? p0 + p.slice : p.HW;
    double s = 0.0, ss = 0.0;
    if (prow < rpp) {
        const float* xb = p.x + ((int64_t)b * p.HW) * p.C + q * 4;
        int64_t px = p0 + prow;
        for (; px + (int64_t)(kGnUnroll - 1) * rpp < p1; px += (int64_t)kGnUnroll * rpp) {
            f32x4 v[kGnUnroll];
#pragma unroll
            for (int u = 0; u < kGnUnroll; ++u) v[u] = global_load_f32x4(xb + (px + (int64_t)u * rpp) * p.C);
#pragma unroll
            for (int u = 0; u < kGnUnroll; ++u)      // same per-thread order as one row at a time
                for (int j = 0; j < 4; ++j) {
                    const double d = (double)v[u][j];
                    s += d;
                    ss += d * d;
                }
        }
        for (; px < p1; px += rpp) {
            f32x4 v = global_load_f32x4(xb + px * p.C);
            for (int j = 0; j < 4; ++j) {
                const double d = (double)v[j];
                s += d;
                ss += d * d;
            }
        }
    }
    lds_write_f64(lds + tid * 16, s);
    lds_write_f64(lds + tid * 16 + 8, ss);
    block_sync();
    if (tid < p.G) {
        double ts = 0.0, tss = 0.0;
        for (int r = 0; r < rpp; ++r)
            for (int c = 0; c < cg4; ++c) {
                const int t = r * tpr + tid * cg4 + c;
                ts += lds_read_f64(lds + t * 16);
                tss += lds_read_f64(lds + t * 16 + 8);
            }
        double* o = p.part + (((int64_t)b * p.NS + sl) * p.G + tid) * 2;
        o[0] = ts;
        o[1] = tss;
    }
}

// grid = B*G workgroups of 64 threads, LDS = NS * 16 bytes: the lanes fetch the (b, g) column of the slice
// partials side by side, then ONE lane adds them in slice order (the contract's order) out of LDS -- a
// single thread walking the 256 slices through global memory took 40 us per launch.
LWM_KERNEL(64) void gn_finalize_kernel(GnParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int b = block_idx_x() / p.G, g = block_idx_x() % p.G;
    for (int i = tid; i < p.NS; i += 64) {
        const double* o = p.part + (((int64_t)b * p.NS + i) * p.G + g) * 2;
        lds_write_f64(lds + i * 16, o[0]);
        lds_write_f64(lds + i * 16 + 8, o[1]);
    }
    block_sync();
    if (tid != 0) return;
    double ts = 0.0, tss = 0.0;
    for (int i = 0; i < p.NS; ++i) {
        ts += lds_read_f64(lds + i * 16);
        tss += lds_read_f64(lds + i * 16 + 8);
    }
    const double n = (double)p.HW * (double)(p.C / p.G);
    const double mean = ts / n;
    double var = tss / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    p.stats[block_idx_x() * 2] = (float)mean;
    p.stats[block_idx_x() * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
}

LWM_DEVICE f32x4 gn_apply_quad(f32x4 v, float mean, const float (&mul)[4], const float (&bet)[4], int silu) {
    f32x4 o;
    for (int j = 0; j < 4; ++j) {
        float t = fmaf(v[j] - mean, mul[j], bet[j]);
        if (silu) t = det_silu(t);
        o[j] = t;
    }
    return o;
}

LWM_KERNEL(kGnThreads) void gn_apply_kernel(GnParams p) {
    const int tid = thread_idx();
    const int tpr = p.C >> 2, rpp = kGnThreads / tpr;
    const int cg = p.C / p.G;
    const int q = tid % tpr, prow = tid / tpr;
    const int64_t nas = (p.HW + p.aslice - 1) / p.aslice;
    const int b = (int)(block_idx_x() / nas);
    const int64_t sl = block_idx_x() % nas;
    if (prow >= rpp) return;
    const int g = (q * 4) / cg;
    const float mean = p.stats[(b * p.G + g) * 2], rstd = p.stats[(b * p.G + g) * 2 + 1];
    float mul[4], bet[4];
    for (int j = 0; j < 4; ++j) {
        mul[j] = rstd * p.gamma[q * 4 + j];
        bet[j] = p.beta[q * 4 + j];
    }
    const int64_t p0 = sl * p.aslice;
    const int64_t p1 = p0 + p.aslice < p.HW ? p0 + p.aslice : p.HW;
    const int64_t base = ((int64_t)b * p.HW) * p.C + q * 4;
    int64_t px = p0 + prow;
    for (; px + (int64_t)(kGnUnroll - 1) * rpp < p1; px += (int64_t)kGnUnroll * rpp) {
        f32x4 v[kGnUnroll];
#pragma unroll
        for (int u = 0; u < kGnUnroll; ++u) v[u] = global_load_f32x4(p.x + base + (px + (int64_t)u * rpp) * p.C);
#pragma unroll
        for (int u = 0; u < kGnUnroll; ++u)
            global_store_f32x4(p.y + base + (px + (int64_t)u * rpp) * p.C, gn_apply_quad(v[u], mean, mul, bet, p.silu));
    }
    for (; px < p1; px += rpp)
        global_store_f32x4(p.y + base + px * p.C, gn_apply_quad(global_load_f32x4(p.x + base + px * p.C), mean, mul, bet, p.silu));
}

// ---------------------------------------------------------------- VQ
// se[e] = fmaf chain over d of codebook[e][d]^2 (once per codebook).
LWM_KERNEL(256) void vq_sqnorm_kernel(const float* cb, float* se, int E, int D) {
    const int e = block_idx_x() * 256 + thread_idx();
    if (e >= E) return;
    float s = 0.0f;
    for (int d = 0; d < D; ++d) s = fmaf(cb[(int64_t)e * D + d], cb[(int64_t)e * D + d], s);
    se[e] = s;
}

// argmin_e (|z|^2 + |e|^2) - 2 z.e   (lwm/vqgan.py:207-212), D = 64.
// Workgroup = 8 waves x the same 32 z rows; wave w scans codes
// [w*E/8, (w+1)*E/8) in blocks of 32.  The dot products run on the exact-f32 MFMA
// with the CODEBOOK as the A operand (rows = codes) and z^T as B (cols = z
// rows): a lane then owns one z row and sees its 16 codes of the block in
// increasing index order, so the running (min, first index) is an in-lane scan
// and "first minimum" needs no cross-lane traffic until the very end.
constexpr int kVqD = 64;
constexpr int kVqThreads = 512;

LWM_KERNEL(kVqThreads) void vq_argmin_kernel(const float* z, const float* cb, const float* se,
                                             int32_t* idx, int64_t N, int E) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int64_t n = (int64_t)block_idx_x() * 32 + l31;
    const bool n_ok = n < N;
    // this lane's z row: B operand z[n][2s+hi], and |z|^2 as the ordered chain
    float zb[32];
    float sz = 0.0f;
    {
        const float* zr = z + (n_ok ? n : 0) * kVqD;
        for (int u = 0; u < 16; ++u) {
            f32x4 v = n_ok ? global_load_f32x4(zr + 4 * u) : zero_f32x4();
            const float f0 = v[0], f1 = v[1], f2 = v[2], f3 = v[3];
            sz = fmaf(f0, f0, sz);
            sz = fmaf(f1, f1, sz);
            sz = fmaf(f2, f2, sz);
            sz = fmaf(f3, f3, sz);
            zb[2 * u] = hi ? f1 : f0;
            zb[2 * u + 1] = hi ? f3 : f2;
        }
    }
    const int per_wave = (E + 7) / 8;
    const int e_begin = wave * per_wave;
    const int e_end = e_begin + per_wave < E ? e_begin + per_wave : E;
    float best = INFINITY;
    int32_t bidx = 0;
    for (int c0 = e_begin; c0 < e_end; c0 += 32) {
        const int code = c0 + l31;                   // A operand row of this lane
        const bool c_ok = code < e_end;
        const float* er = cb + (int64_t)(c_ok ? code : 0) * kVqD;
        f32x16 acc = zero_f32x16();
        for (int u = 0; u < 16; ++u) {
            f32x4 v = c_ok ? global_load_f32x4(er + 4 * u) : zero_f32x4();
            const float a0 = hi ? v[1] : v[0];
            const float a1 = hi ? v[3] : v[2];
            acc = mfma_32x32x2_f32(a0, zb[2 * u], acc);
            acc = mfma_32x32x2_f32(a1, zb[2 * u + 1], acc);
        }
        for (int r = 0; r < 16; ++r) {
            const int e = c0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (e < e_end) {
                const float d = (sz + se[e]) - 2.0f * acc[r];
                if (d < best) {
                    best = d;
                    bidx = e;
                }
            }
        }
    }
    // the other half-wave holds the interleaved other 16 codes of every block
    {
        const float ob = xhalf(best);
        const int32_t oi = shfl_xor_i(bidx, 32);
        if (ob < best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
    }
    if (hi == 0) {
        lds_write_f32(lds + (wave * 32 + l31) * 8, best);
        lds_write_i32(lds + (wave * 32 + l31) * 8 + 4, bidx);
    }
    block_sync();
    if (tid < 32 && n_ok) {
        float fb = INFINITY;
        int32_t fi = 0;
        for (int w = 0; w < 8; ++w) {       // waves hold increasing code ranges
            const float wb = lds_read_f32(lds + (w * 32 + tid) * 8);
            const int32_t wi = lds_read_i32(lds + (w * 32 + tid) * 8 + 4);
            if (wb < fb) {
                fb = wb;
                fi = wi;
            }
        }
        idx[n] = fi;
    }
}

// out[n] = cb[idx[n]]  (z == null: decode, lwm/vqgan.py:204-205)  or
// out[n] = z[n] + (cb[idx[n]] - z[n])  (encode's straight-through value, :214)
LWM_KERNEL(256) void vq_gather_kernel(const float* cb, const int32_t* idx, const float* z, float* out,
                                      int64_t N, int D, int E) {
    const int dq = D >> 2;
    const int64_t total = N * dq;
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < total;
         i += (int64_t)grid_dim_x() * 256) {
        const int64_t nrow = i / dq;
        const int c = (int)(i % dq) * 4;
        int32_t e = idx[nrow];
        e = e < 0 ? 0 : (e >= E ? E - 1 : e);
        f32x4 ev = global_load_f32x4(cb + (int64_t)e * D + c);
        if (z) {
            f32x4 zv = global_load_f32x4(z + nrow * D + c);
            for (int j = 0; j < 4; ++j) ev[j] = zv[j] + (ev[j] - zv[j]);
        }
        global_store_f32x4(out + nrow * D + c, ev);
    }
}

}  // namespace lwm
