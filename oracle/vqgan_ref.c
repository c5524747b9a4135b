/* vqgan_ref.c -- CPU oracle for the VQGAN video-tokeniser primitives.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library (oracle/vqgan_ref.py).
 * Nothing under lwm_amd/ does.
 *
 * PARITY: the ARITHMETIC below is unpinned -- the reference (lwm/vqgan.py) is flax/JAX code, jax and flax
 * cannot be installed here and the reference ships no tests or golden vectors; XLA's
 * rounding is not reproducible.  How these primitives are WIRED into the tokeniser, and the quantiser as a whole, are
 * pinned to a run of the reference's own module code (oracle/vqgan_ref.py header; tests/golden/gen_ref_run_golden.py).
 * This file restates, in plain C with f32 arithmetic, what lwm/vqgan.py asks flax to compute:
 *
 *   ref_conv2d      nn.Conv(features, [k,k]) NHWC, kernel HWIO, bias on,
 *                   padding SAME (lwm/vqgan.py:155,163,172-175,183,253,257,262,
 *                   :114-115); Downsample = pad (0,1),(0,1) + stride-2 VALID
 *                   (lwm/vqgan.py:291-300); Upsample = jax.image.resize
 *                   nearest x2 + conv (lwm/vqgan.py:312-318), expressed with
 *                   `up_shift`; residual add (lwm/vqgan.py:263); final clip
 *                   (lwm/vqgan.py:141).
 *   ref_groupnorm   nn.GroupNorm() defaults: 32 groups, eps 1e-6, affine
 *                   (lwm/vqgan.py:161,181,251,254), optional nn.silu
 *                   (lwm/vqgan.py:162,182,252,255).
 *   ref_vq_argmin   VectorQuantizer distances + argmin (lwm/vqgan.py:207-212)
 *   ref_vq_gather   codebook lookup (lwm/vqgan.py:193-195) and the forward
 *                   value of z + stop_gradient(z_q - z) (lwm/vqgan.py:214).
 *
 * f32 addition is not associative and the reference leaves the order to XLA, so
 * the restatement fixes one:
 *   conv   out = ((P_0 + P_1) + ... + P_{T-1}) + bias [+ residual], taps t in
 *          (kh, kw) raster order, P_t = fmaf chain over c_in = 0..Cin-1 from 0;
 *   GN     sum and sum of squares in f64 over the group, mean/var in f64,
 *          var = max(0, E[x^2] - E[x]^2) (flax use_fast_variance), mean and
 *          1/sqrt(var+eps) rounded to f32; y = fmaf(x - mean, rstd*gamma, beta);
 *   SiLU   y * (1 / (1 + exp(-y))) with the f32 exp below (Cody-Waite + degree-6
 *          Horner, fmaf only) so that the value does not depend on a libm;
 *   VQ     d = (sum z^2 + sum e^2) - 2 * (z.e), each sum an fmaf chain over
 *          d = 0..D-1 from 0, argmin = first minimum.
 * tests/test_vqgan_oracle.py checks these against an independent float64
 * PyTorch implementation.
 *
 * Build: oracle/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* exp(x) for x in [-87, 88], f32, deterministic (no libm). */
static inline float ref_expf(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    const float k = rintf(x * 1.44269504088896341f);
    float r = fmaf(k, -0.693145751953125f, x);        /* ln2 high part (exact product) */
    r = fmaf(k, -1.42860682030941723212e-6f, r);      /* ln2 low part */
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int32_t)k + 127) << 23;         /* 2^k, k in [-126, 127] */
    return p * s.f;
}

float ref_expf_scalar(float x) { return ref_expf(x); }

static inline float ref_silu(float y) {
    const float e = ref_expf(-y);
    const float sig = 1.0f / (1.0f + e);
    return y * sig;
}

/* nn.silu on its own (lwm/vqgan.py:162,182,252,255 apply it to the GroupNorm's output): the same arithmetic as the fused
 * form of ref_groupnorm below -- used where the reference's own module code is executed with these primitives standing in
 * for flax's (tests/golden/gen_ref_run_golden.py). */
void ref_silu_array(const float* x, float* y, long n) {
    for (long i = 0; i < n; ++i) y[i] = ref_silu(x[i]);
}

/* x: [B,Hin,Win,Cin]  w: [KH,KW,Cin,Cout]  bias: [Cout] or NULL
 * residual: [B,Ho,Wo,Cout] or NULL   y: [B,Ho,Wo,Cout]
 * virtual input = x upsampled (nearest) by 2^up_shift; tap (kh,kw) of output
 * (oy,ox) reads virtual (oy*stride + kh - pad, ox*stride + kw - pad), zero
 * outside [0, Hin<<up_shift) x [0, Win<<up_shift). */
void ref_conv2d(const float* x, const float* w, const float* bias, const float* residual, float* y,
                int B, int Hin, int Win, int Cin, int Cout, int KH, int KW, int stride, int pad,
                int up_shift, int Ho, int Wo, int clip) {
    const int Hv = Hin << up_shift, Wv = Win << up_shift;
    const long npix = (long)B * Ho * Wo;
#pragma omp parallel
    {
        float* s = (float*)malloc(sizeof(float) * Cout);
        float* p = (float*)malloc(sizeof(float) * Cout);
#pragma omp for schedule(static)
        for (long m = 0; m < npix; ++m) {
            const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long)Wo * Ho));
            for (int co = 0; co < Cout; ++co) s[co] = 0.0f;
            for (int kh = 0; kh < KH; ++kh)
                for (int kw = 0; kw < KW; ++kw) {
                    const int vy = oy * stride + kh - pad, vx = ox * stride + kw - pad;
                    if (vy < 0 || vy >= Hv || vx < 0 || vx >= Wv) continue; /* P_t = 0 */
                    const float* xr =
                        x + (((long)b * Hin + (vy >> up_shift)) * Win + (vx >> up_shift)) * Cin;
                    const float* wt = w + (long)(kh * KW + kw) * Cin * Cout;
                    for (int co = 0; co < Cout; ++co) p[co] = 0.0f;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float xv = xr[ci];
                        const float* wr = wt + (long)ci * Cout;
                        for (int co = 0; co < Cout; ++co) p[co] = fmaf(xv, wr[co], p[co]);
                    }
                    for (int co = 0; co < Cout; ++co) s[co] = s[co] + p[co];
                }
            float* yr = y + m * Cout;
            const float* rr = residual ? residual + m * Cout : NULL;
            for (int co = 0; co < Cout; ++co) {
                float v = s[co];
                if (bias) v = v + bias[co];
                if (rr) v = v + rr[co];
                if (clip) v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
                yr[co] = v;
            }
        }
        free(s);
        free(p);
    }
}

/* x,y: [B,HW,C]; G groups of C/G contiguous channels; stats over (HW, C/G). */
void ref_groupnorm(const float* x, const float* gamma, const float* beta, float* y, int B, long HW,
                   int C, int G, float eps, int silu) {
    const int cg = C / G;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
            const float* xb = x + (long)b * HW * C + (long)g * cg;
            float* yb = y + (long)b * HW * C + (long)g * cg;
            double sum = 0.0, sq = 0.0;
            for (long p = 0; p < HW; ++p)
                for (int c = 0; c < cg; ++c) {
                    const double v = (double)xb[p * C + c];
                    sum += v;
                    sq += v * v;
                }
            const double n = (double)HW * cg;
            const double mean = sum / n;
            double var = sq / n - mean * mean;
            if (var < 0.0) var = 0.0;
            const float mean_f = (float)mean;
            const float rstd_f = (float)(1.0 / sqrt(var + (double)eps));
            for (long p = 0; p < HW; ++p)
                for (int c = 0; c < cg; ++c) {
                    const float mul = rstd_f * gamma[g * cg + c];
                    float v = fmaf(xb[p * C + c] - mean_f, mul, beta[g * cg + c]);
                    if (silu) v = ref_silu(v);
                    yb[p * C + c] = v;
                }
        }
}

/* z: [N,D], codebook: [E,D] -> idx[N] (first minimum of the f32 distances) */
void ref_vq_argmin(const float* z, const float* codebook, int32_t* idx, long N, int E, int D) {
    float* se = (float*)malloc(sizeof(float) * E);
    for (int e = 0; e < E; ++e) {
        float s = 0.0f;
        for (int d = 0; d < D; ++d) s = fmaf(codebook[(long)e * D + d], codebook[(long)e * D + d], s);
        se[e] = s;
    }
#pragma omp parallel for schedule(static)
    for (long n = 0; n < N; ++n) {
        const float* zr = z + n * D;
        float sz = 0.0f;
        for (int d = 0; d < D; ++d) sz = fmaf(zr[d], zr[d], sz);
        float best = INFINITY;
        int32_t bi = 0;
        for (int e = 0; e < E; ++e) {
            const float* er = codebook + (long)e * D;
            float t = 0.0f;
            for (int d = 0; d < D; ++d) t = fmaf(er[d], zr[d], t);
            const float dist = (sz + se[e]) - 2.0f * t;
            if (dist < best) {
                best = dist;
                bi = e;
            }
        }
        idx[n] = bi;
    }
    free(se);
}

/* out[n] = codebook[idx[n]] (z == NULL) or z[n] + (codebook[idx[n]] - z[n]) */
void ref_vq_gather(const float* codebook, const int32_t* idx, const float* z, float* out, long N,
                   int D) {
    for (long n = 0; n < N; ++n)
        for (int d = 0; d < D; ++d) {
            const float e = codebook[(long)idx[n] * D + d];
            out[n * D + d] = z ? z[n * D + d] + (e - z[n * D + d]) : e;
        }
}
