// gemv.h -- y[r, :] = x[r, :] . W for a handful of rows (the projections of a cached-decode step: one token per
// batch row against the (K, N) bf16 kernels of wq/wk/wv/wo, w1/w2/w3 and lm_head -- lwm/llama.py:427-432, :659,
// :1075-1106 with q_len = 1).  Requires wave_ops.h.
//
// HBM-bound: W is read exactly once, 2*K*N bytes; everything else is noise.  A library GEMM at M = 1 spends a
// launch (or two, with split-K) per projection and streams W at a fraction of the HBM rate; here:
//   * workgroup = a (128 rows of K) x (512 columns of N) tile of W; wave w walks rows 32w .. 32w+31, lane l owns
//     columns 8l .. 8l+7: every wave load is 1 KiB contiguous (one row of the tile), eight of them in flight;
//   * x[r, k] reaches the FMAs as a scalar (v_readlane of a register that holds the wave's 32 x values);
//   * the four waves' sums meet in LDS in wave order, the workgroup writes f32 partials [K/128][R][N];
//   * gemv_reduce_kernel adds the K/128 partials of an output along a FIXED tree (deterministic; no atomics) and
//     writes bf16 and / or f32;
//   * up to three matrices that share x (wq | wk | wv, w1 | w3) ride in ONE pair of launches;
//   * the small launches either side of a projection can ride along (lwm_gemv_fused_bf16; a decode step is a dozen
//     3-5 us launches per layer next to ~90 us of weight streaming):
//       - RMSNorm ON LOAD: x is normalised as it is read -- bf16(bf16(x * rstd) * gamma), the arithmetic of
//         rmsnorm_fwd_kernel (lwm/llama.py:320-341) -- with rstd from partial sums of squares that the reduction
//         which PRODUCED x left behind (ss_in: a few dozen floats per row, summed along a fixed tree by every wave);
//       - RESIDUAL ADD in the reduction: y = bf16(bf16(x . W) + residual), the bf16 add the block would launch next
//         (lwm/llama.py:719, :737), and the partial sums of squares of y for the next norm (ss_out).
// R <= 4 rows, N % 8 == 0, K % 32 == 0, K <= 12288 (LWM-7B: 4096, 11008, 32000 all qualify).
#pragma once

namespace lwm {

constexpr int kGemvThreads = 256;
constexpr int kGemvRPW = 32;           // rows of W per wave (64, and `nt` loads, measured: 5.28 / 5.36 vs 5.46 TB/s)
constexpr int kGemvKT = 4 * kGemvRPW;  // rows of W per workgroup
constexpr int kGemvNT = 512;       // columns of W per workgroup
constexpr int kGemvMaxRows = 4;

constexpr int kGemvMaxMats = 3;    // matrices that share one x in a launch (wq | wk | wv, w1 | w3)

struct GemvParams {
    const bf16_t* x;                  // [R, K], row stride ldx
    const bf16_t* w[kGemvMaxMats];    // [K, N_i] dense
    bf16_t* y[kGemvMaxMats];          // [R, N_i] row stride ldy[i], or null
    float* y_f32[kGemvMaxMats];       // [R, N_i] dense, or null
    float* part;                      // per matrix i at part_off[i]: [KS][R][N_i]
    int64_t ldx, ldy[kGemvMaxMats], part_off[kGemvMaxMats];
    int32_t N[kGemvMaxMats], blk0[kGemvMaxMats + 1], quad0[kGemvMaxMats + 1];   // first workgroup / first reduce quad of matrix i
    int32_t R, K, KS, nmat;
    // fused neighbours (all optional)
    const bf16_t* gamma;              // RMSNorm weight [K]: x is normalised on load
    const float* ss_in;               // [R][ss_n] partial sums of squares of x's rows (their sum = sum_k x[r,k]^2)
    int32_t ss_n;                     // <= 64
    float eps;
    const bf16_t* res[kGemvMaxMats];  // residual [R, N_i] (row stride ldres[i]) added in the reduction
    int64_t ldres[kGemvMaxMats];
    float* ss_out;                    // [R][N_0 / 128]: partial sums of squares of matrix 0's bf16 output rows (N_0 % 128 == 0)
};

constexpr int kGemvSsCols = 128;      // output columns per reduce workgroup = per ss_out partial

LWM_DEVICE float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
LWM_DEVICE float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

template <int R>
LWM_DEVICE void gemv_body(const GemvParams& p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = wave_uniform(tid >> 6), lane = tid & 63;
    int mi = 0;                                    // which matrix this workgroup belongs to (uniform)
    for (int i = 1; i < p.nmat; ++i) mi = block_idx_x() >= p.blk0[i] ? i : mi;
    const int N = p.N[mi];
    const int nbn = (N + kGemvNT - 1) / kGemvNT;
    const int bl = block_idx_x() - p.blk0[mi];
    const int ks = bl / nbn, nb = bl % nbn;
    const int k0 = ks * kGemvKT + wave * kGemvRPW;
    int n = nb * kGemvNT + lane * 8;
    const bool n_ok = n < N;
    n = n_ok ? n : N - 8;                          // (clamped: the loads stay inside the matrix)
    // this wave's 32 x values per row r, one per lane (lanes 32..63 repeat)
    float xv[R];
    for (int r = 0; r < R; ++r) {
        const int k = k0 + (lane & (kGemvRPW - 1));
        const int kc = k < p.K ? k : p.K - 1;
        const bf16_t raw = p.x[(int64_t)r * p.ldx + kc];
        float xf = bf16_lo((uint32_t)__builtin_bit_cast(uint16_t, raw));
        if (p.gamma) {
            // rstd of row r from the partials (every wave sums the same <= 64 numbers along the same tree)
            float t = lane < p.ss_n ? p.ss_in[(int64_t)r * p.ss_n + lane] : 0.0f;
            for (int m = 1; m < 64; m <<= 1) t += shfl_xor_f(t, m);
            const float rstd = 1.0f / sqrtf(t / (float)p.K + p.eps);
            const float g = bf16_lo((uint32_t)__builtin_bit_cast(uint16_t, p.gamma[kc]));
            xf = (float)(bf16_t)((float)(bf16_t)(xf * rstd) * g);
        }
        xv[r] = k < p.K ? xf : 0.0f;
    }
    float acc[R][8];
    for (int r = 0; r < R; ++r)
        for (int j = 0; j < 8; ++j) acc[r][j] = 0.0f;
    const bf16_t* wp = p.w[mi] + n;
#pragma unroll
    for (int i0 = 0; i0 < kGemvRPW; i0 += 8) {
        u32x4 wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + i0 + u;
            wv[u] = global_load_b128(wp + (int64_t)(k < p.K ? k : p.K - 1) * N);        // rows past K meet x = 0
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float xs = lane_value(xv[r], i0 + u);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[r][2 * c] = fmaf(xs, bf16_lo(wv[u][c]), acc[r][2 * c]);
                    acc[r][2 * c + 1] = fmaf(xs, bf16_hi(wv[u][c]), acc[r][2 * c + 1]);
                }
            }
    }
    // waves 1..3 hand their sums to wave 0 through LDS; wave 0 adds them in wave order
    if (wave > 0) {
        for (int r = 0; r < R; ++r) {
            const lds_t slot = lds + (uint32_t)(((wave - 1) * R + r) * kGemvNT + lane * 8) * 4;
            lds_write_f32x4(slot, f32x4{acc[r][0], acc[r][1], acc[r][2], acc[r][3]});
            lds_write_f32x4(slot + 16, f32x4{acc[r][4], acc[r][5], acc[r][6], acc[r][7]});
        }
    }
    block_sync();
    if (wave == 0 && n_ok) {
        for (int r = 0; r < R; ++r) {
            f32x4 lo = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]}, hi = {acc[r][4], acc[r][5], acc[r][6], acc[r][7]};
            for (int w = 0; w < 3; ++w) {
                const lds_t slot = lds + (uint32_t)((w * R + r) * kGemvNT + lane * 8) * 4;
                lo = lo + lds_read_f32x4(slot);
                hi = hi + lds_read_f32x4(slot + 16);
            }
            float* dst = p.part + p.part_off[mi] + ((int64_t)ks * R + r) * N + n;
            global_store_f32x4(dst, lo);
            global_store_f32x4(dst + 4, hi);
        }
    }
}

LWM_KERNEL(kGemvThreads) void gemv_bf16_kernel(GemvParams p) {
    switch (p.R) {            // (uniform; the row count is a compile-time constant inside each body)
        case 1: gemv_body<1>(p); break;
        case 2: gemv_body<2>(p); break;
        case 3: gemv_body<3>(p); break;
        default: gemv_body<4>(p); break;
    }
}

// y[r, n..n+3] = sum over the KS partials.  Eight lanes per output quad: lane j adds partials j, j+8, ... (all
// its loads in flight at once), then the eight sums meet by xor-shuffles -- a FIXED tree, so the result is
// deterministic (one thread walking 32..86 partials four at a time took 5 us per launch, a quarter of a decode step).
LWM_KERNEL(256) void gemv_reduce_kernel(GemvParams p) {
    const int64_t t = (int64_t)block_idx_x() * 256 + thread_idx();
    const int sub = (int)(t & 7);
    int64_t quad = t >> 3;
    const bool live = quad < p.quad0[p.nmat];
    quad = live ? quad : p.quad0[p.nmat] - 1;                  // (idle lanes repeat the last quad: the shuffles need every lane)
    int mi = 0;
    for (int i = 1; i < p.nmat; ++i) mi = quad >= p.quad0[i] ? i : mi;
    const int N = p.N[mi], nq = N >> 2;
    const int64_t q = quad - p.quad0[mi];
    const int r = (int)(q / nq), n = (int)(q % nq) * 4;
    const float* src = p.part + p.part_off[mi] + (int64_t)r * N + n;
    const int64_t step = (int64_t)p.R * N;
    constexpr int kMaxPer = 12;                                 // K <= 8 * 12 * 128 = 12288
    f32x4 v[kMaxPer];
#pragma unroll
    for (int u = 0; u < kMaxPer; ++u) {
        const int k = sub + 8 * u;
        v[u] = global_load_f32x4(src + (int64_t)(k < p.KS ? k : 0) * step);
    }
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < kMaxPer; ++u)
        if (sub + 8 * u < p.KS) s = s + v[u];
    for (int m = 1; m < 8; m <<= 1)
        for (int j = 0; j < 4; ++j) s[j] = s[j] + shfl_xor_f(s[j], m);
    const bool writer = live && sub == 0;
    float sq = 0.0f;
    if (writer) {
        if (p.y_f32[mi]) global_store_f32x4(p.y_f32[mi] + (int64_t)r * N + n, s);
        if (p.y[mi]) {
            float o[4];
            for (int j = 0; j < 4; ++j) o[j] = (float)(bf16_t)s[j];
            if (p.res[mi]) {
                const u32x2 rr = *(const u32x2*)(p.res[mi] + (int64_t)r * p.ldres[mi] + n);
                o[0] = (float)(bf16_t)(o[0] + bf16_lo(rr[0]));
                o[1] = (float)(bf16_t)(o[1] + bf16_hi(rr[0]));
                o[2] = (float)(bf16_t)(o[2] + bf16_lo(rr[1]));
                o[3] = (float)(bf16_t)(o[3] + bf16_hi(rr[1]));
            }
            global_store_b64(p.y[mi] + (int64_t)r * p.ldy[mi] + n, u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])});
            for (int j = 0; j < 4; ++j) sq = fmaf(o[j], o[j], sq);
        }
    }
    if (p.ss_out) {
        // one partial per workgroup = per 128 output columns of one row (N_0 % 128 == 0, one matrix: checked by the host)
        const float tot = block_sum_256(sq, dyn_lds(), thread_idx());
        if (thread_idx() == 0) {
            const int64_t wg = block_idx_x();
            const int per_row = p.N[0] / kGemvSsCols;
            if (wg < (int64_t)p.R * per_row) p.ss_out[wg] = tot;        // [r][wg % per_row], r = wg / per_row
        }
    }
}

}  // namespace lwm
