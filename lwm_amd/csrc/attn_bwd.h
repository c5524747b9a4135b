// attn_bwd.h -- blockwise attention backward for one (q block, kv block) ring
// step on gfx950.  Requires wave_ops.h + attn_common.h.
//
// Replaces the custom-VJP backward of `ringattention` (call site
// lwm/llama.py:539-569): recompute p from the saved
// LSE, dv += p^T do, dp = do v^T, ds = p*(dp - rowsum(do*o)), dq += ds k,
// dk += ds^T q, all scaled by 1/sqrt(D) where the reference does.
//
// Three kernels, no atomics, deterministic:
//   attn_bwd_delta_kernel : delta[b,h,q] = sum_d dO*O               (HBM bound)
//   attn_bwd_dq_kernel    : workgroup owns 256 queries, streams K/V (3 GEMMs)
//   attn_bwd_dkdv_kernel  : workgroup owns 256 keys, streams Q/dO   (4 GEMMs)
// The f32 *_acc carries let a ring driver accumulate dq locally and dk/dv in
// buffers that travel with the K/V block.
#pragma once

namespace lwm {

// ------------------------------------------------------------------ delta
constexpr int kDeltaThreads = 256;

LWM_KERNEL(kDeltaThreads) void attn_bwd_delta_kernel(AttnParams p, float* delta) {
    const int tid = thread_idx();
    const int64_t rows = (int64_t)p.B * p.H * p.Sq;
    const int part = tid & 15;
    int64_t row = (int64_t)block_idx_x() * (kDeltaThreads / 16) + (tid >> 4);
    const int64_t row_step = (int64_t)grid_dim_x() * (kDeltaThreads / 16);
    // every lane runs the same number of iterations (shuffles need full waves)
    const int64_t iters = (rows + row_step - 1) / row_step;
    for (int64_t it = 0; it < iters; ++it, row += row_step) {
        float s = 0.0f;
        const bool ok = row < rows;
        int64_t b = 0, h = 0, q = 0;
        if (ok) {
            // row = (b*H + h)*Sq + q   (matches the [B,H,Sq] layout of delta/lse)
            q = row % p.Sq;
            int64_t bh = row / p.Sq;
            h = bh % p.H;
            b = bh / p.H;
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
        if (ok && part == 0) delta[row] = s;
    }
}

// ------------------------------------------------------------------ dQ
// LDS map: K tile 0 | K tile 1 | V tile 0 | V tile 1 (8 KiB each) | key meta 0 | 1
constexpr int kDqBQ = 256;
constexpr int kDqBK = 32;
constexpr int kDqThreads = 512;
constexpr int kDqTileBytes = kDqBK * kRowBytes;                 // 8 KiB
constexpr int kDqLdsBytes = 4 * kDqTileBytes + 2 * kDqBK * 4;  // K,V x2 + kseg x2

struct DqStage {
    u32x4 k;
    u32x4 v;
    int32_t kseg;
    uint8_t kvalid;
};

struct DqCtx {
    RowFragAddr ka;   // K row fragments (tile 0); V tile 0 is +2*kDqTileBytes
    TrFragAddr kta;   // K transposed fragments (tile 0)
    lds_t stage_w, kseg_w, kseg_r;
    int tid, hi;
    int64_t q_pos, wq_min, wq_max;
    int32_t seg_q;
    bool has_kmeta;
    float c, lse2, dlt;
};

// (meta loads first and mutually independent: see fwd_stage_load)
LWM_DEVICE void dq_stage_load(const AttnParams& p, const bf16_t* kb, const bf16_t* vb, int b,
                              int kt, int tid, DqStage& st) {
    if (tid < kDqBK) {
        int krow = kt * kDqBK + tid;
        int kr = krow < p.Sk ? krow : p.Sk - 1;
        st.kvalid = p.key_valid ? p.key_valid[(int64_t)b * p.Sk + kr] : (uint8_t)1;
        st.kseg = p.seg_k ? p.seg_k[(int64_t)b * p.Sk + kr] : 0;
    }
    int row = tid >> 4, slot = tid & 15;
    int krow = kt * kDqBK + row;
    int kr = krow < p.Sk ? krow : p.Sk - 1;
    st.k = global_load_b128(kb + (int64_t)kr * p.k_ss + slot * 8);
    st.v = global_load_b128(vb + (int64_t)kr * p.v_ss + slot * 8);
}

template <int BUF>
LWM_DEVICE void dq_stage_write(const DqCtx& cx, const DqStage& st, int kt, int Sk) {
    lds_write_b128(cx.stage_w + BUF * kDqTileBytes, st.k);
    lds_write_b128(cx.stage_w + (2 + BUF) * kDqTileBytes, st.v);
    if (cx.tid < kDqBK) {
        const bool ok = (kt * kDqBK + cx.tid < Sk) && st.kvalid != 0;
        lds_write_i32(cx.kseg_w + BUF * kDqBK * 4, ok ? st.kseg : kSegInvalid);
    }
}

template <int BUF>
LWM_DEVICE void dq_tile(const AttnParams& p, const DqCtx& cx, const bf16x8 (&qf)[8],
                        const bf16x8 (&dof)[8], int kt, f32x16 (&acc)[4]) {
    const int64_t k_pos0 = p.k_start + (int64_t)kt * kDqBK;
    if (p.causal && k_pos0 > cx.wq_max) return;
    constexpr uint32_t KB = BUF * kDqTileBytes;
    constexpr uint32_t VB = (2 + BUF) * kDqTileBytes;

    f32x16 st = zero_f32x16(), dpt = zero_f32x16();
    // fragment bases re-derived per tile (XOR form, see attn_common.h) and operand
    // fragments requested kRing-1 steps ahead through a register ring, pinned by
    // sched_fence (see dkv_tile)
    const uint32_t ka0 = opaque(cx.ka.a[0]);
    const uint32_t lo0 = opaque(cx.kta.lo[0]), up0 = opaque(cx.kta.up[0]);
    constexpr int kRing = 3;
    bf16x8 fa[kRing];
    auto load1 = [&](int g) { fa[g % kRing] = lds_read_b128(row_frag_at(ka0, g & 7) + (g < 8 ? KB : VB)); };
    prio_hi();
#pragma unroll
    for (int g = 0; g < kRing - 1; ++g) load1(g);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g + kRing - 1 < 16) load1(g + kRing - 1);
        sched_fence();
        if (g < 8) st = mfma_32x32x16(fa[g % kRing], qf[g], st);
        else dpt = mfma_32x32x16(fa[g % kRing], dof[g - 8], dpt);
        sched_fence();
    }
    prio_lo();
    const bool need_mask = cx.has_kmeta || (p.causal && k_pos0 + kDqBK - 1 > cx.wq_min);
    for (int r = 0; r < 16; ++r) st[r] = fast_exp2(fmaf(st[r], cx.c, -cx.lse2));
    if (need_mask) {
        int64_t rel64 = p.causal ? (cx.q_pos - k_pos0) : (int64_t)kDqBK;
        const int rel = rel64 > kDqBK ? kDqBK : (rel64 < -1 ? -1 : (int)rel64);
        for (int g = 0; g < 4; ++g) {
            const int kl0 = 8 * g + 4 * cx.hi;
            if (cx.has_kmeta) {
                u32x4 sg = lds_read_u32x4(cx.kseg_r + BUF * kDqBK * 4 + 8 * g * 4);
                for (int j = 0; j < 4; ++j) {
                    bool vis = ((int32_t)sg[j] == cx.seg_q) && (kl0 + j <= rel);
                    st[4 * g + j] = vis ? st[4 * g + j] : 0.0f;
                }
            } else {
                for (int j = 0; j < 4; ++j)
                    st[4 * g + j] = (kl0 + j <= rel) ? st[4 * g + j] : 0.0f;
            }
        }
    }
    for (int r = 0; r < 16; ++r) st[r] = st[r] * (dpt[r] - cx.dlt);  // dS^T (unscaled)
    bf16x8 dsb[2];
    for (int t = 0; t < 2; ++t) dsb[t] = cvt_frag(st, 8 * t);
    bf16x8 ft[kRing];
    auto load_tr = [&](int h) { ft[h % kRing] = read_tr_frag_x(lo0, up0, h & 3, KB + 16 * (h >> 2) * kRowBytes); };
    prio_hi();
#pragma unroll
    for (int h = 0; h < kRing - 1; ++h) load_tr(h);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        if (h + kRing - 1 < 8) load_tr(h + kRing - 1);
        sched_fence();
        acc[h & 3] = mfma_32x32x16(ft[h % kRing], dsb[h >> 2], acc[h & 3]);
        sched_fence();
    }
    prio_lo();
}

LWM_KERNEL(kDqThreads) void attn_bwd_dq_kernel(AttnParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;

    const int nqt = (p.Sq + kDqBQ - 1) / kDqBQ;
    const int HB = p.H * p.B;
    int lin = block_idx_x(), qt, hb;
    if ((HB & 7) == 0) {
        int xcd = lin & 7, i = lin >> 3;
        hb = xcd + 8 * (i / nqt);
        qt = nqt - 1 - (i % nqt);
    } else {
        hb = lin / nqt;
        qt = nqt - 1 - (lin % nqt);
    }
    const int b = hb / p.H, h = hb % p.H;

    const bf16_t* qb = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* kb = p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const bf16_t* vb = p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const bf16_t* dob = p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;

    const int q_row = qt * kDqBQ + wave * 32 + l31;
    const bool q_ok = q_row < p.Sq;
    bf16x8 qf[8], dof[8];
    for (int s = 0; s < 8; ++s) {
        if (q_ok) {
            qf[s] = __builtin_bit_cast(
                bf16x8, global_load_b128(qb + (int64_t)q_row * p.q_ss + 16 * s + 8 * hi));
            dof[s] = __builtin_bit_cast(
                bf16x8, global_load_b128(dob + (int64_t)q_row * p.do_ss + 16 * s + 8 * hi));
        } else {
            qf[s] = zero_bf16x8();
            dof[s] = zero_bf16x8();
        }
    }
    DqCtx cx;
    cx.tid = tid;
    cx.hi = hi;
    cx.ka = frag_rows_addr(lds, 0, l31, hi);
    cx.kta = frag_tr_addr(lds, lane);
    cx.stage_w = lds + tile_off(tid >> 4, tid & 15);
    cx.kseg_w = lds + 4 * kDqTileBytes + tid * 4;
    cx.kseg_r = lds + 4 * kDqTileBytes + 16 * hi;
    cx.q_pos = p.q_start + q_row;
    const int64_t stat_idx = ((int64_t)b * p.H + h) * p.Sq + q_row;
    cx.lse2 = INFINITY;
    cx.dlt = 0.0f;
    if (q_ok) {
        float l = p.lse[stat_idx];
        cx.lse2 = (l == -INFINITY) ? INFINITY : l * kLog2e;
        cx.dlt = p.delta[stat_idx];
    }
    cx.seg_q = (q_ok && p.seg_q) ? p.seg_q[(int64_t)b * p.Sq + q_row] : 0;
    cx.has_kmeta = (p.seg_k != nullptr) || (p.key_valid != nullptr) || (p.Sk % kDqBK != 0);
    cx.wq_min = p.q_start + qt * kDqBQ + wave * 32;
    cx.wq_max = cx.wq_min + 31;
    cx.c = p.scale * kLog2e;

    const int nkt_all = (p.Sk + kDqBK - 1) / kDqBK;
    int nkt = nkt_all;
    const int q_last = (qt * kDqBQ + kDqBQ < p.Sq ? qt * kDqBQ + kDqBQ : p.Sq) - 1;
    if (p.causal) {
        int64_t d = p.q_start + q_last - p.k_start;
        if (d < 0) nkt = 0;
        else {
            int64_t t = d / kDqBK + 1;
            nkt = t < nkt_all ? (int)t : nkt_all;
        }
    }

    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = zero_f32x16();

    int kt0 = 0;
    if (p.segb_q && p.segb_k && nkt > 0) {   // packed sequences: skip other documents' key tiles
        const int nbq = (p.Sq + 31) >> 5, nbk = (p.Sk + 31) >> 5;
        int smin, smax, lo, hi;
        seg_own_range(p.segb_q + (int64_t)b * nbq * 2, nbq, qt * (kDqBQ / 32), kDqBQ / 32, smin, smax);
        seg_narrow<kDqThreads>(p.segb_k + (int64_t)b * nbk * 2, nbk, kDqBK / 32, 0, nkt, smin, smax,
                               lds + 4 * kDqTileBytes, tid, lo, hi);
        kt0 = lo;
        nkt = hi;
    }
    // (pipeline under `kt0 < nkt`: see attn_fwd_kernel)
    if (kt0 < nkt) {
        DqStage stg;
        dq_stage_load(p, kb, vb, b, kt0, tid, stg);
        dq_stage_write<0>(cx, stg, kt0, p.Sk);
        block_sync();
        for (int kt = kt0; kt < nkt; kt += 2) {
            const bool more1 = kt + 1 < nkt;
            if (more1) dq_stage_load(p, kb, vb, b, kt + 1, tid, stg);
            dq_tile<0>(p, cx, qf, dof, kt, acc);
            if (more1) dq_stage_write<1>(cx, stg, kt + 1, p.Sk);
            block_sync();
            if (!more1) break;
            const bool more2 = kt + 2 < nkt;
            if (more2) dq_stage_load(p, kb, vb, b, kt + 2, tid, stg);
            dq_tile<1>(p, cx, qf, dof, kt + 1, acc);
            if (more2) dq_stage_write<0>(cx, stg, kt + 2, p.Sk);
            block_sync();
        }
    }

    if (q_ok) {
        const int64_t orow = (int64_t)b * p.dq_sb + (int64_t)q_row * p.dq_ss + (int64_t)h * p.dq_sh;
        const int64_t arow = (int64_t)b * p.dqa_sb + (int64_t)q_row * p.dqa_ss + (int64_t)h * p.dqa_sh;
        for (int db = 0; db < 4; ++db)
            for (int rq = 0; rq < 4; ++rq) {
                int d0 = 32 * db + 8 * rq + 4 * hi;
                float o0 = acc[db][4 * rq + 0] * p.scale, o1 = acc[db][4 * rq + 1] * p.scale;
                float o2 = acc[db][4 * rq + 2] * p.scale, o3 = acc[db][4 * rq + 3] * p.scale;
                if (p.carry_in) {
                    const float* a = p.dq_acc + arow + d0;
                    o0 += a[0]; o1 += a[1]; o2 += a[2]; o3 += a[3];
                }
                if (p.final_out) {
                    u32x2 pk = {pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)};
                    global_store_b64(p.dq + orow + d0, pk);
                } else {
                    u32x4 pk = {__builtin_bit_cast(uint32_t, o0), __builtin_bit_cast(uint32_t, o1),
                                __builtin_bit_cast(uint32_t, o2), __builtin_bit_cast(uint32_t, o3)};
                    global_store_b128(p.dq_acc + arow + d0, pk);
                }
            }
    }
}

// ------------------------------------------------------------------ dK, dV
// Workgroup = NW waves; wave w owns NKB blocks of 32 keys (keys
// [w*NKB*32, (w+1)*NKB*32) of the workgroup's 256-key block) and keeps their
// K fragments and f32 dK^T/dV^T accumulators in registers for the whole launch.
// Instantiated as <NW=8, NKB=1>: 2 waves/SIMD, 256 registers each.
// LDS map: V (resident, 64 KiB) | Q tile 0 | Q tile 1 | dO tile 0 | dO tile 1
//          (8 KiB each) | stats 0 | stats 1 (lse2[32], delta[32], seg_q[32])
constexpr int kDkvBK = 256;   // keys per workgroup
constexpr int kDkvBQ = 32;    // queries per LDS tile
constexpr int kDkvQTileBytes = kDkvBQ * kRowBytes;  // 8 KiB
constexpr int kDkvVBytes = kDkvBK * kRowBytes;      // 64 KiB
constexpr int kDkvStatBytes = 3 * kDkvBQ * 4;
constexpr int kDkvLdsBytes = kDkvVBytes + 4 * kDkvQTileBytes + 2 * kDkvStatBytes;

struct DkvStage {  // only the row statistics travel through registers
    float lse2, delta;
    int32_t segq;
};

template <int NKB>
struct DkvCtx {
    RowFragAddr qa;   // Q row fragments (tile 0); dO tile 0 is +2*kDkvQTileBytes
    RowFragAddr va;   // this wave's first 32 rows of the resident V tile
    TrFragAddr qta;   // Q transposed fragments (tile 0); dO likewise +2 tiles
    lds_t qtiles, stat_w, stat_r;
    int tid, hi, wave, lane_row, lane_slot;
    int64_t k_pos[NKB], wk_min, wk_max;
    int32_t kseg[NKB];
    bool has_meta;
    float c;
};

// Next Q/dO tile: global -> LDS directly (no VGPRs, no ds_write).  A wave
// instruction fills 4 tile rows (1 KiB, lane-linear), so the XOR swizzle is
// applied to the SOURCE column: lane l writes physical slot l&15 of row
// 4*piece + (l>>4) and therefore fetches logical slot (l&15) ^ swz(row).  Each
// instruction still covers 4 whole 256-B rows of global memory.
template <int NW, int NKB, int BUF>
LWM_DEVICE void dkv_stage_issue(const AttnParams& p, const DkvCtx<NKB>& cx, const bf16_t* qb,
                                const bf16_t* dob, int b, int h, int qt, DkvStage& st) {
    if (cx.tid < kDkvBQ) {
        int qr = qt * kDkvBQ + cx.tid;
        int qc = qr < p.Sq ? qr : p.Sq - 1;
        int64_t idx = ((int64_t)b * p.H + h) * p.Sq + qc;
        st.lse2 = p.lse[idx];
        st.delta = p.delta[idx];
        st.segq = p.seg_q ? p.seg_q[(int64_t)b * p.Sq + qc] : 0;
    }
    for (int j = 0; j < 8 / NW; ++j) {
        const int piece = cx.wave + NW * j;
        const int r = 4 * piece + cx.lane_row;
        int qrow = qt * kDkvBQ + r;
        // rows past Sq re-read the last row; their lse2 is +inf so p = 0
        qrow = qrow < p.Sq ? qrow : p.Sq - 1;
        const int col = ((cx.lane_slot ^ swz(r)) << 3);
        glds_load_b128(qb + (int64_t)qrow * p.q_ss + col, cx.qtiles + BUF * kDkvQTileBytes + piece * 1024);
        glds_load_b128(dob + (int64_t)qrow * p.do_ss + col,
                       cx.qtiles + (2 + BUF) * kDkvQTileBytes + piece * 1024);
    }
}

template <int NKB, int BUF>
LWM_DEVICE void dkv_stage_finish(const DkvCtx<NKB>& cx, const DkvStage& st, int qt, int Sq) {
    if (cx.tid < kDkvBQ) {
        const bool ok = (qt * kDkvBQ + cx.tid < Sq) && st.lse2 != -INFINITY;
        lds_write_f32(cx.stat_w + BUF * kDkvStatBytes, ok ? st.lse2 * kLog2e : INFINITY);
        lds_write_f32(cx.stat_w + BUF * kDkvStatBytes + kDkvBQ * 4, st.delta);
        lds_write_i32(cx.stat_w + BUF * kDkvStatBytes + 2 * kDkvBQ * 4, st.segq);
    }
}

template <int NKB, int BUF>
LWM_DEVICE void dkv_tile(const AttnParams& p, const DkvCtx<NKB>& cx, const bf16x8 (&kf)[NKB][8],
                         int qt, f32x16 (&dk)[NKB][4], f32x16 (&dv)[NKB][4], ProfAcc& pa) {
    (void)pa;
    PROF_DECL(4);
    PROF_T(0);
    const int64_t q_pos0 = p.q_start + (int64_t)qt * kDkvBQ;
    if (p.causal && q_pos0 + kDkvBQ - 1 < cx.wk_min) return;  // all queries before this wave's keys
    constexpr uint32_t QB = BUF * kDkvQTileBytes;
    constexpr uint32_t DB = (2 + BUF) * kDkvQTileBytes;
    constexpr uint32_t SB = BUF * kDkvStatBytes;

    // S = Q K^T and dP = dO V^T  (rows = queries, cols = keys)
    f32x16 s[NKB], dp[NKB];
    for (int kb = 0; kb < NKB; ++kb) {
        s[kb] = zero_f32x16();
        dp[kb] = zero_f32x16();
    }
    // per-tile opaque copies of the fragment bases: the XOR-derived addresses are
    // recomputed here instead of living in 24 registers across the whole launch
    const uint32_t qa0 = opaque(cx.qa.a[0]), va0 = opaque(cx.va.a[0]);
    const uint32_t lo0 = opaque(cx.qta.lo[0]), up0 = opaque(cx.qta.up[0]);
    // S and dP: 16 steps (8 k-steps each); operand fragments go through a register
    // ring and are requested kRing1-1 steps ahead, pinned by sched_fence (hipcc sinks
    // every ds_read next to its use otherwise and the wave eats the LDS latency of
    // each step).  pb/dsb are dead here, which pays for the ring.
    constexpr int kRing1 = 3;
    bf16x8 fa[kRing1], fv[kRing1][NKB];
    auto load1 = [&](int g) {
        if (g < 8) {
            fa[g % kRing1] = lds_read_b128(row_frag_at(qa0, g) + QB);
        } else {
            fa[g % kRing1] = lds_read_b128(row_frag_at(qa0, g - 8) + DB);
            for (int kb = 0; kb < NKB; ++kb)
                fv[g % kRing1][kb] = lds_read_b128(row_frag_at(va0, g - 8) + kb * 32 * kRowBytes);
        }
    };
    prio_hi();
#pragma unroll
    for (int g = 0; g < kRing1 - 1; ++g) load1(g);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g + kRing1 - 1 < 16) load1(g + kRing1 - 1);
        sched_fence();
        if (g < 8)
            for (int kb = 0; kb < NKB; ++kb) s[kb] = mfma_32x32x16(fa[g % kRing1], kf[kb][g], s[kb]);
        else
            for (int kb = 0; kb < NKB; ++kb) dp[kb] = mfma_32x32x16(fa[g % kRing1], fv[g % kRing1][kb], dp[kb]);
        sched_fence();
    }
    prio_lo();
    PROF_KEEP(dp[0][15]);
    PROF_T(1);
    const bool need_mask = cx.has_meta || (p.causal && q_pos0 < cx.wk_max);
    // row statistics of the 16 query rows this lane's C/D registers hold
    for (int g = 0; g < 4; ++g) {
        f32x4 l2 = lds_read_f32x4(cx.stat_r + SB + 8 * g * 4);
        for (int kb = 0; kb < NKB; ++kb)
            for (int j = 0; j < 4; ++j)
                s[kb][4 * g + j] = fast_exp2(fmaf(s[kb][4 * g + j], cx.c, -l2[j]));
    }
    if (need_mask) {
        for (int g = 0; g < 4; ++g) {
            const int ql0 = 8 * g + 4 * cx.hi;
            u32x4 sg = lds_read_u32x4(cx.stat_r + SB + 2 * kDkvBQ * 4 + 8 * g * 4);
            for (int kb = 0; kb < NKB; ++kb) {
                // query row ql sees this lane's key iff ql >= rel  (causal)
                int64_t rel64 = p.causal ? (cx.k_pos[kb] - q_pos0) : (int64_t)-1;
                const int rel = rel64 > kDkvBQ ? kDkvBQ : (rel64 < -1 ? -1 : (int)rel64);
                for (int j = 0; j < 4; ++j) {
                    bool vis = ((int32_t)sg[j] == cx.kseg[kb]) && (ql0 + j >= rel);
                    s[kb][4 * g + j] = vis ? s[kb][4 * g + j] : 0.0f;
                }
            }
        }
    }
    for (int g = 0; g < 4; ++g) {
        f32x4 dl = lds_read_f32x4(cx.stat_r + SB + kDkvBQ * 4 + 8 * g * 4);
        for (int kb = 0; kb < NKB; ++kb)
            for (int j = 0; j < 4; ++j)
                dp[kb][4 * g + j] = s[kb][4 * g + j] * (dp[kb][4 * g + j] - dl[j]);
    }
    bf16x8 pb[NKB][2], dsb[NKB][2];
    for (int kb = 0; kb < NKB; ++kb)
        for (int t = 0; t < 2; ++t) {
            pb[kb][t] = cvt_frag(s[kb], 8 * t);
            dsb[kb][t] = cvt_frag(dp[kb], 8 * t);
        }
    PROF_KEEP(dsb[0][1]);
    PROF_T(2);
    // dV += P^T dO, dK += dS^T Q: 16 steps, each one transposed fragment (2 LDS
    // reads) and NKB MFMAs.  The fragments go through a ring of kRing registers and
    // are requested kRing-1 steps ahead (s and dp are dead here, so the ring is free).
    constexpr int kRing = 3;
    bf16x8 ft[kRing];
    auto load_tr = [&](int h) {
        const int t = (h & 7) >> 2, db = h & 3;
        ft[h % kRing] = read_tr_frag_x(lo0, up0, db, (h < 8 ? DB : QB) + 16 * t * kRowBytes);
    };
    prio_hi();
#pragma unroll
    for (int h = 0; h < kRing - 1; ++h) load_tr(h);
#pragma unroll
    for (int h = 0; h < 16; ++h) {
        if (h + kRing - 1 < 16) load_tr(h + kRing - 1);
        sched_fence();   // hipcc otherwise sinks the request next to its use (one live fragment)
        const int t = (h & 7) >> 2, db = h & 3;
        if (h < 8)
            for (int kb = 0; kb < NKB; ++kb) dv[kb][db] = mfma_32x32x16(ft[h % kRing], pb[kb][t], dv[kb][db]);
        else
            for (int kb = 0; kb < NKB; ++kb) dk[kb][db] = mfma_32x32x16(ft[h % kRing], dsb[kb][t], dk[kb][db]);
        sched_fence();
    }
    prio_lo();
    PROF_KEEP(dk[0][3][0]);
    PROF_T(3);
    PROF_ADD(pa, 0, 0, 1);   // S, dP MFMAs
    PROF_ADD(pa, 1, 1, 2);   // exp / mask / dS
    PROF_ADD(pa, 2, 2, 3);   // dV, dK MFMAs
#ifdef LWM_PROF
    pa.v[5] += 1;
#endif
}

template <int NW, int NKB>
LWM_DEVICE void attn_bwd_dkdv_body(const AttnParams& p) {
    static_assert(NW * NKB * 32 == kDkvBK, "workgroup covers 256 keys");
    constexpr int NT = NW * 64;
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;

    const int nkb = (p.Sk + kDkvBK - 1) / kDkvBK;
    const int HB = p.H * p.B;
    int lin = block_idx_x(), kbi, hb;
    if ((HB & 7) == 0) {
        int xcd = lin & 7, i = lin >> 3;
        hb = xcd + 8 * (i / nkb);
        kbi = i % nkb;  // causal: low key blocks have the longest q loops -> first
    } else {
        hb = lin / nkb;
        kbi = lin % nkb;
    }
    const int b = hb / p.H, h = hb % p.H;

    const bf16_t* qb = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* kb = p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const bf16_t* vb = p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const bf16_t* dob = p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;

    DkvCtx<NKB> cx;
    cx.tid = tid;
    cx.hi = hi;
    // ---- this lane's keys
    const int wave_row0 = wave * NKB * 32;
    bf16x8 kf[NKB][8];
#pragma unroll
    for (int kbk = 0; kbk < NKB; ++kbk) {
        const int k_row = kbi * kDkvBK + wave_row0 + 32 * kbk + l31;
        const bool k_ok = k_row < p.Sk;
        for (int s = 0; s < 8; ++s) {
            if (k_ok)
                kf[kbk][s] = __builtin_bit_cast(
                    bf16x8, global_load_b128(kb + (int64_t)k_row * p.k_ss + 16 * s + 8 * hi));
            else
                kf[kbk][s] = zero_bf16x8();
        }
        cx.k_pos[kbk] = p.k_start + k_row;
        cx.kseg[kbk] = kSegInvalid;
        if (k_ok) {
            bool valid = p.key_valid ? (p.key_valid[(int64_t)b * p.Sk + k_row] != 0) : true;
            if (valid) cx.kseg[kbk] = p.seg_k ? p.seg_k[(int64_t)b * p.Sk + k_row] : 0;
        }
    }
    const lds_t qtiles = lds + kDkvVBytes;
    const lds_t stats = qtiles + 4 * kDkvQTileBytes;
    cx.qa = frag_rows_addr(qtiles, 0, l31, hi);
    cx.va = frag_rows_addr(lds + wave_row0 * kRowBytes, 0, l31, hi);
    cx.qta = frag_tr_addr(qtiles, lane);
    cx.qtiles = qtiles;
    cx.wave = wave_uniform(wave);
    cx.lane_row = lane >> 4;
    cx.lane_slot = lane & 15;
    cx.stat_w = stats + tid * 4;
    cx.stat_r = stats + 16 * hi;
    cx.has_meta =
        (p.seg_k != nullptr) || (p.key_valid != nullptr) || (kbi * kDkvBK + kDkvBK > p.Sk);
    cx.wk_min = p.k_start + (int64_t)kbi * kDkvBK + wave_row0;
    cx.wk_max = cx.wk_min + NKB * 32 - 1;
    cx.c = p.scale * kLog2e;

    // ---- resident V tile (this workgroup's 256 keys)
    for (int i = 0; i < 4096 / NT; ++i) {
        int cidx = tid + NT * i;
        int row = cidx >> 4, slot = cidx & 15;
        int kr = kbi * kDkvBK + row;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (kr < p.Sk) val = global_load_b128(vb + (int64_t)kr * p.v_ss + slot * 8);
        lds_write_b128(lds + tile_off(row, slot), val);
    }

    // ---- q tile range (causal: skip q tiles wholly before this key block)
    int nqt = (p.Sq + kDkvBQ - 1) / kDkvBQ;
    int qt0 = 0;
    if (p.causal) {
        int64_t d = p.k_start + (int64_t)kbi * kDkvBK - p.q_start;  // first q row that can see key 0
        if (d > 0) qt0 = (int)(d / kDkvBQ < nqt ? d / kDkvBQ : nqt);
    }
    if (p.segb_q && p.segb_k && qt0 < nqt) {   // packed sequences: skip other documents' query tiles
        const int nbq = (p.Sq + 31) >> 5, nbk = (p.Sk + 31) >> 5;
        int smin, smax, lo, hi;
        seg_own_range(p.segb_k + (int64_t)b * nbk * 2, nbk, kbi * (kDkvBK / 32), kDkvBK / 32, smin, smax);
        seg_narrow<NT>(p.segb_q + (int64_t)b * nbq * 2, nbq, kDkvBQ / 32, qt0, nqt, smin, smax, stats, tid, lo, hi);
        qt0 = lo;
        nqt = hi;
    }

    f32x16 dk[NKB][4], dv[NKB][4];
    for (int kbk = 0; kbk < NKB; ++kbk)
        for (int i = 0; i < 4; ++i) {
            dk[kbk][i] = zero_f32x16();
            dv[kbk][i] = zero_f32x16();
        }

    // (pipeline under `qt0 < nqt`: see attn_fwd_kernel)
    ProfAcc pa = {};
    PROF_DECL(3);
#ifdef LWM_PROF
    const unsigned long long prof_k0 = __builtin_amdgcn_s_memtime();
#endif
    // Query tiles are walked from the LAST one down to the diagonal: the workgroups that are
    // resident on an XCD at the same time (consecutive key blocks of one head) then read the same
    // Q / dO tile at about the same time and share it in that XCD's L2.  Walking up from each
    // block's own diagonal put the 32 streams at 32 different places (measured: 20.6 GB fetched
    // per launch for 0.5 GB of Q + dO per head set).
    const int q_top = nqt - 1 + qt0;             // loop index i  ->  tile q_top - i
#define LWM_QT(i) (q_top - (i))
    if (qt0 < nqt) {
        DkvStage stg;
        dkv_stage_issue<NW, NKB, 0>(p, cx, qb, dob, b, h, LWM_QT(qt0), stg);
        dkv_stage_finish<NKB, 0>(cx, stg, LWM_QT(qt0), p.Sq);
        glds_wait_all();
        block_sync();
        for (int qt = qt0; qt < nqt; qt += 2) {
            const bool more1 = qt + 1 < nqt;
            if (more1) dkv_stage_issue<NW, NKB, 1>(p, cx, qb, dob, b, h, LWM_QT(qt + 1), stg);
            dkv_tile<NKB, 0>(p, cx, kf, LWM_QT(qt), dk, dv, pa);
            PROF_T(0);
            if (more1) dkv_stage_finish<NKB, 1>(cx, stg, LWM_QT(qt + 1), p.Sq);
            glds_wait_all();
            PROF_T(1);
            block_sync();
            PROF_T(2);
            PROF_ADD(pa, 3, 0, 1);   // statistics write + DMA wait (every second tile is sampled)
            PROF_ADD(pa, 4, 1, 2);   // barrier wait
            if (!more1) break;
            const bool more2 = qt + 2 < nqt;
            if (more2) dkv_stage_issue<NW, NKB, 0>(p, cx, qb, dob, b, h, LWM_QT(qt + 2), stg);
            dkv_tile<NKB, 1>(p, cx, kf, LWM_QT(qt + 1), dk, dv, pa);
            if (more2) dkv_stage_finish<NKB, 0>(cx, stg, LWM_QT(qt + 2), p.Sq);
            glds_wait_all();
            block_sync();
        }
    }

#ifdef LWM_PROF
    if (hb == 0 && kbi == 0 && lane == 0 && p.out_acc) {   // the longest key block of head 0
        pa.v[6] = __builtin_amdgcn_s_memtime() - prof_k0;
        *((ProfAcc*)p.out_acc + wave) = pa;
    }
#endif
#pragma unroll
    for (int kbk = 0; kbk < NKB; ++kbk) {
        const int k_row = kbi * kDkvBK + wave_row0 + 32 * kbk + l31;
        if (k_row < p.Sk) {
        const int64_t krow_o = (int64_t)b * p.dk_sb + (int64_t)k_row * p.dk_ss + (int64_t)h * p.dk_sh;
        const int64_t vrow_o = (int64_t)b * p.dv_sb + (int64_t)k_row * p.dv_ss + (int64_t)h * p.dv_sh;
        const int64_t arow = (((int64_t)b * p.Sk + k_row) * p.H + h) * kHeadDim;
        for (int db = 0; db < 4; ++db)
            for (int rq = 0; rq < 4; ++rq) {
                int d0 = 32 * db + 8 * rq + 4 * hi;
                float k0 = dk[kbk][db][4 * rq + 0] * p.scale, k1 = dk[kbk][db][4 * rq + 1] * p.scale;
                float k2 = dk[kbk][db][4 * rq + 2] * p.scale, k3 = dk[kbk][db][4 * rq + 3] * p.scale;
                float v0 = dv[kbk][db][4 * rq + 0], v1 = dv[kbk][db][4 * rq + 1];
                float v2 = dv[kbk][db][4 * rq + 2], v3 = dv[kbk][db][4 * rq + 3];
                if (p.carry_in) {
                    const float* ka = p.dk_acc + arow + d0;
                    const float* va = p.dv_acc + arow + d0;
                    k0 += ka[0]; k1 += ka[1]; k2 += ka[2]; k3 += ka[3];
                    v0 += va[0]; v1 += va[1]; v2 += va[2]; v3 += va[3];
                }
                if (p.final_out) {
                    u32x2 pk = {pack_bf16x2(k0, k1), pack_bf16x2(k2, k3)};
                    u32x2 pv = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                    global_store_b64(p.dk + krow_o + d0, pk);
                    global_store_b64(p.dv + vrow_o + d0, pv);
                } else {
                    u32x4 pk = {__builtin_bit_cast(uint32_t, k0), __builtin_bit_cast(uint32_t, k1),
                                __builtin_bit_cast(uint32_t, k2), __builtin_bit_cast(uint32_t, k3)};
                    u32x4 pv = {__builtin_bit_cast(uint32_t, v0), __builtin_bit_cast(uint32_t, v1),
                                __builtin_bit_cast(uint32_t, v2), __builtin_bit_cast(uint32_t, v3)};
                    global_store_b128(p.dk_acc + arow + d0, pk);
                    global_store_b128(p.dv_acc + arow + d0, pv);
                }
            }
        }
    }
}

LWM_KERNEL(512) void attn_bwd_dkdv_kernel_w8(AttnParams p) { attn_bwd_dkdv_body<8, 1>(p); }

}  // namespace lwm
