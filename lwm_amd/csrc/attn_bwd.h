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
//   attn_bwd_dq_kernel    : workgroup owns 256 queries, streams K/V      (3 GEMMs)
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
    cx.lse2 = INFINITY;
    cx.dlt = 0.0f;
    if (q_ok) {
        const int64_t Sqp = bwd_stat_pad(p.Sq), srow = bwd_stat_row((int64_t)b * p.H + h, Sqp);
        cx.lse2 = -p.delta[srow + q_row];         // (+inf for a row that saw no key: p = 0)
        cx.dlt = -p.delta[srow + Sqp + q_row];
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

}  // namespace lwm
