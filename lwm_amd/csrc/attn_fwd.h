// attn_fwd.h -- blockwise attention forward for one (q block, kv block) ring
// step on gfx950.  Requires wave_ops.h + attn_common.h.
//
// Replaces the forward of `ringattention` as called at lwm/llama.py:539-569
// (blockwise online-softmax update, fp32 logits, causal_block_size=1, masks per
// lwm/llama.py:527-537 and :572-592).  One launch = the reference's whole
// "scan over q chunks x scan over k chunks" for one ring step; chunk sizes are
// an implementation detail here (256-query workgroup tile, 64-key LDS tile).
//
// Workgroup = 512 threads = 8 waves; wave w owns 32 query rows.  Scores are
// computed TRANSPOSED, S^T = K Q^T, so a lane owns ONE query column of the
// 32x32 C/D fragment: the row max / row sum of the online softmax are in-lane
// reductions plus one half-wave exchange, and exp(S^T) converted to bf16 is
// already the B operand of O^T += V^T P^T.
//
// LDS map (bytes from the dynamic base):
//   [0,16K) K tile 0 | [16K,32K) K tile 1 | [32K,48K) V tile 0 | [48K,64K) V tile 1
//   [64K, 64K+256) key meta 0 | [64K+256, 64K+512) key meta 1
#pragma once

namespace lwm {

constexpr int kFwdBQ = 256;    // queries per workgroup
constexpr int kFwdBK = 64;     // keys per LDS tile
constexpr int kFwdThreads = 512;
constexpr float kDeferLog2 = 8.0f;   // p stays below 2^8 between rescales of the running maximum
constexpr int kFwdTileBytes = kFwdBK * kRowBytes;                  // 16 KiB
constexpr int kFwdLdsBytes = 4 * kFwdTileBytes + 2 * kFwdBK * 4;  // K,V x2 + kseg x2

struct FwdStage {
    u32x4 k[2];
    u32x4 v[2];
    int32_t kseg;
    uint8_t kvalid;
};

struct FwdCtx {
    // per-lane constants of the tile loop
    RowFragAddr ka;     // K row fragments (tile 0)
    TrFragAddr va;      // V transposed fragments (tile 0)
    lds_t stage_w;      // this thread's staging slot in K tile 0
    lds_t kseg_w;       // this thread's key-meta slot (buffer 0), tid < 64
    lds_t kseg_r;       // key-meta read base (buffer 0) + 16*hi
    int tid, hi;
    int64_t q_pos, wq_min, wq_max;
    int32_t seg_q;
    bool has_kmeta;
    bool wave_idle;             // none of this wave's 32 query rows exists (q tail / decode)
    const uint8_t* mask_row;    // this lane's row of the dense mask, or null
    float c;
};

// Global loads of the next tile.  Key-meta loads go FIRST and are independent of
// each other: the tile's s_waitcnt for them then leaves the four K/V loads in
// flight (a dependent or later-issued meta load forces vmcnt(0) = a full HBM
// round trip in front of the first MFMA of every tile).
LWM_DEVICE void fwd_stage_load(const AttnParams& p, const bf16_t* kb, const bf16_t* vb,
                               int b, int kt, int tid, FwdStage& st) {
    if (tid < kFwdBK) {
        int krow = kt * kFwdBK + tid;
        int kr = krow < p.Sk ? krow : p.Sk - 1;
        st.kvalid = p.key_valid ? p.key_valid[(int64_t)b * p.Sk + kr] : (uint8_t)1;
        st.kseg = p.seg_k ? p.seg_k[(int64_t)b * p.Sk + kr] : 0;
    }
    for (int i = 0; i < 2; ++i) {
        int c = tid + kFwdThreads * i;
        int row = c >> 4, slot = c & 15;
        int krow = kt * kFwdBK + row;
        // rows past Sk re-read the last row: their keys are masked via key meta
        int kr = krow < p.Sk ? krow : p.Sk - 1;
        st.k[i] = global_load_b128(kb + (int64_t)kr * p.k_ss + slot * 8);
        st.v[i] = global_load_b128(vb + (int64_t)kr * p.v_ss + slot * 8);
    }
}

template <int BUF>
LWM_DEVICE void fwd_stage_write(const FwdCtx& cx, const FwdStage& st, int kt, int Sk) {
    // thread's chunk i lives 32 rows (8 KiB) below chunk 0: same swizzle
    for (int i = 0; i < 2; ++i) {
        lds_write_b128(cx.stage_w + BUF * kFwdTileBytes + i * 32 * kRowBytes, st.k[i]);
        lds_write_b128(cx.stage_w + (2 + BUF) * kFwdTileBytes + i * 32 * kRowBytes, st.v[i]);
    }
    if (cx.tid < kFwdBK) {
        const bool ok = (kt * kFwdBK + cx.tid < Sk) && st.kvalid != 0;
        lds_write_i32(cx.kseg_w + BUF * kFwdBK * 4, ok ? st.kseg : kSegInvalid);
    }
}

// ---- the three phases of one 64-key tile against this wave's 32 queries.
// INFER = the dense-mask / split-K flavour (ringattention_inference); the training
// kernel is compiled without that code.

// S^T = K Q^T  (rows = keys, cols = queries); K tile in LDS buffer BUF.
// Fragment bases are re-derived per tile (XOR form, attn_common.h) and the K / V
// fragments are requested kRing-1 steps ahead through a register ring, pinned by
// sched_fence: hipcc otherwise sinks each ds_read next to its MFMA and the wave
// pays the LDS latency at every step.
template <int BUF, int kRing = 4>
LWM_DEVICE void fwd_phase_s(const FwdCtx& cx, const bf16x8 (&qf)[8], f32x16 (&st)[2]) {
    constexpr uint32_t KB = BUF * kFwdTileBytes;
    const uint32_t ka0 = opaque(cx.ka.a[0]);
    st[0] = zero_f32x16();
    st[1] = zero_f32x16();
    bf16x8 fa[kRing];
    // step g = (d slice s = g >> 1, key half kb2 = g & 1): consecutive MFMAs alternate between the
    // two accumulators, so that the ds_read issued between them never sits between two MFMAs on
    // the SAME accumulator (a dependent pair must be adjacent or it pays ~40 cycles; with one
    // matrix-phase wave per SIMD nothing hides that).
    auto load1 = [&](int g) {
        fa[g % kRing] = lds_read_b128(row_frag_at(ka0, g >> 1) + KB + (g & 1) * 32 * kRowBytes);
    };
    prio_hi();
#pragma unroll
    for (int g = 0; g < kRing - 1; ++g) load1(g);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g + kRing - 1 < 16) load1(g + kRing - 1);
        sched_fence();
        st[g & 1] = mfma_32x32x16(fa[g % kRing], qf[g >> 1], st[g & 1]);
        sched_fence();
    }
    prio_lo();
}

// masks + online softmax of the scores in st; leaves P^T (bf16) in pb and updates m_run, l_run, acc.
// Key meta of the tile in meta buffer BUF.
template <int BUF, bool INFER>
LWM_DEVICE void fwd_phase_softmax(const AttnParams& p, const FwdCtx& cx, int kt, f32x16 (&st)[2],
                                  bf16x8 (&pb)[2][2], float& m_run, float& l_run, f32x16 (&acc)[4]) {
    const int64_t k_pos0 = p.k_start + (int64_t)kt * kFwdBK;
    // ---- masks (lwm/llama.py:572-592): causal, same segment, key valid
    const bool need_mask = cx.has_kmeta || (p.causal && k_pos0 + kFwdBK - 1 > cx.wq_min);
    if (need_mask) {
        // key kl of this tile is causally visible iff kl <= rel.  (Every loop here is fully
        // unrolled: a rolled loop indexes the score registers dynamically -- s_set_gpr_idx --
        // which made every masked tile ~4x slower than an unmasked one.)
        int64_t rel64 = p.causal ? (cx.q_pos - k_pos0) : (int64_t)kFwdBK;
        const int rel = rel64 > kFwdBK ? kFwdBK : (rel64 < -1 ? -1 : (int)rel64);
        const int relh = rel - 4 * cx.hi;   // kl = 32*kb2 + 8*g + 4*hi + j <= rel
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (cx.has_kmeta) {
                    u32x4 sg = lds_read_u32x4(cx.kseg_r + BUF * kFwdBK * 4 + (32 * kb2 + 8 * g) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bool vis = ((int32_t)sg[j] == cx.seg_q) && (32 * kb2 + 8 * g + j <= relh);
                        st[kb2][4 * g + j] = vis ? st[kb2][4 * g + j] : -INFINITY;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        st[kb2][4 * g + j] = (32 * kb2 + 8 * g + j <= relh) ? st[kb2][4 * g + j] : -INFINITY;
                }
            }
    }
    // ---- arbitrary boolean mask (ringattention_inference, lwm/llama.py:577-614)
    if (INFER && cx.mask_row) {
        const uint8_t* mr = cx.mask_row + (int64_t)kt * kFwdBK;
        for (int kb2 = 0; kb2 < 2; ++kb2)
            for (int r = 0; r < 16; ++r) {
                const int kl = 32 * kb2 + cd_row(r, cx.hi);
                const bool in = kt * kFwdBK + kl < p.Sk;
                if (!in || mr[kl] == 0) st[kb2][r] = -INFINITY;
            }
    }
    // ---- online softmax (per query column)
    float mxp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // 4 independent chains
    for (int kb2 = 0; kb2 < 2; ++kb2)
        for (int r = 0; r < 16; ++r) mxp[r & 3] = fmaxf(mxp[r & 3], st[kb2][r]);
    float mx = fmaxf(fmaxf(mxp[0], mxp[1]), fmaxf(mxp[2], mxp[3]));
    mx = fmaxf(mx, xhalf(mx));
    // Deferred rescale: the reference maximum m_run moves (and the 64 accumulator
    // registers are rescaled) only when some row of this wave would otherwise produce
    // p > 2^kDeferLog2; until then p = exp2((s - m_run)*c) is merely allowed to exceed 1.
    // The decision precedes this tile's exponentials and the previous tile's P.V is
    // complete, so everything at the old scale is rescaled exactly once.  The first
    // visible tile always takes the branch (m_run = -inf).
    if (wave_any((mx - m_run) * cx.c > kDeferLog2)) {
        const float m_new = fmaxf(m_run, mx);
        const float ms = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = fast_exp2((m_run - ms) * cx.c);
        l_run *= alpha;
        for (int i = 0; i < 4; ++i) acc[i] *= alpha;
        m_run = m_new;
    }
    const float msc = ((m_run == -INFINITY) ? 0.0f : m_run) * cx.c;
    float ps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int kb2 = 0; kb2 < 2; ++kb2)
        for (int r = 0; r < 16; ++r) {
            float pv = fast_exp2(fmaf(st[kb2][r], cx.c, -msc));
            st[kb2][r] = pv;
            ps[r & 3] += pv;
        }
    l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
    for (int kb2 = 0; kb2 < 2; ++kb2)
        for (int t = 0; t < 2; ++t) pb[kb2][t] = cvt_frag(st[kb2], 8 * t);
}

// O^T += V^T P^T; V tile in LDS buffer BUF.
template <int BUF, int kRing = 4>
LWM_DEVICE void fwd_phase_pv(const FwdCtx& cx, const bf16x8 (&pb)[2][2], f32x16 (&acc)[4]) {
    constexpr uint32_t VB = BUF * kFwdTileBytes;  // va already points at V tile 0
    const uint32_t lo0 = opaque(cx.va.lo[0]), up0 = opaque(cx.va.up[0]);
    bf16x8 ft[kRing];
    auto load_tr = [&](int h) {   // h = (kb2, t, db)
        ft[h % kRing] = read_tr_frag_x(lo0, up0, h & 3, VB + 16 * (h >> 2) * kRowBytes);
    };
    prio_hi();
#pragma unroll
    for (int h = 0; h < kRing - 1; ++h) load_tr(h);
#pragma unroll
    for (int h = 0; h < 16; ++h) {
        if (h + kRing - 1 < 16) load_tr(h + kRing - 1);
        sched_fence();
        acc[h & 3] = mfma_32x32x16(ft[h % kRing], pb[h >> 3][(h >> 2) & 1], acc[h & 3]);
        sched_fence();
    }
    prio_lo();
}

// true when the tile cannot contribute to this wave's rows (staging only)
LWM_DEVICE bool fwd_wave_skips(const AttnParams& p, const FwdCtx& cx, int kt, bool infer) {
    if (infer && cx.wave_idle) return true;           // short query blocks
    return p.causal && p.k_start + (int64_t)kt * kFwdBK > cx.wq_max;   // wholly in this wave's future
}

// One 64-key tile held in LDS buffer BUF, the three phases back to back.
template <int BUF, bool INFER>
LWM_DEVICE void fwd_tile(const AttnParams& p, const FwdCtx& cx, const bf16x8 (&qf)[8], int kt,
                         float& m_run, float& l_run, f32x16 (&acc)[4], ProfAcc& pa) {
    (void)pa;
    PROF_DECL(4);
    PROF_T(0);
    if (fwd_wave_skips(p, cx, kt, INFER)) return;
    f32x16 st[2];
    bf16x8 pb[2][2];
    fwd_phase_s<BUF>(cx, qf, st);
    PROF_KEEP(st[1][15]);
    PROF_T(1);
    fwd_phase_softmax<BUF, INFER>(p, cx, kt, st, pb, m_run, l_run, acc);
    PROF_KEEP(pb[1][1]);
    PROF_T(2);
    fwd_phase_pv<BUF>(cx, pb, acc);
    PROF_KEEP(acc[3][0]);
    PROF_T(3);
    PROF_ADD(pa, 0, 0, 1);   // S = K Q^T
    PROF_ADD(pa, 1, 1, 2);   // masks + softmax
    PROF_ADD(pa, 2, 2, 3);   // O += V^T P^T
#ifdef LWM_PROF
    pa.v[5] += 1;
#endif
}

template <bool INFER>
LWM_DEVICE void attn_fwd_body(const AttnParams& p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;

    // ---- block -> (q tile, head, batch): longest q tiles first; all q tiles of
    // one (b,h) stay on one XCD (block b runs on XCD b%8) so its K/V stream is
    // shared in that XCD's L2.
    const int nqt = (p.Sq + kFwdBQ - 1) / kFwdBQ;
    const int HB = p.H * p.B;
    const int nsplit = (INFER && p.k_splits > 1) ? p.k_splits : 1;
    const int split = INFER ? block_idx_x() / (nqt * HB) : 0;
    int lin = block_idx_x() - split * (nqt * HB), qt, hb;
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

    // ---- this lane's query row
    const int q_row = qt * kFwdBQ + wave * 32 + l31;
    const bool q_ok = q_row < p.Sq;
    bf16x8 qf[8];
    for (int s = 0; s < 8; ++s) {
        if (q_ok) {
            u32x4 raw = global_load_b128(qb + (int64_t)q_row * p.q_ss + 16 * s + 8 * hi);
            qf[s] = __builtin_bit_cast(bf16x8, raw);
        } else {
            qf[s] = zero_bf16x8();
        }
    }

    FwdCtx cx;
    cx.tid = tid;
    cx.hi = hi;
    cx.ka = frag_rows_addr(lds, 0, l31, hi);
    cx.va = frag_tr_addr(lds + 2 * kFwdTileBytes, lane);
    cx.stage_w = lds + tile_off(tid >> 4, tid & 15);
    cx.kseg_w = lds + 4 * kFwdTileBytes + tid * 4;
    cx.kseg_r = lds + 4 * kFwdTileBytes + 16 * hi;
    cx.q_pos = p.q_start + q_row;
    cx.seg_q = (q_ok && p.seg_q) ? p.seg_q[(int64_t)b * p.Sq + q_row] : 0;
    cx.has_kmeta = (p.seg_k != nullptr) || (p.key_valid != nullptr) || (p.Sk % kFwdBK != 0);
    cx.wq_min = p.q_start + qt * kFwdBQ + wave * 32;
    cx.wq_max = cx.wq_min + 31;
    cx.c = p.scale * kLog2e;
    cx.wave_idle = INFER && wave_uniform(qt * kFwdBQ + wave * 32 >= p.Sq ? 1 : 0) != 0;
    cx.mask_row = (INFER && p.dense_mask && q_ok)
                      ? p.dense_mask + (int64_t)b * p.msk_sb + (int64_t)q_row * p.msk_sq
                      : nullptr;

    // ---- kv tile range (causal: skip tiles wholly in the future of this q tile)
    const int nkt_all = (p.Sk + kFwdBK - 1) / kFwdBK;
    int nkt = nkt_all;
    const int q_last = (qt * kFwdBQ + kFwdBQ < p.Sq ? qt * kFwdBQ + kFwdBQ : p.Sq) - 1;
    if (p.causal) {
        int64_t d = p.q_start + q_last - p.k_start;  // last visible key index
        if (d < 0) nkt = 0;
        else {
            int64_t t = d / kFwdBK + 1;
            nkt = t < nkt_all ? (int)t : nkt_all;
        }
    }

    ProfAcc pa = {};
    PROF_DECL(5);
#ifdef LWM_PROF
    const unsigned long long prof_k0 = __builtin_amdgcn_s_memtime();
#endif
    float m_run = -INFINITY;  // running max of raw scores (q.k, unscaled)
    float l_run = 0.0f;       // this half-wave's partial row sum
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = zero_f32x16();

    // The whole pipeline sits under `nkt > 0` so that no CFG path reaches the
    // tile loop without passing the prologue's s_waitcnt (otherwise the Q-fragment
    // loads count as possibly pending at the loop's first MFMA and the compiler
    // drains vmcnt to 0 there every tile, serialising the staging loads).
    int kt0 = 0;
    // packed sequences: skip key tiles that belong to other documents
    if (p.segb_q && p.segb_k && nkt > 0) {
        const int nbq = (p.Sq + 31) >> 5, nbk = (p.Sk + 31) >> 5;
        int smin, smax, lo, hi;
        seg_own_range(p.segb_q + (int64_t)b * nbq * 2, nbq, qt * (kFwdBQ / 32), kFwdBQ / 32, smin, smax);
        seg_narrow<kFwdThreads>(p.segb_k + (int64_t)b * nbk * 2, nbk, kFwdBK / 32, 0, nkt, smin, smax,
                                lds + 4 * kFwdTileBytes, tid, lo, hi);
        kt0 = lo;
        nkt = hi;
    }
    // split-K: this workgroup walks tiles [kt0, nkt) of its piece only
    if (nsplit > 1) {
        const int per = (nkt_all + nsplit - 1) / nsplit;
        const int s0 = split * per, s1 = s0 + per;
        kt0 = kt0 > s0 ? kt0 : s0;
        nkt = nkt < s1 ? nkt : s1;
    }
    if (kt0 < nkt) {
        FwdStage stg;
        fwd_stage_load(p, kb, vb, b, kt0, tid, stg);
        fwd_stage_write<0>(cx, stg, kt0, p.Sk);
        block_sync();
        // two tiles per trip so the LDS buffer index is a compile-time constant
        for (int kt = kt0; kt < nkt; kt += 2) {
            const bool more1 = kt + 1 < nkt;
            if (more1) fwd_stage_load(p, kb, vb, b, kt + 1, tid, stg);
            fwd_tile<0, INFER>(p, cx, qf, kt, m_run, l_run, acc, pa);
            PROF_T(0);
            if (more1) fwd_stage_write<1>(cx, stg, kt + 1, p.Sk);
            PROF_T(1);
            block_sync();
            PROF_T(2);
            PROF_ADD(pa, 3, 0, 1);   // staging ds_writes (every second tile is sampled)
            PROF_ADD(pa, 4, 1, 2);   // barrier wait
            if (!more1) break;
            const bool more2 = kt + 2 < nkt;
            if (more2) fwd_stage_load(p, kb, vb, b, kt + 2, tid, stg);
            fwd_tile<1, INFER>(p, cx, qf, kt + 1, m_run, l_run, acc, pa);
            if (more2) fwd_stage_write<0>(cx, stg, kt + 2, p.Sk);
            block_sync();
        }
    }

#ifdef LWM_PROF
    if (!INFER && hb == 0 && qt == nqt - 1 && lane == 0 && p.out_acc) {   // the longest q tile of head 0
        pa.v[6] = __builtin_amdgcn_s_memtime() - prof_k0;
        *((ProfAcc*)p.out_acc + wave) = pa;
    }
#endif
    // ---- epilogue: normalise, merge with the ring carry, store
    const float l_tot = l_run + xhalf(l_run);
    float inv = 0.0f, lse_b = -INFINITY;
    if (l_tot > 0.0f) {
        inv = 1.0f / l_tot;
        lse_b = m_run * p.scale + logf(l_tot);
    }
    float w_a = 0.0f, w_b = 1.0f, lse_new = lse_b;
    const int64_t lse_idx = ((int64_t)b * p.H + h) * p.Sq + q_row + (int64_t)split * p.B * p.H * p.Sq;
    if (p.carry_in && q_ok) {
        float lse_a = p.lse_acc[lse_idx];
        float mx = fmaxf(lse_a, lse_b);
        if (mx == -INFINITY) {
            lse_new = -INFINITY;
            w_a = 0.0f;
            w_b = 0.0f;
        } else {
            float ea = expf(lse_a - mx), eb = expf(lse_b - mx);
            lse_new = mx + logf(ea + eb);
            w_a = ea / (ea + eb);
            w_b = eb / (ea + eb);
        }
    }
    if (q_ok) {
        const float sc = inv * w_b;
        const int64_t orow = (int64_t)b * p.o_sb + (int64_t)q_row * p.o_ss + (int64_t)h * p.o_sh;
        // the f32 carry is dense [B,Sq,H,D]
        const int64_t arow = (((int64_t)b * p.Sq + q_row) * p.H + h) * kHeadDim +
                             (int64_t)split * p.B * p.Sq * p.H * kHeadDim;
        for (int db = 0; db < 4; ++db)
            for (int rq = 0; rq < 4; ++rq) {
                int d0 = 32 * db + 8 * rq + 4 * hi;
                float o0 = acc[db][4 * rq + 0] * sc, o1 = acc[db][4 * rq + 1] * sc;
                float o2 = acc[db][4 * rq + 2] * sc, o3 = acc[db][4 * rq + 3] * sc;
                if (p.carry_in) {
                    const float* a = p.out_acc + arow + d0;
                    o0 += a[0] * w_a; o1 += a[1] * w_a; o2 += a[2] * w_a; o3 += a[3] * w_a;
                }
                if (p.final_out) {
                    u32x2 pk = {pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)};
                    global_store_b64(p.out + orow + d0, pk);
                } else {
                    u32x4 pk = {__builtin_bit_cast(uint32_t, o0), __builtin_bit_cast(uint32_t, o1),
                                __builtin_bit_cast(uint32_t, o2), __builtin_bit_cast(uint32_t, o3)};
                    global_store_b128(p.out_acc + arow + d0, pk);
                }
            }
        if (hi == 0) {
            if (p.final_out) p.lse[lse_idx] = lse_new;
            else p.lse_acc[lse_idx] = lse_new;
        }
    }
}

LWM_KERNEL(kFwdThreads) void attn_fwd_kernel(AttnParams p) { attn_fwd_body<false>(p); }
LWM_KERNEL(kFwdThreads) void attn_fwd_infer_kernel(AttnParams p) { attn_fwd_body<true>(p); }

}  // namespace lwm
