// attn_bwd_fused.h -- the whole backward of one (q block, kv block) ring step in ONE launch that
// computes S and dP once: 5 GEMM units (S, dP, dV, dK, dQ) instead of the 7 of the two-kernel
// backward (attn_bwd.h: dQ kernel S,dP,dQ + dK/dV kernel S,dP,dV,dK).  Requires wave_ops.h,
// attn_common.h and attn_bwd.h (tile constants).
//
// Replaces the custom-VJP backward of `ringattention` (call site lwm/llama.py:539-569; SURVEY.md
// Appendix A.1): p from the saved LSE, dv += p^T do, dp = do v^T, ds = p*(dp - rowsum(do*o)),
// dq += ds k, dk += ds^T q.
//
// Shape.  A work item = (batch*head, block of 256 keys).  The workgroup (8 waves, wave w owns keys
// 32w..32w+31) keeps dK^T/dV^T accumulators and its V fragments in registers and the K block in LDS,
// and walks the 32-query tiles from the last one down to its causal diagonal, as attn_bwd_dkdv does.
// New per tile:
//   * dS^T (bf16, [256 keys][32 queries]) is written to LDS by the waves that produced it;
//   * one step later every wave w multiplies it by K^T for its 16 head dims:
//         dQ^T[16w..16w+15][32 q] = K^T[16 d][256 keys] . dS^T[256 keys][32 q]
//     on v_mfma_f32_16x16x32_bf16 (both operands by ds_read_b64_tr_b16 from the [key][*] images),
//     i.e. the reduction over the workgroup's 256 keys happens inside the MFMA accumulator;
//   * the 32 x 128 f32 partial is ADDED to the dq accumulator in global memory.  The key blocks of
//     one head add to a query tile in a FIXED order (ascending key block) enforced by a per-tile
//     counter -- deterministic, no atomics on data.  A tile's accumulator is touched by consecutive
//     key blocks within a few microseconds of each other and stays in L2 meanwhile.
//
// Who may read what (MI355X: one L2 per XCD, L2s not coherent with each other, per-CU L1 never
// refreshed by other CUs' stores -- MI355X_MICROARCH.md "inter-workgroup visibility"):
//   * all key blocks of one (batch, head) are executed by workgroups of ONE XCD.  This is enforced,
//     not assumed: work is handed out through 8 queues (heads hb with hb % 8 == q), a queue is
//     CLAIMED (atomic CAS) by the XCC id (s_getreg HW_REG_XCC_ID) of the first workgroup that takes
//     from it, and only workgroups with that XCC id take from it afterwards.  Persistent workgroups
//     drain their own XCD's queue first and then claim whatever is unclaimed, so every item is
//     executed whatever the dispatcher does;
//   * within an XCD: a wave stores its slice of the tile plainly (L1 is write-through; the store is
//     acknowledged by L2), waits vmcnt(0) -- one step later, when it costs nothing -- and then stores the
//     slice's counter; the reader polls the counter (sc1: never from L1) and then loads the slice with PLAIN
//     loads from that same L2 -- its own L1 cannot hold the lines (invalidated once per work item; a slice
//     is whole 128-byte lines read once per item by the one wave that owns them).  An sc1 / nt load of a
//     line that is dirty in L2 is served from memory instead (measured: the whole read-modify-write stream
//     became HBM traffic, 59 GB per launch).  Ordering is per wave slice (32 head dims x 16 queries).
//   * progress: an item waits only for the item with the next-lower ticket of the same queue, which
//     was taken earlier by a workgroup that is running -- no dependence on dispatch order.
#pragma once

namespace lwm {

constexpr int kFbThreads = 512;
constexpr int kFbDsBytes = kDkvBK * 64;     // dS^T image: 256 key rows x 32 queries bf16 = 16 KiB
constexpr int kFbQueues = 8;
// LDS map: K (64 KiB) | Q tile 0,1 | dO tile 0,1 (8 KiB each) | dS^T 0,1 (16 KiB each) | stats 0,1 | ctl
constexpr int kFbOffTiles = kDkvVBytes;
constexpr int kFbOffDs = kFbOffTiles + 4 * kDkvQTileBytes;
constexpr int kFbOffStats = kFbOffDs + 2 * kFbDsBytes;
// stats block: lse2[32] | delta[32] | seg[32] | 128 B landing pad of the 64-lane segment-id DMA
constexpr int kFbStatBytes = 4 * kDkvBQ * 4;
constexpr int kFbOffCtl = kFbOffStats + 2 * kFbStatBytes;
constexpr int kFbOffPoll = kFbOffCtl + 64;          // 8 waves x 256 B: where each wave's flag poll lands
constexpr int kFbOffDq = kFbOffPoll + 8 * 256;      // 8 waves x 2 KiB: the accumulator slice each wave is about to add to
constexpr int kFbLdsBytes = kFbOffDq + 8 * 2048;

// workspace (int32): [0,8) tickets | [8,16) queue owner (0 = unclaimed, xcc+1) | [16] spin-limit flag |
// [32, 32 + B*H*nqt*8) per-(b,h,tile,wave) counters | then f32 [B,H,Sq]: LSE in log2 units (+inf where the
// row has no visible key), written by attn_bwd_lse2_kernel before the main launch
constexpr int kFbWsHeader = 32;
constexpr int kFbWsErr = 16;
// A wait that does not end within this many polls (~0.5 s; a healthy wait is a few polls) gives up and
// raises ws[kFbWsErr]: results are then wrong, but the GPU is not left spinning (every spin is bounded).
constexpr int kFbSpinLimit = 1 << 20;
// A key block starts its walk only when its predecessor is already kFbSlack tiles ahead.  Consecutive key
// blocks form a pipeline with blocking and no buffers (a block can never overtake the one before it); the
// distance between two of them is the hand-off latency (store -> flag -> poll: about two steps) plus this
// slack.  A tile of dq (16 KiB f32) and its Q/dO tiles (16 KiB) are re-used by the next block one distance
// later, with the tiles of all 32 co-resident blocks of the XCD in between: LRU distance = 32 x distance x
// 32 KiB, i.e. 2.5 MiB at distance 2.5 and 4.6 MiB at 4.5 -- against a 4 MiB L2.  rocprofv3 at S = 32768:
// slack 2 -> 36 GB fetched per launch (TCC_MISS 2.9e8), slack 0 -> 19 GB (1.5e8); time 28.3 vs 27.7 ms
// (0/1/2/3: 27.7 / 27.9 / 28.3 / 29.9; scripts/micro/l2_handoff.hip shows the same chain, alone, running
// entirely out of L2).
#ifndef LWM_FB_SLACK
#define LWM_FB_SLACK 0
#endif
constexpr int kFbSlack = LWM_FB_SLACK;

// Per-lane state that lives across the whole tile loop is kept to a minimum (the loop runs at the
// 256-register limit: 128 accumulator + 32 V-fragment registers are pinned): fragment addresses are
// re-derived from the lane id inside each phase (a dozen VALU per 4k-cycle step) instead of being held,
// and every position test is done on ONE wave-uniform 32-bit offset.
struct FusedCtx {
    lds_t lds;          // dynamic LDS base
    int lane, wave;     // wave is wave-uniform (SGPR)
    int tid;
    int32_t kseg;       // segment id of this lane's key (kSegInvalid: padded / out of range)
    bool has_meta;
    int n_dma;          // LDS-DMA instructions this wave issues per fb_stage_issue (wave-uniform)
    float c;            // scale * log2(e)
};

// dS^T image: key row r = 64 bytes = 8 chunks of 4 queries; chunk c sits at c ^ ((r >> 1) & 7) so that
// the 8-byte row writes of a wave (16 consecutive keys per LDS cycle) and the transposed reads are
// both bank-conflict free.
LWM_DEVICE uint32_t ds_off(int key, int chunk) { return (uint32_t)(key * 64 + ((chunk ^ ((key >> 1) & 7)) << 3)); }

// lse2[i] = lse[i] * log2(e), +inf for rows without a visible key (p = exp2(s*c - lse2) is then 0)
LWM_KERNEL(256) void attn_bwd_lse2_kernel(const float* lse, float* lse2, int64_t n) {
    for (int64_t i = (int64_t)block_idx_x() * 256 + thread_idx(); i < n; i += (int64_t)grid_dim_x() * 256) {
        const float l = lse[i];
        lse2[i] = (l == -INFINITY) ? INFINITY : l * kLog2e;
    }
}

// Staging of the next tile: Q and dO rows AND the row statistics go global -> LDS directly (inline-asm
// DMA, see wave_ops.h): no VGPR is involved and hipcc sees no load.  That matters beyond the registers:
// hipcc's waitcnt insertion is flow-insensitive, and a compiler-visible load that is issued or consumed
// under a predicate leaves a "maybe pending" mark that turns into an s_waitcnt vmcnt(N) at some later
// redefinition of its register -- here in front of the first MFMA of the NEXT step, where vmcnt also
// counts the just-issued DMA and the dq stores (measured: ~1 us per step).  Rows past Sq re-read the
// last row (clamped, not predicated); the ragged tile masks them (fb_tile_ab).
// Wave 0: lanes 0..31 lse2, lanes 32..63 delta; wave 1: segment ids (when given).
template <int BUF>
LWM_DEVICE void fb_stage_issue(const AttnParams& p, const FusedCtx& cx, const bf16_t* qb, const bf16_t* dob,
                               const float* lse2, int b, int h, int qt) {
    const int lane = (int)opaque((uint32_t)cx.lane);
    const int piece = cx.wave;                 // 8 waves x 1 KiB = one 8 KiB tile
    const int r = 4 * piece + (lane >> 4);
    int qrow = qt * kDkvBQ + r;
    qrow = qrow < p.Sq ? qrow : p.Sq - 1;
    const int col = (((lane & 15) ^ swz(r)) << 3);
    const lds_t qtiles = cx.lds + kFbOffTiles;
    glds_load_b128(qb + (int64_t)qrow * p.q_ss + col, qtiles + BUF * kDkvQTileBytes + piece * 1024);
    glds_load_b128(dob + (int64_t)qrow * p.do_ss + col, qtiles + (2 + BUF) * kDkvQTileBytes + piece * 1024);
    if (cx.wave < 2) {
        int qr = qt * kDkvBQ + (lane & 31);
        qr = qr < p.Sq ? qr : p.Sq - 1;
        const int64_t idx = ((int64_t)b * p.H + h) * p.Sq + qr;
        const lds_t stats = cx.lds + kFbOffStats + BUF * kFbStatBytes;
        if (cx.wave == 0) {
            glds_load_b32((lane < 32 ? lse2 : p.delta) + idx, stats);
        } else if (p.seg_q) {
            // lanes 32..63 land in the 128 bytes after the table (the other buffer's lse2 slot is NOT there:
            // the stats block is padded, see kFbStatBytes)
            glds_load_b32(p.seg_q + (int64_t)b * p.Sq + qr, stats + 2 * kDkvBQ * 4);
        }
    }
}

// S, dP, P, dS, dV, dK for one 32-query tile in buffer BUF; leaves dS^T (bf16) in LDS buffer BUF.
// krel = (global position of this WAVE's first key) - (global position of the tile's first query),
// clamped to +-64 (wave-uniform).
template <int BUF>
LWM_DEVICE void fb_tile_ab(const AttnParams& p, const FusedCtx& cx, const bf16x8 (&vf)[8], int krel, int qlim,
                           bf16x8 (&pb)[2], bf16x8 (&dsb)[2]) {
    constexpr uint32_t QB = BUF * kDkvQTileBytes;
    constexpr uint32_t DB = (2 + BUF) * kDkvQTileBytes;
    const int lane = (int)opaque((uint32_t)cx.lane);
    const int l31 = lane & 31, hi = lane >> 5;
    // this lane's dS^T row (key 32*wave + l31): row base + XOR term
    const uint32_t ds_row = cx.lds + kFbOffDs + BUF * kFbDsBytes + (uint32_t)(32 * cx.wave + l31) * 64;
    const uint32_t ds_x = (uint32_t)((l31 >> 1) & 7);
    // (no early-out for waves whose keys all lie after the tile's queries: that only happens on the 8
    // tiles of the block's own diagonal -- 28 of ~8000 wave-tiles at S = 32768 -- and a second exit
    // with LDS stores makes hipcc spill the accumulators; the mask below zeroes P and dS there)
    f32x16 s = zero_f32x16(), dp = zero_f32x16();
    const lds_t qtiles = cx.lds + kFbOffTiles;
    const uint32_t qa0 = qtiles + tile_off(l31, hi);
    const uint32_t ka0 = cx.lds + (uint32_t)cx.wave * (32 * kRowBytes) + tile_off(l31, hi);
    // S = Q K^T (K row fragments from the resident K block), dP = dO V^T (V fragments in registers):
    // 16 steps; LDS operands go through a register ring, requested kRing1-1 steps ahead and pinned by
    // sched_fence (see attn_bwd.h::dkv_tile).
    constexpr int kRing1 = 3;
    bf16x8 fa[kRing1], fk[kRing1];
    auto load1 = [&](int g) {
        if (g < 8) {
            fa[g % kRing1] = lds_read_b128(row_frag_at(qa0, g) + QB);
            fk[g % kRing1] = lds_read_b128(row_frag_at(ka0, g));
        } else {
            fa[g % kRing1] = lds_read_b128(row_frag_at(qa0, g - 8) + DB);
        }
    };
    prio_hi();
#pragma unroll
    for (int g = 0; g < kRing1 - 1; ++g) load1(g);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g + kRing1 - 1 < 16) load1(g + kRing1 - 1);
        sched_fence();
        if (g < 8) s = mfma_32x32x16(fa[g % kRing1], fk[g % kRing1], s);
        else dp = mfma_32x32x16(fa[g % kRing1], vf[g - 8], dp);
        sched_fence();
    }
    prio_lo();
    const uint32_t stat_r = cx.lds + kFbOffStats + BUF * kFbStatBytes + 16 * hi;
    // qlim = query rows of the tile that exist (32 except in a ragged last tile)
    const bool need_mask = cx.has_meta || (p.causal && krel > -(kDkvBQ - 1)) || qlim < kDkvBQ;
    for (int g = 0; g < 4; ++g) {
        f32x4 l2 = lds_read_f32x4(stat_r + 8 * g * 4);
        for (int j = 0; j < 4; ++j) s[4 * g + j] = fast_exp2(fmaf(s[4 * g + j], cx.c, -l2[j]));
    }
    if (need_mask) {
        // query row ql of the tile sees this lane's key iff ql >= rel (causal)
        int rel = p.causal ? krel + l31 : -1;
        rel = rel > kDkvBQ ? kDkvBQ : (rel < -1 ? -1 : rel);
        for (int g = 0; g < 4; ++g) {
            const int ql0 = 8 * g + 4 * hi;
            u32x4 sg = lds_read_u32x4(stat_r + 2 * kDkvBQ * 4 + 8 * g * 4);
            for (int j = 0; j < 4; ++j) {
                bool vis = ((int32_t)sg[j] == cx.kseg) && (ql0 + j >= rel) && (ql0 + j < qlim);
                s[4 * g + j] = vis ? s[4 * g + j] : 0.0f;
            }
        }
    }
    for (int g = 0; g < 4; ++g) {
        f32x4 dl = lds_read_f32x4(stat_r + kDkvBQ * 4 + 8 * g * 4);
        for (int j = 0; j < 4; ++j) dp[4 * g + j] = s[4 * g + j] * (dp[4 * g + j] - dl[j]);
    }
    for (int t = 0; t < 2; ++t) {
        pb[t] = cvt_frag(s, 8 * t);
        dsb[t] = cvt_frag(dp, 8 * t);
    }
    // dS^T -> LDS: this lane's key row, queries 8g + 4hi .. +3 = chunk 2g + hi
    {
        const u32x4 d0 = __builtin_bit_cast(u32x4, dsb[0]), d1 = __builtin_bit_cast(u32x4, dsb[1]);
        lds_write_b64(ds_row + (((0 + hi) ^ ds_x) << 3), u32x2{d0[0], d0[1]});
        lds_write_b64(ds_row + (((2 + hi) ^ ds_x) << 3), u32x2{d0[2], d0[3]});
        lds_write_b64(ds_row + (((4 + hi) ^ ds_x) << 3), u32x2{d1[0], d1[1]});
        lds_write_b64(ds_row + (((6 + hi) ^ ds_x) << 3), u32x2{d1[2], d1[3]});
    }
}

// dV += P^T dO, dK += dS^T Q (transposed fragments of the dO / Q tiles in buffer BUF)
template <int BUF>
LWM_DEVICE void fb_tile_c(const FusedCtx& cx, const bf16x8 (&pb)[2], const bf16x8 (&dsb)[2],
                          f32x16 (&dk)[4], f32x16 (&dv)[4]) {
    constexpr uint32_t QB = BUF * kDkvQTileBytes;
    constexpr uint32_t DB = (2 + BUF) * kDkvQTileBytes;
    const int lane = (int)opaque((uint32_t)cx.lane);
    const lds_t qtiles = cx.lds + kFbOffTiles;
    uint32_t lo0, up0;
    {
        const int g4 = lane >> 4, i = lane & 15;
        const int row = 4 * (g4 >> 1) + (i >> 2);
        const int d = 16 * (g4 & 1) + 4 * (i & 3);
        lo0 = qtiles + tile_off(row, d >> 3) + (d & 7) * 2;
        up0 = qtiles + tile_off(row + 8, d >> 3) + (d & 7) * 2;
    }
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
        sched_fence();
        const int t = (h & 7) >> 2, db = h & 3;
        if (h < 8) dv[db] = mfma_32x32x16(ft[h % kRing], pb[t], dv[db]);
        else dk[db] = mfma_32x32x16(ft[h % kRing], dsb[t], dk[db]);
        sched_fence();
    }
    prio_lo();
}

// dQ^T[32 head dims][16 queries] of this wave = K^T . dS^T over the workgroup's 256 keys (dS^T buffer BUF).
// Wave w owns head dims 32*(w&3) .. +31 and queries 16*(w>>2) .. +15 of the tile: per query row that is
// 128 contiguous bytes of the f32 accumulator -- whole cache lines, shared with no other wave (the ordered
// hand-off is per wave, and a line shared by two waves could be stale in L1 for the second one).
// acc[t] is the 16x16 tile of head dims 32*(w&3) + 16t .. +15.
template <int BUF>
LWM_DEVICE void fb_dq_product(const FusedCtx& cx, f32x4 (&acc)[2]) {
    // operand addresses of k-step 0: 16-lane group kg covers keys 8kg..8kg+7 of the step
    const int lane = (int)opaque((uint32_t)cx.lane);
    const int i = lane & 15, kg = lane >> 4, j = i >> 2, cc = i & 3;
    const int db = cx.wave & 3, qh = cx.wave >> 2;
    const int d = 32 * db + 4 * cc;              // d-tile 1 = +16 head dims = +2 sixteen-byte slots (swizzled per row below)
    const int row = 8 * kg + j;
    const uint32_t klo0 = cx.lds + tile_off(row, d >> 3) + (d & 7) * 2;
    const uint32_t kup0 = cx.lds + tile_off(row + 4, d >> 3) + (d & 7) * 2;
    const uint32_t klo1 = cx.lds + tile_off(row, (d + 16) >> 3) + (d & 7) * 2;
    const uint32_t kup1 = cx.lds + tile_off(row + 4, (d + 16) >> 3) + (d & 7) * 2;
    const lds_t dsb = cx.lds + kFbOffDs + BUF * kFbDsBytes;
    const uint32_t l0 = dsb + ds_off(row, 4 * qh + cc), u0 = dsb + ds_off(row + 4, 4 * qh + cc);
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto cat = [](bf16x4 lo, bf16x4 up) {
        bf16x8 o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
        o[4] = up[0]; o[5] = up[1]; o[6] = up[2]; o[7] = up[3];
        return o;
    };
    // k-step ks covers keys 32ks..32ks+31: +32 key rows = +8192 B in the K block, +2048 B in dS^T
    bf16x8 a0[2], a1[2], bq[2];
    auto load = [&](int ks) {
        a0[ks & 1] = cat(lds_read_tr16(klo0 + ks * 8192), lds_read_tr16(kup0 + ks * 8192));
        a1[ks & 1] = cat(lds_read_tr16(klo1 + ks * 8192), lds_read_tr16(kup1 + ks * 8192));
        bq[ks & 1] = cat(lds_read_tr16(l0 + ks * 2048), lds_read_tr16(u0 + ks * 2048));
    };
    prio_hi();
    load(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) load(ks + 1);
        sched_fence();
        acc[0] = mfma_16x16x32(a0[ks & 1], bq[ks & 1], acc[0]);
        acc[1] = mfma_16x16x32(a1[ks & 1], bq[ks & 1], acc[1]);
        sched_fence();
    }
    prio_lo();
}

// ---- ordered accumulation of this wave's dQ slice (32 head dims x 16 queries of tile `qt`) into global
// memory.  Ordering is PER WAVE SLICE: wave w of key block kbi waits for wave w of key block kbi-1 on
// flag[(hb*nqt + qt)*8 + w], so no workgroup barrier sits between a wave's stores and its publication.
struct DqRmw {
    float* acc;             // this lane's 4 floats of d-tile 0 in the f32 accumulator (d-tile 1: + 16 floats)
    bool ok;                // the lane's query row exists
};

// after the flag says the previous contributor is done: start reading what it left.  Agent-scope (sc1) loads:
// they bypass this CU's L1 (which could hold the line from an earlier work item) and are served by the XCD's
// L2, where the predecessor's plain stores left the line (scripts/micro/l2_handoff.hip: FETCH_SIZE = first
// touch only; the same chain with plain loads hits L2 too but takes 2.3 us per hop instead of 0.8).
// Rows past Sq are clamped, not predicated (fb_stage_issue).
LWM_DEVICE void fb_dq_load(const AttnParams& p, const FusedCtx& cx, int b, int h, int qt, bool real, const float* dummy,
                           DqRmw& w) {
    const int lane = (int)opaque((uint32_t)cx.lane);
    const int n = lane & 15, kg = lane >> 4;
    const int db = cx.wave & 3, qh = cx.wave >> 2;
    const int64_t row0 = (int64_t)qt * kDkvBQ;
    const bool use = real && p.dq_acc != nullptr;
    const float* tile = use ? p.dq_acc + ((int64_t)b * p.dqa_sb + row0 * p.dqa_ss + (int64_t)h * p.dqa_sh) : dummy;
    const int64_t rstride = use ? p.dqa_ss : 0;
    const int rows_left = (int)(p.Sq - row0);        // >= 1
    const int r = 16 * qh + n;
    w.ok = r < rows_left;
    const int rc = w.ok ? r : rows_left - 1;
    // by LDS-DMA: lane l's 16 bytes of d-tile t land at slot + 1024 t + 16 l -- no registers are held while the
    // dQ product and the dV/dK phase run, and the read can be issued before both
    const lds_t slot = cx.lds + kFbOffDq + (uint32_t)cx.wave * 2048;
    const float* src = tile + (int64_t)rc * rstride + (use ? 32 * db + 4 * kg : 0);
    w.acc = const_cast<float*>(src);                 // (stored to only when `use` and w.ok: then it IS the lane's slot)
    for (int t = 0; t < 2; ++t) glds_load_b128_l2(src + (use ? 16 * t : 0), slot + 1024 * t);
}

// prev + scale * partial -> f32 accumulator, or bf16 dq when this is the tile's last contributor
LWM_DEVICE void fb_dq_store(const AttnParams& p, const FusedCtx& cx, int b, int h, int qt, bool use_prev, bool last,
                            bool do_store, const DqRmw& w, const f32x4 (&acc)[2]) {
    const int lane = (int)opaque((uint32_t)cx.lane);
    const int n = lane & 15, kg = lane >> 4;
    const int db = cx.wave & 3, qh = cx.wave >> 2;
    const int64_t row = (int64_t)qt * kDkvBQ + 16 * qh + n;
    const bool to_bf16 = last && p.dq_final_out;
    // both 16-dim tiles are computed (both accumulator reads retired) BEFORE the first store is issued: with a
    // store in between, the wait for the second read would also wait for that store (one in-order counter).
    // The accumulator that was read is dropped by a bit mask, not a branch (a uniform branch around the only
    // use of a loaded register leaves it "maybe pending" for hipcc, see fb_stage_issue).
    const uint32_t keep = use_prev ? 0xffffffffu : 0u;
    const lds_t slot = cx.lds + kFbOffDq + (uint32_t)cx.wave * 2048 + 16 * (uint32_t)lane;
    f32x4 o[2];
    for (int t = 0; t < 2; ++t) {
        const f32x4 prev = lds_read_f32x4(slot + 1024 * t);
        for (int j = 0; j < 4; ++j) {
            const float pj = prev[j];          // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0)
            const float pv = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, pj) & keep);
            o[t][j] = fmaf(acc[t][j], p.scale, pv);
            pin_value(o[t][j]);                 // pin the use HERE: LLVM otherwise sinks it into the store's branch
        }
    }
    sched_fence();
    if (!do_store || !w.ok) return;
    for (int t = 0; t < 2; ++t) {
        const int d0 = 32 * db + 16 * t + 4 * kg;
        if (to_bf16) {
            bf16_t* dst = p.dq + (int64_t)b * p.dq_sb + row * p.dq_ss + (int64_t)h * p.dq_sh + d0;
            global_store_b64_async(dst, u32x2{pack_bf16x2(o[t][0], o[t][1]), pack_bf16x2(o[t][2], o[t][3])});
        } else {
            global_store_f32x4_async(w.acc + 16 * t, o[t]);
        }
    }
}

// per-phase accounting of fb_step, -DLWM_PROF builds only (scripts/micro/fused_bench prints it): slots =
// S/dP phase, counted wait, turn wait, dQ product, dV/dK phase, full wait, accumulate + store, barrier, steps
#ifdef LWM_PROF
struct FbProf { uint32_t v[9]; uint32_t last; };
#define FB_LAP(pf, slot)                                               \
    do {                                                               \
        const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();  \
        (pf).v[slot] += now_ - (pf).last;                              \
        (pf).last = now_;                                              \
    } while (0)
#else
struct FbProf {};
#define FB_LAP(pf, slot)
#endif

// wait until `flag` (this wave's slice of the tile) shows that `order` contributors are done
LWM_DEVICE void fb_wait_turn(int seen, int order, const int32_t* flag, int32_t* err) {
    // The common case (the predecessor is ahead) must not meet a compiler-visible vector-memory load: hipcc
    // would put the s_waitcnt vmcnt(0) of the spin loop's load in front of the first test as well, and that
    // wait also drains the staging DMA in flight.  Hence the test outside, the loop bottom-tested.
    if (wave_uniform(seen) >= order) return;
    int spins = 0;
    do {
        if (++spins > kFbSpinLimit) {
            store_i32_plain(err, 1);
            break;
        }
        spin_pause();
        seen = load_i32_l2(flag);
    } while (wave_uniform(seen) < order);
}

// One step of the tile loop: tile `qt` (LDS buffer BUF) and the dQ of the PREVIOUS tile `qt_prev` (its dS^T
// in buffer PB).  Memory latencies are kept off the critical path:
//   flag poll   issued at the top, consumed after the S/dP and softmax phases;
//   read of the accumulator   issued after the dQ product, consumed after the dV/dK phase;
//   stores   issued at the end, NOT waited for here: the next step's mid-point vmcnt(0) covers them and
//            only then is this wave's flag for that tile published (one step later than the stores).
template <int BUF, int PB>
LWM_DEVICE void fb_step(const AttnParams& p, const FusedCtx& cx, const bf16x8 (&vf)[8], f32x16 (&dk)[4], f32x16 (&dv)[4],
                        int b, int h, int qt, int krel, int qlim, bool has_prev, int qt_prev, bool has_pub, int qt_pub,
                        int kbi, int qt_next, bool tail_block, int32_t* flags_h, int32_t* err, const float* dummy,
                        const bf16_t* qb, const bf16_t* dob, int qt_stage, FbProf& pf) {
    (void)pf;
    // Without a previous tile (step 0 of an item) the same instruction stream runs with nothing stored and the
    // accumulator read pointed at `dummy` (read-only data): straight-line code keeps hipcc's waits where they
    // belong (fb_stage_issue), and no line of the accumulator enters this CU's L1 before its turn.
    const int qp = has_prev ? qt_prev : qt;
    const int32_t* const flag_prev = flags_h + ((int64_t)qp * 8 + cx.wave);
    const int order = has_prev ? kbi : 0;            // key block 0 waits for nobody
    // Vector-memory operations retire in issue order.  Issued here, in this order: [the dq stores of the last
    // step] -> the flag poll (an LDS-DMA: its answer lands in LDS, no register, no compiler-tracked load) ->
    // the staging DMA of the next tile.  After the S/dP phase "at most the staging DMA outstanding" therefore
    // means "stores in L2, poll answered" -- without waiting for the staging, which has the whole step to land.
    const lds_t poll = cx.lds + kFbOffPoll + (uint32_t)cx.wave * 256;
    FB_LAP(pf, 7);             // since the end of the previous step: the tile barrier
    glds_load_b32_l2(flag_prev, poll);
    fb_stage_issue<BUF ^ 1>(p, cx, qb, dob, dummy, b, h, qt_stage);
    bf16x8 pb[2], dsb[2];
    fb_tile_ab<BUF>(p, cx, vf, krel, qlim, pb, dsb);
    FB_LAP(pf, 0);
    if (cx.n_dma == 3) wait_vmem_le<3>();
    else wait_vmem_le<2>();
    FB_LAP(pf, 1);
    if (has_pub && cx.lane == 0) store_i32_plain(flags_h + ((int64_t)qt_pub * 8 + cx.wave), kbi + 1);
    fb_wait_turn(lds_read_i32(poll + 4 * (uint32_t)cx.lane), order, flag_prev, err);
    FB_LAP(pf, 2);
    f32x4 acc[2];
    DqRmw w;
    fb_dq_load(p, cx, b, h, qp, has_prev, dummy, w);
    fb_dq_product<PB>(cx, acc);
    FB_LAP(pf, 3);
    fb_tile_c<BUF>(cx, pb, dsb, dk, dv);
    FB_LAP(pf, 4);
    wait_vmem_all();           // the staged tile (the barrier that follows publishes it) and the accumulator read
    FB_LAP(pf, 5);
    fb_dq_store(p, cx, b, h, qp, kbi != 0 || p.dq_carry_in, qp < qt_next || tail_block, has_prev, w, acc);
    FB_LAP(pf, 6);
#ifdef LWM_PROF
    pf.v[8] += 1;
#endif
}

LWM_KERNEL(kFbThreads) void attn_bwd_fused_kernel(AttnParams p, int32_t* ws) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = wave_uniform(tid >> 6);
    const int nkb = (p.Sk + kDkvBK - 1) / kDkvBK;
    const int nqt_all = (p.Sq + kDkvBQ - 1) / kDkvBQ;
    const int HB = p.H * p.B;
    int32_t* const tickets = ws;
    int32_t* const owners = ws + kFbQueues;
    int32_t* const sems = ws + kFbWsHeader;
    const float* const lse2 = (const float*)(ws + kFbWsHeader + (int64_t)HB * nqt_all * 8);
    const lds_t ctl = lds + kFbOffCtl;

    FusedCtx cx;
    cx.lds = lds;
    cx.tid = tid;
    cx.lane = lane;
    cx.wave = wave;
    cx.c = p.scale * kLog2e;
    cx.n_dma = (wave == 0 || (wave == 1 && p.seg_q)) ? 3 : 2;   // Q, dO (+ row statistics / segment ids)

    const int my_xcc = wave_uniform(xcc_id()) & (kFbQueues - 1);
    for (int qi = 0; qi < kFbQueues; ++qi) {
        const int que = (my_xcc + qi) & (kFbQueues - 1);
        const int heads = HB > que ? (HB - que + kFbQueues - 1) / kFbQueues : 0;
        const int items = heads * nkb;
        if (items == 0) continue;
        // ---- claim the queue for this XCD, or leave it to its owner
        if (tid == 0) {
            const int old = atomic_cas_i32(owners + que, 0, my_xcc + 1);
            lds_write_i32(ctl, (old == 0 || old == my_xcc + 1) ? 1 : 0);
        }
        block_sync();
        const bool mine = wave_uniform(lds_read_i32(ctl)) != 0;
        block_sync();
        if (!mine) continue;
        for (;;) {
            if (tid == 0) lds_write_i32(ctl + 4, atomic_add_i32(tickets + que, 1));
            block_sync();
            const int ticket = wave_uniform(lds_read_i32(ctl + 4));
            block_sync();
            if (ticket >= items) break;
            const int hb = que + kFbQueues * (ticket / nkb);
            const int kbi = ticket % nkb;      // ascending: the order of accumulation into dq
            const int b = hb / p.H, h = hb % p.H;

            const bf16_t* qb = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
            const bf16_t* kb = p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
            const bf16_t* vb = p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
            const bf16_t* dob = p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;
            int32_t* const flags_h = sems + (int64_t)hb * nqt_all * 8;

            // ---- this lane's key: V fragments in registers, key meta
            const int k_row = kbi * kDkvBK + wave * 32 + l31;
            const bool k_ok = k_row < p.Sk;
            bf16x8 vf[8];
            for (int s = 0; s < 8; ++s) {
                if (k_ok) vf[s] = __builtin_bit_cast(bf16x8, global_load_b128(vb + (int64_t)k_row * p.v_ss + 16 * s + 8 * hi));
                else vf[s] = zero_bf16x8();
            }
            cx.kseg = kSegInvalid;
            if (k_ok) {
                bool valid = p.key_valid ? (p.key_valid[(int64_t)b * p.Sk + k_row] != 0) : true;
                if (valid) cx.kseg = p.seg_k ? p.seg_k[(int64_t)b * p.Sk + k_row] : 0;
            }
            cx.has_meta = (p.seg_k != nullptr) || (p.key_valid != nullptr) || (kbi * kDkvBK + kDkvBK > p.Sk);
            // global position of this wave's first key minus that of query row 0 (wave-uniform, 64-bit)
            const int64_t wk_rel = p.k_start + (int64_t)kbi * kDkvBK + wave * 32 - p.q_start;
            // ---- resident K block (zero rows past Sk: they meet dS = 0)
            for (int i = 0; i < 4096 / kFbThreads; ++i) {
                int cidx = tid + kFbThreads * i;
                int row = cidx >> 4, slot = cidx & 15;
                int kr = kbi * kDkvBK + row;
                u32x4 val = {0u, 0u, 0u, 0u};
                if (kr < p.Sk) val = global_load_b128(kb + (int64_t)kr * p.k_ss + slot * 8);
                lds_write_b128(lds + tile_off(row, slot), val);
            }
            // ---- query tile range.  Key block 0 walks EVERY tile (it is the first contributor of each
            // tile's dq, also of tiles that see none of this K/V block's keys: those get exact zeros);
            // the others start at their causal diagonal.
            auto first_tile = [&](int kblk) -> int {
                if (!p.causal || kblk == 0) return 0;
                int64_t d = p.k_start + (int64_t)kblk * kDkvBK - p.q_start;
                if (d <= 0) return 0;
                int64_t t = d / kDkvBQ;
                return t < nqt_all ? (int)t : nqt_all;
            };
            const int qt0 = first_tile(kbi);
            const int qt_next = kbi + 1 < nkb ? first_tile(kbi + 1) : nqt_all;   // tiles < qt_next: this block is the last
            const bool tail_block = kbi == nkb - 1;
            const int n = nqt_all - qt0;               // tiles to walk: loop index i -> tile nqt_all-1-i (walk DOWN)
            auto krel_of = [&](int qt) -> int {
                int64_t r = wk_rel - (int64_t)qt * kDkvBQ;
                return r > 64 ? 64 : (r < -64 ? -64 : (int)r);
            };

            f32x16 dk[4], dv[4];
            for (int i = 0; i < 4; ++i) {
                dk[i] = zero_f32x16();
                dv[i] = zero_f32x16();
            }
            FbProf pf = {};
            FB_LAP(pf, 7);
#define LWM_FQT(i) (nqt_all - 1 - (i))
            if (n > 0) {
                if (!p.seg_q && tid < 2 * kDkvBQ)       // no segment ids: the table reads 0 (= kseg of every valid key)
                    lds_write_i32(lds + kFbOffStats + (tid >> 5) * kFbStatBytes + 2 * kDkvBQ * 4 + (tid & 31) * 4, 0);
                fb_stage_issue<0>(p, cx, qb, dob, lse2, b, h, LWM_FQT(0));
                if (kbi > 0 && kFbSlack > 0) {
                    const int ahead = n - 1 < kFbSlack ? n - 1 : kFbSlack;
                    const int32_t* const fl = flags_h + ((int64_t)LWM_FQT(ahead) * 8 + wave);
                    fb_wait_turn(load_i32_l2(fl), kbi, fl, ws + kFbWsErr);
                }
                wait_vmem_all();
                block_sync();
                auto qlim_of = [&](int qt) -> int {
                    const int left = p.Sq - qt * kDkvBQ;
                    return left < kDkvBQ ? left : kDkvBQ;
                };
                // step i: tile i + the dQ of tile i-1 + the publication of tile i-2 (see fb_step).  Two steps
                // per trip: the LDS buffers alternate.  Past the last tile the staging re-fetches the last
                // tile (unconditional instruction stream).
                for (int i = 0; i < n; i += 2) {
                    const bool more1 = i + 1 < n;
                    fb_step<0, 1>(p, cx, vf, dk, dv, b, h, LWM_FQT(i), krel_of(LWM_FQT(i)), qlim_of(LWM_FQT(i)), i > 0,
                                  LWM_FQT(i - 1), i > 1, LWM_FQT(i - 2), kbi, qt_next, tail_block, flags_h, ws + kFbWsErr, lse2,
                                  qb, dob, LWM_FQT(more1 ? i + 1 : i), pf);
                    block_sync_lds();      // (not __syncthreads: the dq stores stay in flight, see wave_ops.h)
                    if (!more1) break;
                    const bool more2 = i + 2 < n;
                    fb_step<1, 0>(p, cx, vf, dk, dv, b, h, LWM_FQT(i + 1), krel_of(LWM_FQT(i + 1)), qlim_of(LWM_FQT(i + 1)),
                                  true, LWM_FQT(i), i > 0, LWM_FQT(i - 1), kbi, qt_next, tail_block, flags_h,
                                  ws + kFbWsErr, lse2, qb, dob, LWM_FQT(more2 ? i + 2 : i + 1), pf);
                    block_sync_lds();
                }
                // drain: the dQ of the last tile (n-1) and the two publications still owed
                {
                    const int qt_last = LWM_FQT(n - 1);
                    int32_t* const flag_last = flags_h + ((int64_t)qt_last * 8 + wave);
                    const bool first = kbi == 0;
                    const int seen = load_i32_l2(flag_last);
                    wait_vmem_all();
                    if (n > 1 && lane == 0) store_i32_plain(flags_h + ((int64_t)LWM_FQT(n - 2) * 8 + wave), kbi + 1);
                    fb_wait_turn(seen, kbi, flag_last, ws + kFbWsErr);
                    f32x4 acc[2];
                    DqRmw w;
                    fb_dq_load(p, cx, b, h, qt_last, true, lse2, w);
                    if ((n - 1) & 1) fb_dq_product<1>(cx, acc);
                    else fb_dq_product<0>(cx, acc);
                    wait_vmem_all();       // the accumulator slice has landed in LDS
                    fb_dq_store(p, cx, b, h, qt_last, !first || p.dq_carry_in, qt_last < qt_next || tail_block, true, w, acc);
                    wait_vmem_all();
                    if (lane == 0) store_i32_plain(flag_last, kbi + 1);
                }
            }
#undef LWM_FQT
#ifdef LWM_PROF
            if (hb == 0 && kbi == nkb / 2 && lane == 0 && p.out_acc)     // one mid-chain item reports
                for (int i = 0; i < 9; ++i) ((unsigned long long*)p.out_acc)[wave * 10 + i] = pf.v[i];
#endif
            // ---- dK, dV of this key block
            if (k_ok) {
                const int64_t krow_o = (int64_t)b * p.dk_sb + (int64_t)k_row * p.dk_ss + (int64_t)h * p.dk_sh;
                const int64_t vrow_o = (int64_t)b * p.dv_sb + (int64_t)k_row * p.dv_ss + (int64_t)h * p.dv_sh;
                const int64_t arow = (((int64_t)b * p.Sk + k_row) * p.H + h) * kHeadDim;
                for (int db = 0; db < 4; ++db)
                    for (int rq = 0; rq < 4; ++rq) {
                        int d0 = 32 * db + 8 * rq + 4 * hi;
                        float k0 = dk[db][4 * rq + 0] * p.scale, k1 = dk[db][4 * rq + 1] * p.scale;
                        float k2 = dk[db][4 * rq + 2] * p.scale, k3 = dk[db][4 * rq + 3] * p.scale;
                        float v0 = dv[db][4 * rq + 0], v1 = dv[db][4 * rq + 1];
                        float v2 = dv[db][4 * rq + 2], v3 = dv[db][4 * rq + 3];
                        if (p.carry_in) {
                            const float* ka = p.dk_acc + arow + d0;
                            const float* va = p.dv_acc + arow + d0;
                            k0 += ka[0]; k1 += ka[1]; k2 += ka[2]; k3 += ka[3];
                            v0 += va[0]; v1 += va[1]; v2 += va[2]; v3 += va[3];
                        }
                        if (p.final_out) {
                            global_store_b64(p.dk + krow_o + d0, u32x2{pack_bf16x2(k0, k1), pack_bf16x2(k2, k3)});
                            global_store_b64(p.dv + vrow_o + d0, u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)});
                        } else {
                            global_store_b128(p.dk_acc + arow + d0,
                                              u32x4{__builtin_bit_cast(uint32_t, k0), __builtin_bit_cast(uint32_t, k1),
                                                    __builtin_bit_cast(uint32_t, k2), __builtin_bit_cast(uint32_t, k3)});
                            global_store_b128(p.dv_acc + arow + d0,
                                              u32x4{__builtin_bit_cast(uint32_t, v0), __builtin_bit_cast(uint32_t, v1),
                                                    __builtin_bit_cast(uint32_t, v2), __builtin_bit_cast(uint32_t, v3)});
                        }
                    }
            }
            block_sync();     // the K block and the tile buffers are rewritten by the next item
        }
    }
}

}  // namespace lwm
