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
// Per tile, beyond what that kernel does:
//   * dS^T (bf16, [256 keys][32 queries]) is written to LDS by the waves that produced it;
//   * one step later every wave w multiplies it by K for its 32 head dims and 16 of the queries:
//         dQ[16 q][32 d] = dS[16 q][256 keys] . K[256 keys][32 d]
//     on v_mfma_f32_16x16x32_bf16 (both operands by ds_read_b64_tr_b16 from the [key][*] images):
//     the reduction over the workgroup's 256 keys happens inside the MFMA accumulator, and a lane ends
//     up holding one head dim of four query rows, so that
//   * the partial is STORED -- not added: one 16-byte non-temporal store per lane, 8 bf16 = 4 query rows x 2 head
//     dims, 8 KiB per (key block, query tile) -- into a partial buffer in the caller's workspace, slot
//     (head, key block, query tile) of a causal-triangular arrangement (fb_slot).  Nothing is read back, nothing
//     is waited for, no key block waits for another.  attn_bwd_dq_reduce_kernel then streams the buffer once and
//     sums, per query tile, the partials of its key blocks IN ASCENDING KEY ORDER in f32 (+ the ring's dq carry),
//     scales, and writes dq (bf16) or the f32 accumulator: dq is bit-reproducible, like dk and dv.
//     Why stores and not adds (round 3, profiles/r03_atomic_probe.txt, r03_fused_atomic_timing.txt): f32 atomic
//     adds are performed at the memory side at ~1.25 TB/s whatever their width or scope (4- and 8-byte integer,
//     f32, f64, packed bf16: same BYTE rate), and a layer at S = 32768 issues 34 GB of them: 26.3 ms against 20.9
//     for the same launch without the adds.  bf16 partials are half the bytes, ordinary write-back traffic, and
//     one instruction per lane instead of eight; the reduction reads them back at streaming rate.
//     Numerics: a partial (the sum over 256 keys, accumulated in f32 by the MFMA) is rounded to bf16 once
//     (2^-9 relative, unbiased) before the f32 sum over key blocks -- the same order as the bf16 rounding of P and
//     dS that every backward kernel here already applies, and inside the stated tolerance (tests/_parity.py).
//
// Work is handed out through 8 queues (heads hb with hb % 8 == q); a queue is CLAIMED (atomic CAS) by the XCC id
// (s_getreg HW_REG_XCC_ID) of the first workgroup that takes from it and only workgroups of that XCD take from it
// afterwards; persistent workgroups drain their own XCD's queue and then claim what is unclaimed.  All key blocks
// of one (batch, head) are therefore executed by ONE XCD whatever the dispatcher does: their Q / dO stream is
// shared in that L2.  Placement is a matter of speed only: no workgroup reads what another wrote.
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
constexpr int kFbOffScan = kFbOffCtl + 64;          // seg_narrow scratch (8 waves x 8 B)
constexpr int kFbLdsBytes = kFbOffScan + 64;

// workspace (int32 words): [0,8) tickets | [8,16) queue owner (0 = unclaimed, xcc+1) | [16,32) unused |
// f32 [B,H,Sq]: LSE in log2 units (+inf where the row has no visible key), written by attn_bwd_lse2_kernel before
// the main launch | int32 [B*H][nkb][2]: the query-tile range [lo, hi) each key block walked (written by the main
// launch, read by the reduction) | 256-byte aligned: the partial tiles of the heads of ONE launch group.
constexpr int kFbWsHeader = 32;
constexpr int kFbPartBytes = kDkvBQ * kHeadDim * 2;      // one partial tile: 32 queries x 128 dims bf16 = 8 KiB
constexpr int kFbKbPerQt = kDkvBK / kDkvBQ;              // 8 query tiles per key block along the diagonal

// Where the partial tiles live.  Without hints key block j walks the query tiles [qt0(j), nqt), qt0(j) = the tile
// holding its causal diagonal = clamp(f + 8j, 0, nqt) with f = floor((k_start - q_start) / 32) (all of them when not
// causal).  Seen from query tile t that is the key blocks [0, count(t)), count(t) = clamp(floor((t - f) / 8) + 1, 0,
// nkb).  The tiles are stored QUERY-TILE MAJOR: the partials of tile t are count(t) consecutive 8 KiB pieces from slot
// prefix_q(t) = sum_{t' < t} count(t') of the head's area, key block j at + j -- so that the reduction of a tile is
// one contiguous streaming read (key-block-major slots made it 64 reads 8 MB apart: 2.6 TB/s instead of > 5).
struct FbGeom {
    int64_t f;          // floor((k_start - q_start) / 32)
    int32_t causal, nqt, nkb;
};
LWM_HD int fb_qt0(const FbGeom& g, int j) {
    if (!g.causal) return 0;
    const int64_t t = g.f + (int64_t)kFbKbPerQt * j;
    return t < 0 ? 0 : (t > g.nqt ? g.nqt : (int)t);
}
LWM_HD int fb_count(const FbGeom& g, int qt) {
    if (!g.causal) return g.nkb;
    const int64_t u = (int64_t)qt - g.f;
    if (u < 0) return 0;
    const int64_t c = u / kFbKbPerQt + 1;
    return c > g.nkb ? g.nkb : (int)c;
}
// sum_{u=0}^{m-1} min(nkb, floor(u / 8) + 1), m >= 0
LWM_HD int64_t fb_sum_counts(int64_t m, int nkb) {
    const int64_t lim = (int64_t)kFbKbPerQt * nkb;
    const int64_t mm = m < lim ? m : lim;
    const int64_t a = mm / kFbKbPerQt, b = mm % kFbKbPerQt;
    int64_t s = (kFbKbPerQt / 2) * a * (a + 1) + b * (a + 1);
    if (m > lim) s += (m - lim) * nkb;
    return s;
}
LWM_HD int64_t fb_prefix_q(const FbGeom& g, int qt) {
    if (!g.causal) return (int64_t)qt * g.nkb;
    const int64_t u1 = (int64_t)qt - g.f;          // tiles t in [0, qt) have u = t - f in [-f, u1); only u >= 0 count
    if (u1 <= 0) return 0;
    const int64_t u0 = g.f < 0 ? -g.f : 0;
    return fb_sum_counts(u1, g.nkb) - fb_sum_counts(u0, g.nkb);
}
struct FbPart {
    u32x4* tiles;           // partial tiles of head hb0 (16-byte units: a tile = 512 of them)
    int32_t* ranges;        // [B*H][nkb][2]
    int64_t tiles_per_head; // fb_prefix_q(nqt)
    FbGeom g;
    int32_t hb0, hbn;       // this launch covers batch*head indices [hb0, hb0 + hbn), hb0 % 8 == 0
    int32_t hints;          // segment-block hints given: a key block may have walked less than its causal range
};

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

// The reduction of the partial tiles: dq[tile] = scale * sum over key blocks (ascending) of its partials
// (+ the ring's f32 dq carry), written as bf16 dq (dq_final_out) or into the f32 accumulator.  A workgroup takes
// one (head, 32-query tile) at a time -- its partials are contiguous (8 KiB per key block); thread (wave w, lane l)
// owns the 16 bytes the producing lane (w, l) stored (fb_dq_partial), kFbRedBatch partials per trip, two trips in
// flight.  The sums cross an LDS tile (row stride 132 floats) so that the carry is read and the result written in
// whole 32- / 16-byte pieces of one (query, head) row.
constexpr int kFbRedThreads = 512;
constexpr int kFbRedStride = kHeadDim + 4;
constexpr int kFbRedLdsBytes = kDkvBQ * kFbRedStride * 4;
constexpr int kFbRedBatch = 8;
LWM_DEVICE void fb_red_add(float (&acc)[8], u32x4 v) {
    for (int j = 0; j < 4; ++j) {
        acc[2 * j] += __builtin_bit_cast(float, v[j] << 16);
        acc[2 * j + 1] += __builtin_bit_cast(float, v[j] & 0xffff0000u);
    }
}
LWM_KERNEL(kFbRedThreads) void attn_bwd_dq_reduce_kernel(AttnParams p, FbPart part) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx(), w = tid >> 6, l = tid & 63;
    const int nqt = part.g.nqt, nkb = part.g.nkb;
    const int64_t items = (int64_t)part.hbn * nqt;
    for (int64_t it = block_idx_x(); it < items; it += grid_dim_x()) {
        const int hbl = (int)(it / nqt), qt = (int)(it % nqt);
        const int hb = part.hb0 + hbl;
        const int b = hb / p.H, h = hb % p.H;
        const int cnt = fb_count(part.g, qt);
        const u32x4* const src = part.tiles + ((int64_t)hbl * part.tiles_per_head + fb_prefix_q(part.g, qt)) * (kFbPartBytes / 16) + tid;
        float acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        if (!part.hints) {
            // every key block of [0, cnt) wrote its partial: a plain stream, two trips of loads in flight
            u32x4 va[kFbRedBatch], vb[kFbRedBatch];
            int kb = 0;
            const int full = cnt - cnt % (2 * kFbRedBatch);
            if (full > 0)
                for (int u = 0; u < kFbRedBatch; ++u) va[u] = global_load_b128(src + (int64_t)u * (kFbPartBytes / 16));
            for (; kb < full; kb += 2 * kFbRedBatch) {
                for (int u = 0; u < kFbRedBatch; ++u) vb[u] = global_load_b128(src + (int64_t)(kb + kFbRedBatch + u) * (kFbPartBytes / 16));
                for (int u = 0; u < kFbRedBatch; ++u) fb_red_add(acc, va[u]);
                if (kb + 2 * kFbRedBatch < full)
                    for (int u = 0; u < kFbRedBatch; ++u) va[u] = global_load_b128(src + (int64_t)(kb + 2 * kFbRedBatch + u) * (kFbPartBytes / 16));
                for (int u = 0; u < kFbRedBatch; ++u) fb_red_add(acc, vb[u]);
            }
            for (; kb < cnt; ++kb) fb_red_add(acc, global_load_b128(src + (int64_t)kb * (kFbPartBytes / 16)));
        } else {
            // packed batches: a key block walked only the tiles [lo, hi) its documents can meet
            const int32_t* rg = part.ranges + (int64_t)hb * nkb * 2;
            for (int kb0 = 0; kb0 < cnt; kb0 += kFbRedBatch) {
                u32x4 v[kFbRedBatch];
                bool ok[kFbRedBatch];
                bool any = false;
                for (int u = 0; u < kFbRedBatch; ++u) {
                    const int kb = kb0 + u < cnt ? kb0 + u : cnt - 1;
                    ok[u] = kb0 + u < cnt && rg[2 * kb] <= qt && qt < rg[2 * kb + 1];
                    any |= ok[u];
                }
                if (!any) continue;
                // (an absent partial re-reads a present slot position -- the value is discarded; the load stays
                // unconditional so that all of a trip's loads are in flight together)
                for (int u = 0; u < kFbRedBatch; ++u) v[u] = global_load_b128(src + (int64_t)(ok[u] ? kb0 + u : kb0) * (kFbPartBytes / 16));
                for (int u = 0; u < kFbRedBatch; ++u)
                    if (ok[u]) fb_red_add(acc, v[u]);
            }
        }
        // element 2r + t = dQ[row 16*(w>>2) + 4*(l>>4) + r][dim 32*(w&3) + 16t + (l&15)]
        for (int r = 0; r < 4; ++r)
            for (int t = 0; t < 2; ++t)
                lds_write_f32(lds + (uint32_t)((16 * (w >> 2) + 4 * (l >> 4) + r) * kFbRedStride + 32 * (w & 3) + 16 * t + (l & 15)) * 4,
                              acc[2 * r + t] * p.scale);
        block_sync();
        const int row = tid >> 4, c8 = tid & 15;
        const int64_t q = (int64_t)qt * kDkvBQ + row;
        if (q < p.Sq) {
            f32x4 lo = lds_read_f32x4(lds + (uint32_t)(row * kFbRedStride + 8 * c8) * 4);
            f32x4 hi = lds_read_f32x4(lds + (uint32_t)(row * kFbRedStride + 8 * c8 + 4) * 4);
            float* const ap = p.dq_acc ? p.dq_acc + b * p.dqa_sb + q * p.dqa_ss + h * p.dqa_sh + c8 * 8 : nullptr;
            if (p.dq_carry_in) {
                lo += global_load_f32x4(ap);
                hi += global_load_f32x4(ap + 4);
            }
            if (p.dq_final_out) {
                const u32x4 o = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]),
                                 pack_bf16x2(hi[2], hi[3])};
                global_store_b128(p.dq + b * p.dq_sb + q * p.dq_ss + h * p.dq_sh + c8 * 8, o);
            } else {
                global_store_f32x4(ap, lo);
                global_store_f32x4(ap + 4, hi);
            }
        }
        block_sync();
    }
}

// Staging of the next tile: Q and dO rows AND the row statistics go global -> LDS directly (inline-asm
// DMA, see wave_ops.h): no VGPR is involved and hipcc sees no load.  That matters beyond the registers:
// hipcc's waitcnt insertion is flow-insensitive, and a compiler-visible load that is issued or consumed
// under a predicate leaves a "maybe pending" mark that turns into an s_waitcnt vmcnt(N) at some later
// redefinition of its register -- here in front of the first MFMA of the NEXT step, where vmcnt also
// counts the just-issued DMA and the atomic adds (measured: ~1 us per step).  Rows past Sq re-read the
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

// dQ[16 queries][32 head dims] of this wave = dS . K over the workgroup's 256 keys (dS^T buffer BUF), stored as a
// bf16 partial.  Wave w owns head dims 32*(w&3) .. +31 and queries 16*(w>>2) .. +15 of the tile.
// MFMA 16x16x32 with A = dS (row = query l&15, k-group l>>4: 8 keys), B = K (k-group, col = head dim l&15):
// acc[t][r] = dQ[query 16*(w>>2) + 4*(l>>4) + r][head dim 32*(w&3) + 16t + (l&15)].  The lane's 8 values go out as
// ONE 16-byte store at tile + 16 * (64w + l): element 2r + t (attn_bwd_dq_reduce_kernel knows the map).  Rows past
// Sq hold zeros (dS is masked there).
template <int BUF>
LWM_DEVICE void fb_dq_partial(const FusedCtx& cx, u32x4* tile) {
    const int lane = (int)opaque((uint32_t)cx.lane);
    const int i = lane & 15, kg = lane >> 4, j = i >> 2, cc = i & 3;
    const int db = cx.wave & 3, qh = cx.wave >> 2;
    const int d = 32 * db + 4 * cc;              // d-tile 1 = +16 head dims = +2 sixteen-byte slots (swizzled per row below)
    const int row = 8 * kg + j;
    // operand addresses of k-step 0: 16-lane group kg covers keys 8kg..8kg+7 of the step
    const uint32_t klo0 = cx.lds + tile_off(row, d >> 3) + (d & 7) * 2;
    const uint32_t kup0 = cx.lds + tile_off(row + 4, d >> 3) + (d & 7) * 2;
    const uint32_t klo1 = cx.lds + tile_off(row, (d + 16) >> 3) + (d & 7) * 2;
    const uint32_t kup1 = cx.lds + tile_off(row + 4, (d + 16) >> 3) + (d & 7) * 2;
    const lds_t dsb = cx.lds + kFbOffDs + BUF * kFbDsBytes;
    const uint32_t l0 = dsb + ds_off(row, 4 * qh + cc), u0 = dsb + ds_off(row + 4, 4 * qh + cc);
    f32x4 acc[2];
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
        acc[0] = mfma_16x16x32(bq[ks & 1], a0[ks & 1], acc[0]);
        acc[1] = mfma_16x16x32(bq[ks & 1], a1[ks & 1], acc[1]);
        sched_fence();
    }
    prio_lo();
    const u32x4 o = {pack_bf16x2(acc[0][0], acc[1][0]), pack_bf16x2(acc[0][1], acc[1][1]),
                     pack_bf16x2(acc[0][2], acc[1][2]), pack_bf16x2(acc[0][3], acc[1][3])};
#ifndef LWM_FB_NO_STORE      // (timing-only builds switch parts of the step off)
    global_store_b128_nt_at(tile, (uint32_t)(64 * cx.wave + lane) * 16u, o);
#else
    asm volatile("" ::"v"(o));
#endif
}

#ifdef LWM_FB_NO_DQ
#define LWM_FB_DQ(x) (void)0
#else
#define LWM_FB_DQ(x) x
#endif
LWM_KERNEL(kFbThreads) void attn_bwd_fused_kernel(AttnParams p, int32_t* ws, FbPart part) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = wave_uniform(tid >> 6);
    const int nkb = (p.Sk + kDkvBK - 1) / kDkvBK;
    const int nqt_all = (p.Sq + kDkvBQ - 1) / kDkvBQ;
    const int HB = part.hbn;          // heads of this launch group (queues: local head index % 8)
    int32_t* const tickets = ws;
    int32_t* const owners = ws + kFbQueues;
    const float* const lse2 = (const float*)(ws + kFbWsHeader);
    const lds_t ctl = lds + kFbOffCtl;

    FusedCtx cx;
    cx.lds = lds;
    cx.tid = tid;
    cx.lane = lane;
    cx.wave = wave;
    cx.c = p.scale * kLog2e;

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
            const int hbl = que + kFbQueues * (ticket / nkb);      // head within the group
            const int hb = part.hb0 + hbl;
            const int kbi = ticket % nkb;      // ascending: under a causal mask the longest walks first
            const int b = hb / p.H, h = hb % p.H;

            const bf16_t* qb = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
            const bf16_t* kb = p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
            const bf16_t* vb = p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
            const bf16_t* dob = p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;

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
            // ---- query tile range: from the block's causal diagonal to the last tile, narrowed to the
            // documents this key block belongs to when segment-block hints are given
            int qt0 = 0, qt1 = nqt_all;
            if (p.causal) {
                int64_t d = p.k_start + (int64_t)kbi * kDkvBK - p.q_start;
                if (d > 0) qt0 = (int)(d / kDkvBQ < nqt_all ? d / kDkvBQ : nqt_all);
            }
            if (p.segb_q && p.segb_k && qt0 < qt1) {
                const int nbq = (p.Sq + 31) >> 5, nbk = (p.Sk + 31) >> 5;
                int smin, smax, lo, hi2;
                seg_own_range(p.segb_k + (int64_t)b * nbk * 2, nbk, kbi * (kDkvBK / 32), kDkvBK / 32, smin, smax);
                seg_narrow<kFbThreads>(p.segb_q + (int64_t)b * nbq * 2, nbq, kDkvBQ / 32, qt0, qt1, smin, smax,
                                       lds + kFbOffScan, tid, lo, hi2);
                qt0 = lo;
                qt1 = hi2;
            }
            const int n = qt1 - qt0;                   // tiles to walk: loop index i -> tile qt1-1-i (walk DOWN)
            // what the reduction will read of this key block: tiles [qt0, qt1) (empty when n <= 0)
            if (tid == 0) {
                int32_t* rg = part.ranges + ((int64_t)hb * nkb + kbi) * 2;
                rg[0] = qt0;
                rg[1] = n > 0 ? qt1 : qt0;
            }
            // the partial of (this key block, query tile t) goes to slot prefix_q(t) + kbi of this head's area; the walk
            // goes DOWN from tile qt1 - 1, one partial per step, so the slot follows incrementally
            u32x4* const ptiles = part.tiles + ((int64_t)hbl * part.tiles_per_head + kbi) * (kFbPartBytes / 16);
            int pqt = qt1 - 1;                              // tile of the next partial
            int64_t pslot = n > 0 ? fb_prefix_q(part.g, pqt) : 0;
            auto next_tile = [&]() -> u32x4* {
                u32x4* const t = ptiles + pslot * (kFbPartBytes / 16);
                pqt -= 1;
                pslot -= pqt >= 0 ? fb_count(part.g, pqt) : 0;
                return t;
            };
            auto krel_of = [&](int qt) -> int {
                int64_t r = wk_rel - (int64_t)qt * kDkvBQ;
                return r > 64 ? 64 : (r < -64 ? -64 : (int)r);
            };

            f32x16 dk[4], dv[4];
            for (int i = 0; i < 4; ++i) {
                dk[i] = zero_f32x16();
                dv[i] = zero_f32x16();
            }
#define LWM_FQT(i) (qt1 - 1 - (i))
            if (n > 0) {
                if (!p.seg_q && tid < 2 * kDkvBQ)       // no segment ids: the table reads 0 (= kseg of every valid key)
                    lds_write_i32(lds + kFbOffStats + (tid >> 5) * kFbStatBytes + 2 * kDkvBQ * 4 + (tid & 31) * 4, 0);
                fb_stage_issue<0>(p, cx, qb, dob, lse2, b, h, LWM_FQT(0));
                wait_vmem_all();
                block_sync();
                auto qlim_of = [&](int qt) -> int {
                    const int left = p.Sq - qt * kDkvBQ;
                    return left < kDkvBQ ? left : kDkvBQ;
                };
                // step i: [the dQ of tile i-1 from the dS^T it left in LDS -> one partial store] | staging DMA of
                // tile i+1 | S, dP, P, dS of tile i (dS^T -> LDS) | dV, dK | wait for the DMA | barrier.
                // The store is the OLDEST vector-memory operation of the step when its vmcnt(0) comes, with
                // the whole step behind it (the other order -- dQ product and store LAST in the step, vmcnt(1) at its
                // end -- measured 26.2 vs 25.6 ms per layer, profiles/r03_fused_partials.md).  Two steps per trip: the
                // LDS buffers alternate.  Past the last tile the staging re-fetches the last tile (unconditional
                // instruction stream).
                for (int i = 0; i < n; i += 2) {
                    {
                        const bool more1 = i + 1 < n;
                        if (i > 0) LWM_FB_DQ(fb_dq_partial<1>(cx, next_tile()));
                        fb_stage_issue<1>(p, cx, qb, dob, lse2, b, h, LWM_FQT(more1 ? i + 1 : i));
                        bf16x8 pb[2], dsb[2];
                        fb_tile_ab<0>(p, cx, vf, krel_of(LWM_FQT(i)), qlim_of(LWM_FQT(i)), pb, dsb);
                        fb_tile_c<0>(cx, pb, dsb, dk, dv);
                        wait_vmem_all();
                        block_sync_lds();
                        if (!more1) break;
                    }
                    {
                        const bool more2 = i + 2 < n;
                        LWM_FB_DQ(fb_dq_partial<0>(cx, next_tile()));
                        fb_stage_issue<0>(p, cx, qb, dob, lse2, b, h, LWM_FQT(more2 ? i + 2 : i + 1));
                        bf16x8 pb[2], dsb[2];
                        fb_tile_ab<1>(p, cx, vf, krel_of(LWM_FQT(i + 1)), qlim_of(LWM_FQT(i + 1)), pb, dsb);
                        fb_tile_c<1>(cx, pb, dsb, dk, dv);
                        wait_vmem_all();
                        block_sync_lds();
                    }
                }
                // drain: the dQ of the last tile
                if ((n - 1) & 1) LWM_FB_DQ(fb_dq_partial<1>(cx, next_tile()));
                else LWM_FB_DQ(fb_dq_partial<0>(cx, next_tile()));
            }
#undef LWM_FQT
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
