// attn_fwd64.h -- blockwise attention forward, ONE WAVE PER SIMD: the training-path kernel for one
// (q block, kv block) ring step on gfx950.  Requires wave_ops.h + attn_common.h + attn_fwd.h (constants).
//
// Replaces the forward of `ringattention` as called at lwm/llama.py:539-569 (blockwise online-softmax
// update, fp32 logits, causal_block_size=1, masks per lwm/llama.py:527-537 and :572-592) -- the same
// contract as attn_fwd.h, which stays for the dense-mask / split-K inference flavour.
//
// Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD" attention): workgroup = 4 waves = 256
// queries; a wave owns 64 query rows (two 32-row blocks) and the SIMD's whole register file, so every K
// fragment (ds_read_b128) and every V fragment (2 x ds_read_b64_tr_b16) feeds TWO MFMAs, fragment
// addresses live in registers (no per-read address arithmetic), and the softmax of one key tile is
// interleaved, in program order, with the MFMAs of its neighbours -- with a single wave on the SIMD
// nothing else would fill the matrix pipe while the wave does its exponentials:
//
//   iteration i (tile i = 64 keys)       matrix pipe                     vector pipe (same wave)
//     phase 1 (32 MFMAs)   S(i+1) = K(i+1) Q~^T - base        p(i) = exp2(S(i)), P(i) -> bf16
//     phase 2 (32 MFMAs)   O^T   += V(i)^T P(i)^T             row sums of p(i), running max of S(i+1)
//
// Scores are computed transposed (S^T = K Q^T: a lane owns one query column, as in attn_fwd.h).  What is
// new in the arithmetic:
//   * p = exp2(s*c - m*c), c = scale*log2(e), is ONE fma + one exp per score (m = the query's running reference, its
//     maximum at the last rescale; scores stay raw q.k in their registers.  Pre-multiplying Q by c and rounding it to
//     bf16 once per workgroup saves the multiply but moves the LSE by up to ~2e-3: measured in round 3 and dropped,
//     profiles/r03_fwd64_dma.txt);
//   * the reference moves only when some row would exceed 2^kDeferLog2 (deferred rescale, as before) and the
//     decision needs no cross-lane traffic; the rescale itself is a rare, non-interleaved block that runs
//     after the P.V of the tile in flight is complete.
// K/V tiles travel global -> LDS by LDS-DMA (no registers, no ds_write): K two tiles ahead, V one, one
// workgroup barrier per tile.
//
// LDS map: K tile 0 | K tile 1 | V tile 0 | V tile 1 (16 KiB each) | key meta 0 | 1 | 2 (256 B each) | scan scratch
#pragma once

namespace lwm {

constexpr int kF4BQ = 256;      // queries per workgroup
constexpr int kF4BK = 64;       // keys per LDS tile
constexpr int kF4Threads = 256;
constexpr int kF4TileBytes = kF4BK * kRowBytes;                 // 16 KiB
constexpr int kF4OffMeta = 4 * kF4TileBytes;
constexpr int kF4OffScan = kF4OffMeta + 3 * kF4BK * 4;   // three key-meta buffers: tile t in buffer t % 3
constexpr int kF4LdsBytes = kF4OffScan + 64;
// LDS fragments (K row fragments for S, V transposed fragments for P.V: eight of each per half tile) are requested
// kF4Ahead MFMA pairs before the MFMA that consumes them, through register rings of eight (index = fragment): with ONE
// wave on the SIMD nothing else covers the LDS latency, and 512 registers leave room for the deeper ring.  The first
// kF4Ahead fragments of a phase are requested during the last gaps of the phase before it.  Measured (profiles/
// r03_fwd64_dma.txt): distance 3 / 5 / 6 / 7 = 7.50 / 7.40 / 7.40 / 7.36 ms per layer.
#ifndef LWM_F4_AHEAD
#define LWM_F4_AHEAD 7
#endif
constexpr int kF4Ahead = LWM_F4_AHEAD;
static_assert(kF4Ahead >= 1 && kF4Ahead <= 7, "prefetch distance in fragments");
constexpr int f4_pre_gap(int j) { return 17 - 2 * kF4Ahead + 2 * j; }      // the odd gap in which fragment j < kF4Ahead of the NEXT phase is requested
// gaps (MFMA index inside a phase) after which the hot loop issues its LDS-DMA pieces (odd gaps only -- the ones
// without an LDS read -- measured the same: 7.40 - 7.51 vs 7.37 - 7.43 ms)
constexpr int kF4G0a = 2, kF4G0b = 8, kF4G1a = 2, kF4G1b = 7, kF4G1c = 12, kF4G2a = 1, kF4G2b = 5, kF4G2c = 9;

struct F4Ctx {
    uint32_t ka[8];             // K row-fragment addresses, tile 0, keys 0..31, per 16-wide d step
    uint32_t vlo[4], vup[4];    // V transposed-fragment addresses, V tile 0, keys 0..15, per 32-wide d block
    lds_t lds;
    int tid, lane, hi, wave;
    float thr[2];               // raw-score threshold of the deferred rescale: -inf until the row's reference is set,
                                // then mref + kDeferLog2 / c
    float mref[2];              // the reference, a raw score q.k (-inf: none yet)
    float nbase[2];             // -mref * c (0 while mref = -inf) -- the fma's addend
    float lsum[2];              // this half-wave's partial row sum
    float c;                    // scale * log2(e): scores -> log2 units
    float thr_on;               // kDeferLog2 in raw score units
};

// ---- staging.  A tile = 16 pieces of 1 KiB (4 rows); a half tile (32 keys) = 8 pieces, of which wave w moves
// pieces w and w+4.  Lane l of a piece writes physical slot l&15 of row 4*piece + (l>>4), so it fetches logical
// slot (l&15) ^ swz(row) -- the XOR swizzle applied on the SOURCE column; swz(row) is the same for every piece
// of a wave (its pieces are 16 rows apart).
struct F4Stage {
    uint32_t voff_k[4], voff_v[4];    // per-lane byte offsets of the wave's pieces j = 0..3 (rows 4w + 16j + (l>>4))
    const char* kb;                   // K / V of this (batch, head), tile 0 (wave-uniform)
    const char* vb;
    int64_t ktile_bytes, vtile_bytes; // bytes between tiles
};

// N (2 or 4) LDS-DMA instructions of one wave: piece j from src + voff[j] to dst + 4096*j; src and dst
// wave-uniform.  One asm block: M0 is saved once; hipcc sees no load (the kernel owns the wait, wave_ops.h); the
// s_nop after the save covers a readfirstlane-made SGPR pair (VALU -> SGPR -> VMEM).
template <int N>
LWM_DEVICE void f4_dma(const uint32_t* voff, const char* src, lds_t dst) {
#ifdef LWM_EMU
    for (int j = 0; j < N; ++j) glds_load_b128(src + voff[j], dst + 4096 * j);
#else
    const uint64_t a = (uint64_t)src;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    unsigned keep;
    if (N == 4) {
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_nop 3\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
            "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
            "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
            "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(u), "s"(dst)
            : "memory", "scc");
    } else {
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_nop 3\n\t"
            "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
            "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "v"(voff[1]), "s"(u), "s"(dst)
            : "memory", "scc");
    }
#endif
}

// ONE LDS-DMA piece, placed between MFMAs of the hot loop (five scalar/vector-memory instructions; issuing the eight
// pieces of an iteration back to back at its top stalled the in-order wave for their whole issue time)
LWM_DEVICE void f4_dma1(uint32_t voff, const char* src, lds_t dst) {
#ifdef LWM_EMU
    glds_load_b128(src + voff, dst);
#else
    const uint64_t a = (uint64_t)src;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    // M0 is written and left: nothing else in the kernel reads it (LDS instructions do not on gfx9+, and the f4_dma<N>
    // blocks of the slow paths save and restore it themselves).  `src` is wave-uniform and, in the hot loop, produced
    // by scalar adds well before this point: no "VALU writes SGPR -> VMEM reads it" wait states are needed here.
    // Measured against the save / 4 wait states / write / load / restore form: 7.37 - 7.43 vs 7.55 - 7.70 ms per
    // layer (profiles/r03_fwd64_dma.txt: the three pieces of a P.V phase cost 76 cycles there and ~0 here).
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(u), "s"(dst) : "memory");
#endif
}
// the pieces one tile iteration issues inside its four phases (all wave-uniform but the offsets)
struct F4Dma {
    const char* k_lo_src;   // K(i+2), keys 0..31   -> k_lo_dst (+4096 for the second piece)
    const char* k_up_src;   // K(i+1), keys 32..63  -> k_up_dst
    const char* v_src;      // V(i+1)               -> v_dst (+4096 j)
    lds_t k_lo_dst, k_up_dst, v_dst;
};

// per-lane offsets for tile `kt` with rows clamped to Sk-1 (rows past Sk are masked through the key meta)
LWM_DEVICE void f4_stage_offsets(const AttnParams& p, int wave, int lane, int kt, F4Stage& st) {
    const int slot = lane & 15;
    for (int j = 0; j < 4; ++j) {
        const int row = 4 * (wave + 4 * j) + (lane >> 4);
        int krow = kt * kF4BK + row;
        krow = krow < p.Sk ? krow : p.Sk - 1;
        const int rel = krow - kt * kF4BK;      // >= 0 for every tile that exists
        const int col = (slot ^ swz(row)) << 3;
        st.voff_k[j] = (uint32_t)(((int64_t)rel * p.k_ss + col) * 2);
        st.voff_v[j] = (uint32_t)(((int64_t)rel * p.v_ss + col) * 2);
    }
}

// key meta of tile kt -> meta buffer `buf`: segment id, or kSegInvalid for padded / out-of-range keys
LWM_DEVICE void f4_meta_stage(const AttnParams& p, const F4Ctx& cx, int b, int kt, int buf) {
    if (cx.tid < kF4BK) {
        const int krow = kt * kF4BK + cx.tid;
        const int kr = krow < p.Sk ? krow : p.Sk - 1;
        const uint8_t kvalid = p.key_valid ? p.key_valid[(int64_t)b * p.Sk + kr] : (uint8_t)1;
        const int32_t kseg = p.seg_k ? p.seg_k[(int64_t)b * p.Sk + kr] : 0;
        const bool ok = krow < p.Sk && kvalid != 0;
        lds_write_i32(cx.lds + kF4OffMeta + buf * kF4BK * 4 + cx.tid * 4, ok ? kseg : kSegInvalid);
    }
}

// ---- hand-ordered instruction stream.  At one wave per SIMD hipcc selects the AGPR form for EVERY MFMA (C and D in
// the accumulator file) and would copy each score tile back to VGPRs for the softmax (v_accvgpr_read x 64 per tile),
// and its instruction selection lets pure VALU operations float to the end of a block whatever scheduling fences say.
// So the two phases are written as `asm volatile` statements, which keep their source order: the score MFMAs in VGPR
// form (C/D = the VGPR tuple the softmax then works on in place), the P.V MFMAs in AGPR form (O never leaves the
// accumulator file), Q~ as an AGPR B operand, and every filler between them.  What hipcc still does: LDS reads
// (builtins: it counts them and waits before the asm statement that consumes the fragment) and register allocation.
// Hazards hipcc cannot see through asm (cdna_hip_programming.md section 5.7 item 2) are covered here:
//   MFMA D (VGPR) -> VALU read      the fillers first read a score tile behind the next phase's first MFMA, i.e. after the
//                                   chain has left the matrix pipe; the mask code (hipcc's) gets f4_settle_s() first
//   v_exp (trans) -> VALU consumer  the cvt of a value comes at least one MFMA after its exp
//   VALU write -> MFMA A/B/C        P fragments are packed a phase before the MFMA that reads them
#ifdef LWM_EMU
LWM_DEVICE void f4_mfma_s_first(f32x16& d, bf16x8 a, bf16x8 b) { d = mfma_32x32x16(a, b, zero_f32x16()); }
LWM_DEVICE void f4_mfma_s(f32x16& d, bf16x8 a, bf16x8 b) { d = mfma_32x32x16(a, b, d); }
LWM_DEVICE void f4_mfma_o(f32x16& d, bf16x8 a, bf16x8 b) { d = mfma_32x32x16(a, b, d); }
LWM_DEVICE void f4_mfma_settle() {}
LWM_DEVICE void f4_settle_s(f32x16 (&)[2]) {}
LWM_DEVICE void f4_settle_acc(f32x16 (&)[2][4]) {}
LWM_DEVICE float f4_fma(float x, float c, float d) { return fmaf(x, c, d); }
LWM_DEVICE float f4_exp2(float x) { return exp2f(x); }
LWM_DEVICE uint32_t f4_cvt_pk(float lo, float hi) { return pack_bf16x2(lo, hi); }
LWM_DEVICE void f4_add(float& acc, float x) { acc += x; }
LWM_DEVICE float f4_add2(float a, float b) { return a + b; }
LWM_DEVICE void f4_max3(float& m, float a, float b) { m = fmaxf(fmaxf(m, a), b); }
LWM_DEVICE bf16x8 f4_load_agpr(const bf16_t* g) { return __builtin_bit_cast(bf16x8, global_load_b128(g)); }
LWM_DEVICE void f4_load_agpr_wait(bf16x8 (&)[2][8]) {}
LWM_DEVICE void f4_load_agpr_wait8(bf16x8 (&)[8]) {}
LWM_DEVICE void f4_scale_acc(f32x16& d, float alpha) { d *= alpha; }
#else
LWM_DEVICE void f4_mfma_s_first(f32x16& d, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));   // (early clobber: D must not overlap A)
}
LWM_DEVICE void f4_mfma_s(f32x16& d, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
LWM_DEVICE void f4_mfma_o(f32x16& d, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
// After the last MFMA of a chain whose result compiler-generated code reads next.  The wait states must sit between the
// MFMA and its reader IN THE INSTRUCTION STREAM: a bare nop statement orders nothing (hipcc may hoist a pure VALU reader
// of the tuple above it), so the tuples are named as read-write operands -- every later use then follows the nops.
LWM_DEVICE void f4_mfma_settle() { asm volatile("s_nop 7\n\ts_nop 7"); }
LWM_DEVICE void f4_settle_s(f32x16 (&s)[2]) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(s[0]), "+v"(s[1])); }
LWM_DEVICE void f4_settle_acc(f32x16 (&a)[2][4]) {
    asm volatile("s_nop 7\n\ts_nop 7"
                 : "+a"(a[0][0]), "+a"(a[0][1]), "+a"(a[0][2]), "+a"(a[0][3]), "+a"(a[1][0]), "+a"(a[1][1]), "+a"(a[1][2]), "+a"(a[1][3]));
}
LWM_DEVICE float f4_fma(float x, float c, float d) {
    float y;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(x), "v"(c), "v"(d));
    return y;
}
LWM_DEVICE float f4_exp2(float x) {
    float y;
    asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(x));
    return y;
}
LWM_DEVICE uint32_t f4_cvt_pk(float lo, float hi) {
    uint32_t y;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y) : "v"(lo), "v"(hi));
    return y;
}
LWM_DEVICE void f4_add(float& acc, float x) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x)); }
LWM_DEVICE float f4_add2(float a, float b) {
    float y;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}
// 16 bytes from global memory straight into the accumulator file (gfx90a+: a load may target AGPRs).  hipcc does not
// count the load: f4_load_agpr_wait names every destination, so that nothing is read or moved before the data is in
// (cdna_hip_programming.md section 5.7 item 1, form ii).
LWM_DEVICE bf16x8 f4_load_agpr(const bf16_t* g) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(r) : "v"(g) : "memory");
    return __builtin_bit_cast(bf16x8, r);
}
LWM_DEVICE void f4_load_agpr_wait(bf16x8 (&q)[2][8]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+a"(q[0][0]), "+a"(q[0][1]), "+a"(q[0][2]), "+a"(q[0][3]), "+a"(q[0][4]), "+a"(q[0][5]), "+a"(q[0][6]), "+a"(q[0][7]),
                   "+a"(q[1][0]), "+a"(q[1][1]), "+a"(q[1][2]), "+a"(q[1][3]), "+a"(q[1][4]), "+a"(q[1][5]), "+a"(q[1][6]), "+a"(q[1][7])
                 :
                 : "memory");
}
LWM_DEVICE void f4_load_agpr_wait8(bf16x8 (&q)[8]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+a"(q[0]), "+a"(q[1]), "+a"(q[2]), "+a"(q[3]), "+a"(q[4]), "+a"(q[5]), "+a"(q[6]), "+a"(q[7])
                 :
                 : "memory");
}
// acc *= alpha on a tuple that lives in the accumulator file, without ever showing hipcc a VGPR copy of it
LWM_DEVICE void f4_scale_acc(f32x16& d, float alpha) {
#define LWM_F4_SC(i) "v_accvgpr_read_b32 %16, %" #i "\n\ts_nop 0\n\tv_mul_f32 %16, %16, %17\n\ts_nop 0\n\tv_accvgpr_write_b32 %" #i ", %16\n\t"
    float t;
    asm volatile(LWM_F4_SC(0) LWM_F4_SC(1) LWM_F4_SC(2) LWM_F4_SC(3) LWM_F4_SC(4) LWM_F4_SC(5) LWM_F4_SC(6) LWM_F4_SC(7)
                 LWM_F4_SC(8) LWM_F4_SC(9) LWM_F4_SC(10) LWM_F4_SC(11) LWM_F4_SC(12) LWM_F4_SC(13) LWM_F4_SC(14) LWM_F4_SC(15) "s_nop 1"
                 : "+a"(d[0]), "+a"(d[1]), "+a"(d[2]), "+a"(d[3]), "+a"(d[4]), "+a"(d[5]), "+a"(d[6]), "+a"(d[7]),
                   "+a"(d[8]), "+a"(d[9]), "+a"(d[10]), "+a"(d[11]), "+a"(d[12]), "+a"(d[13]), "+a"(d[14]), "+a"(d[15]), "=&v"(t)
                 : "v"(alpha));
#undef LWM_F4_SC
}
LWM_DEVICE void f4_max3(float& m, float a, float b) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b)); }
#endif

// ---- the two phases of one HALF tile (32 keys).  sC = the half tile being finished (scores relative to the
// reference), pC = its probabilities (written by phase 1, summed by phase 2), sN = the next half tile's scores.
// Template switches select what a prologue / last step needs.
// K row fragment (d step s) of key half KHALF in the K buffer cx.ka points at; V transposed fragment f = (key step
// f>>2 of 16 keys, d block f&3) of key half VHALF in the V buffer cx.vlo / cx.vup point at
template <int KHALF>
LWM_DEVICE bf16x8 f4_kread(const F4Ctx& cx, int s) { return lds_read_b128(cx.ka[s] + KHALF * 32 * kRowBytes); }
template <int VHALF>
LWM_DEVICE bf16x8 f4_vread(const F4Ctx& cx, int f) {
    const uint32_t off = VHALF * 32 * kRowBytes + 16 * (f >> 2) * kRowBytes;
    bf16x4 lo = lds_read_tr16(cx.vlo[f & 3] + off);
    bf16x4 up = lds_read_tr16(cx.vup[f & 3] + off);
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = up[0]; o[5] = up[1]; o[6] = up[2]; o[7] = up[3];
    return o;
}

// Phase 1 (16 MFMAs): S(next) = K Q^T  ||  p = exp2(t), P -> bf16, pair sums.  t = s*c - m*c of the half tile being
// finished was left in tC by the phase 2 before (so that the two phases carry about the same number of fillers);
// ps receives p[2j] + p[2j+1].  KHALF = key half of the NEXT half tile inside the K buffer cx.ka points at; its
// fragments 0..kF4Ahead-1 are ALREADY in kfr (requested by whoever ran before: with one wave on the SIMD nothing else
// covers the LDS latency at the head of a phase).  VNEXT >= 0: request the first three V fragments of the phase 2 that
// follows (key half VNEXT) during the last gaps.  DMA: which LDS-DMA pieces of `dm` go out in this phase (-1 none).
template <int KHALF, bool DO_S, bool DO_FIN, int VNEXT, int DMA>
LWM_DEVICE void f4_phase1(const F4Ctx& cx, const bf16x8 (&qf)[2][8], f32x16 (&sN)[2], float (&tC)[2][16], float (&ps)[2][8],
                          bf16x8 (&pb)[2][2], bf16x8 (&kfr)[8], bf16x8 (&vfr)[8], const F4Stage& st, const F4Dma& dm) {
    uint32_t w[4];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int s = g >> 1, qb = g & 1;
        if (DO_S && qb == 0 && s + kF4Ahead < 8) kfr[s + kF4Ahead] = f4_kread<KHALF>(cx, s + kF4Ahead);
#pragma unroll
        for (int j = 0; j < kF4Ahead; ++j)
            if (VNEXT >= 0 && g == f4_pre_gap(j)) vfr[j] = f4_vread<VNEXT < 0 ? 0 : VNEXT>(cx, j);
        sched_fence();
        if (DO_S) {
            if (s == 0) f4_mfma_s_first(sN[qb], kfr[0], qf[qb][0]);
            else f4_mfma_s(sN[qb], kfr[s], qf[qb][s]);
        }
        // LDS-DMA pieces of this phase (measured: a piece needs ~1000 cycles from issue to landed under load, so the
        // last one leaves in the first half of the iteration's third phase; what is needed first goes first)
        if (DMA == 0 && g == kF4G0a) f4_dma1(st.voff_k[2], dm.k_up_src, dm.k_up_dst);            // K(i+1), keys 32..63
        if (DMA == 0 && g == kF4G0b) f4_dma1(st.voff_k[3], dm.k_up_src, dm.k_up_dst + 4096);
        if (DMA == 2 && g == kF4G2a) f4_dma1(st.voff_v[3], dm.v_src, dm.v_dst + 12288);         // V(i+1), last piece
        if (DMA == 2 && g == kF4G2b) f4_dma1(st.voff_k[0], dm.k_lo_src, dm.k_lo_dst);            // K(i+2), keys 0..31
        if (DMA == 2 && g == kF4G2c) f4_dma1(st.voff_k[1], dm.k_lo_src, dm.k_lo_dst + 4096);
        if (DO_FIN) {
            // finish-softmax slice: elements 2g, 2g+1 of the finished half tile, flattened [qb][r]; the pair of the
            // PREVIOUS gap is packed and summed here (a transcendental's result is not read by the very next instruction)
            const int e = 2 * g, fq = e >> 4, r = e & 15;
            tC[fq][r] = f4_exp2(tC[fq][r]);
            tC[fq][r + 1] = f4_exp2(tC[fq][r + 1]);
            if (g > 0) {
                const int e0 = 2 * (g - 1), q0 = e0 >> 4, r0 = e0 & 15;
                w[(g - 1) & 3] = f4_cvt_pk(tC[q0][r0], tC[q0][r0 + 1]);
                ps[q0][r0 >> 1] = f4_add2(tC[q0][r0], tC[q0][r0 + 1]);
                if (((g - 1) & 3) == 3) pb[q0][(e0 >> 3) & 1] = __builtin_bit_cast(bf16x8, u32x4{w[0], w[1], w[2], w[3]});
            }
        }
        sched_fence();
    }
    if (DO_FIN) {
        w[3] = f4_cvt_pk(tC[1][14], tC[1][15]);
        ps[1][7] = f4_add2(tC[1][14], tC[1][15]);
        pb[1][1] = __builtin_bit_cast(bf16x8, u32x4{w[0], w[1], w[2], w[3]});
    }
}

// Phase 2 (16 MFMAs): O^T += V^T P^T  ||  row sums (the pair sums ps), running max of the next half tile's scores (sN)
// and its exponents tN = s*c - m*c (one fma per score).
// VHALF = key half of the half tile being finished inside the V buffer; its fragments 0..kF4Ahead-1 are already in vfr.
// KNEXT >= 0: request the first three K fragments of the phase 1 that follows (key half KNEXT of the buffer cx.ka
// points at NOW) during the last gaps.
template <int VHALF, bool DO_PV, bool DO_MAX, int KNEXT, int DMA>
LWM_DEVICE void f4_phase2(F4Ctx& cx, const bf16x8 (&pb)[2][2], f32x16 (&acc)[2][4], const float (&ps)[2][8],
                          const f32x16 (&sN)[2], float (&tN)[2][16], float (&mx)[2], bf16x8 (&kfr)[8], bf16x8 (&vfr)[8],
                          const F4Stage& st, const F4Dma& dm) {
    float ls[2] = {0.f, 0.f};
    float mp[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int h = 0; h < 16; ++h) {
        const int f = h >> 1, qb = h & 1, t = f >> 2, db = f & 3;
        if (DO_PV && qb == 0 && f + kF4Ahead < 8) vfr[f + kF4Ahead] = f4_vread<VHALF>(cx, f + kF4Ahead);
#pragma unroll
        for (int j = 0; j < kF4Ahead; ++j)
            if (KNEXT >= 0 && h == f4_pre_gap(j)) kfr[j] = f4_kread<KNEXT < 0 ? 0 : KNEXT>(cx, j);
        sched_fence();
        if (DO_PV) f4_mfma_o(acc[qb][db], vfr[f], pb[qb][t]);
        if (DMA == 1 && h == kF4G1a) f4_dma1(st.voff_v[0], dm.v_src, dm.v_dst);                  // V(i+1)
        if (DMA == 1 && h == kF4G1b) f4_dma1(st.voff_v[1], dm.v_src, dm.v_dst + 4096);
        if (DMA == 1 && h == kF4G1c) f4_dma1(st.voff_v[2], dm.v_src, dm.v_dst + 8192);
        const int e = 2 * h, fq = e >> 4, r = e & 15;
        if (DO_PV) f4_add(ls[fq], ps[fq][r >> 1]);
        if (DO_MAX) {
            f4_max3(mp[fq], sN[fq][r], sN[fq][r + 1]);
            tN[fq][r] = f4_fma(sN[fq][r], cx.c, cx.nbase[fq]);
            tN[fq][r + 1] = f4_fma(sN[fq][r + 1], cx.c, cx.nbase[fq]);
        }
        sched_fence();
    }
    if (DO_PV)
        for (int q2 = 0; q2 < 2; ++q2) cx.lsum[q2] += ls[q2];
    if (DO_MAX)
        for (int q2 = 0; q2 < 2; ++q2) mx[q2] = mp[q2];
}

// masks of one half tile on its scores (lwm/llama.py:572-592): causal, same segment, key valid.
// rel[qb] = (query position) - (position of the TILE's key 0), clamped to [-1, 64]; key kl of the tile visible iff
// kl <= rel.  KHALF selects keys 32*KHALF .. +31.
template <bool HAS_META, int KHALF>
LWM_DEVICE void f4_mask(const F4Ctx& cx, f32x16 (&sN)[2], const int (&rel)[2], const int32_t (&seg_q)[2], int mbuf) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int relh = rel[qb] - 4 * cx.hi - 32 * KHALF;       // kl = 32*KHALF + 8*g + 4*hi + j <= rel
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (HAS_META) {
                const u32x4 sg = lds_read_u32x4(cx.lds + kF4OffMeta + mbuf * kF4BK * 4 + 16 * cx.hi + (32 * KHALF + 8 * g) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool vis = ((int32_t)sg[j] == seg_q[qb]) && (8 * g + j <= relh);
                    sN[qb][4 * g + j] = vis ? sN[qb][4 * g + j] : -INFINITY;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) sN[qb][4 * g + j] = (8 * g + j <= relh) ? sN[qb][4 * g + j] : -INFINITY;
            }
        }
    }
}

// The rare block: move the reference of the rows that need it.  Runs after the P.V of the half tile in flight, so
// everything accumulated at the old reference is rescaled exactly once.  Scores stay raw in their registers; the
// exponents of the NEXT half tile (tN, already formed against the old reference) are shifted to the new one.
LWM_DEVICE void f4_rescale(F4Ctx& cx, const float (&mx)[2], f32x16 (&acc)[2][4], float (&tN)[2][16]) {
    f4_settle_acc(acc);    // the P.V MFMAs (asm) have written acc: hipcc does not know they are MFMAs
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float mxa = fmaxf(mx[qb], xhalf(mx[qb]));           // both halves of the wave hold keys of the column
        const float m_new = fmaxf(cx.mref[qb], mxa);
        const float ms = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = fast_exp2((cx.mref[qb] - ms) * cx.c); // 0 while the old reference is -inf
        cx.lsum[qb] *= alpha;
        for (int db = 0; db < 4; ++db) f4_scale_acc(acc[qb][db], alpha);
        const float nb_new = -ms * cx.c;
        const float dt = nb_new - cx.nbase[qb];    // change of the exponents
        for (int r = 0; r < 16; ++r) tN[qb][r] += dt;              // (-inf stays -inf)
        cx.mref[qb] = m_new;
        cx.nbase[qb] = nb_new;
        cx.thr[qb] = (m_new == -INFINITY) ? -INFINITY : m_new + cx.thr_on;
    }
}

template <bool HAS_META>
LWM_DEVICE void attn_fwd64_body(const AttnParams& p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = wave_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;

    // ---- block -> (q tile, head, batch): longest q tiles first; all q tiles of one (b,h) on one XCD
    const int nqt = (p.Sq + kF4BQ - 1) / kF4BQ;
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
    const bf16_t* qb_ = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;

    F4Ctx cx;
    cx.lds = lds;
    cx.tid = tid;
    cx.lane = lane;
    cx.hi = hi;
    cx.wave = wave;
    for (int s = 0; s < 8; ++s) cx.ka[s] = lds + tile_off(l31, 2 * s + hi);
    {
        const TrFragAddr t = frag_tr_addr(lds + 2 * kF4TileBytes, lane);
        for (int db = 0; db < 4; ++db) {
            cx.vlo[db] = t.lo[db];
            cx.vup[db] = t.up[db];
        }
    }
    cx.c = p.scale * kLog2e;
    cx.thr_on = kDeferLog2 / cx.c;

    // ---- this lane's two query rows
    bf16x8 qf[2][8];
    int q_row[2];
    bool q_ok[2];
    int32_t seg_q[2];
    int64_t q_pos[2];
    const PosMap km = k_map(p);
    const PosTab kt_ = load_postab(km);                          // (register copies of the tables: prologue only)
    const int64_t q_base = pos_base(load_postab(q_map(p)), qt * kF4BQ);       // position of row r of this workgroup = q_base + r
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        q_row[qb] = qt * kF4BQ + wave * 64 + 32 * qb + l31;
        q_ok[qb] = q_row[qb] < p.Sq;
        q_pos[qb] = q_base + q_row[qb];
        seg_q[qb] = (HAS_META && q_ok[qb] && p.seg_q) ? p.seg_q[(int64_t)b * p.Sq + q_row[qb]] : 0;
        // straight into the accumulator file (a row past Sq re-reads the last row: its results are never stored)
        const int qr = q_ok[qb] ? q_row[qb] : p.Sq - 1;
        for (int s = 0; s < 8; ++s) qf[qb][s] = f4_load_agpr(qb_ + (int64_t)qr * p.q_ss + 16 * s + 8 * hi);
        cx.thr[qb] = -INFINITY;
        cx.mref[qb] = -INFINITY;
        cx.nbase[qb] = 0.0f;
        cx.lsum[qb] = 0.0f;
    }
    // (the Q fragments are waited for together with the first staged tiles below: one memory latency per workgroup, not
    //  two in a row -- at the short walks of a ring shard that is 2-3 us of a 50 us workgroup)
    const int64_t wq_min = q_base + (int64_t)qt * kF4BQ + wave * 64;   // first / last query position of this wave
    const int64_t wq_max = wq_min + 63;

    // ---- kv tile range of the WORKGROUP (causal: skip tiles wholly in the future of its last row)
    const int nkt_all = (p.Sk + kF4BK - 1) / kF4BK;
    int nkt = nkt_all, kt0 = 0;
    const int q_last = (qt * kF4BQ + kF4BQ < p.Sq ? qt * kF4BQ + kF4BQ : p.Sq) - 1;
    if (p.causal) nkt = tiles_reaching(kt_, kF4BK, nkt_all, q_base + q_last);      // up to the tile of the last visible key
    if (HAS_META && p.segb_q && p.segb_k && nkt > 0) {      // packed sequences: skip other documents' key tiles
        const int nbq = (p.Sq + 31) >> 5, nbk = (p.Sk + 31) >> 5;
        int smin, smax, lo, hi2;
        seg_own_range(p.segb_q + (int64_t)b * nbq * 2, nbq, qt * (kF4BQ / 32), kF4BQ / 32, smin, smax);
        seg_narrow<kF4Threads>(p.segb_k + (int64_t)b * nbk * 2, nbk, kF4BK / 32, 0, nkt, smin, smax, lds + kF4OffScan, tid, lo, hi2);
        // The walk starts ONE tile before the first tile that can meet the workgroup's documents.  The first half tile of
        // a walk runs through a separate instruction stream (there is no P.V to put beside it), and on MI355X that stream
        // and the steady-state stream round a row's running reference differently in the last bit (same source arithmetic,
        // identical in the host emulation; 10 of 4.2M outputs one bf16 ulp apart at S = 8192 -- profiles/r03_fwd64.md).
        // With the extra, fully masked tile a narrowed walk executes the SAME stream on the same tiles as the full walk:
        // hints change the time, never a bit of the result (tests/test_gpu_attention.py::test_packed_documents_skip_is_exact).
        kt0 = lo > 0 ? lo - 1 : 0;
        nkt = hi2;
    }
    const int n_wg = nkt > kt0 ? nkt - kt0 : 0;
    // tiles this WAVE computes: [kt0, kt0 + n_w) -- its rows see nothing beyond its own diagonal tile
    int n_w = n_wg;
    if (p.causal) {
        const int nw = tiles_reaching(kt_, kF4BK, nkt_all, wq_max) - kt0;
        n_w = nw < 0 ? 0 : (nw < n_wg ? nw : n_wg);
    }
    if (wave_uniform(qt * kF4BQ + wave * 64 >= p.Sq ? 1 : 0)) n_w = 0;    // a wave past the ragged end of Sq

    f32x16 acc[2][4];
    for (int qb = 0; qb < 2; ++qb)
        for (int i = 0; i < 4; ++i) acc[qb][i] = zero_f32x16();

    if (n_wg > 0) {
        F4Stage st;
        st.kb = (const char*)(p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh);
        st.vb = (const char*)(p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh);
        st.ktile_bytes = (int64_t)kF4BK * p.k_ss * 2;
        st.vtile_bytes = (int64_t)kF4BK * p.v_ss * 2;
        const bool ragged = (p.Sk % kF4BK) != 0;      // (then HAS_META is true: the tail rows are masked via key meta)
        f4_stage_offsets(p, wave, lane, 0, st);       // full tiles: nothing is clamped
        // Tile `rel` (relative to kt0) lives in LDS buffer rel & 1; K travels in HALF tiles (keys 0..31 / 32..63 =
        // the wave's pieces j = 0,1 / 2,3), V in whole tiles.  The ragged last tile of the K/V block clamps its rows
        // (its own offsets, recomputed: once per workgroup at most).
        const lds_t kdst0 = lds + (uint32_t)wave * 1024, vdst0 = lds + 2 * kF4TileBytes + (uint32_t)wave * 1024;
        auto stage_k = [&](int rel, int half) {
            const int kt = kt0 + rel;
            const lds_t dst = kdst0 + (rel & 1) * kF4TileBytes + half * 8192;
            if (ragged && kt == nkt_all - 1) {
                F4Stage s2 = st;
                f4_stage_offsets(p, wave, lane, kt, s2);
                f4_dma<2>(s2.voff_k + 2 * half, st.kb + kt * st.ktile_bytes, dst);
            } else {
                f4_dma<2>(st.voff_k + 2 * half, st.kb + kt * st.ktile_bytes, dst);
            }
        };
        auto stage_v = [&](int rel) {
            const int kt = kt0 + rel;
            const lds_t dst = vdst0 + (rel & 1) * kF4TileBytes;
            if (ragged && kt == nkt_all - 1) {
                F4Stage s2 = st;
                f4_stage_offsets(p, wave, lane, kt, s2);
                f4_dma<4>(s2.voff_v, st.vb + kt * st.vtile_bytes, dst);
            } else {
                f4_dma<4>(st.voff_v, st.vb + kt * st.vtile_bytes, dst);
            }
        };
        // the staging of tile iteration i: K(i+2)[0..31] -> K buffer i&1 (lower half: last read by the S of half tile
        // 2i, one iteration ago), K(i+1)[32..63] -> K buffer (i+1)&1 (upper half: last read one iteration ago),
        // V(i+1) -> V buffer (i+1)&1; key meta of tile i+2
        auto stage_iter = [&](int i) {
            if (i + 2 < n_wg) stage_k(i + 2, 0);
            if (i + 1 < n_wg) {
                stage_k(i + 1, 1);
                stage_v(i + 1);
            }
            if (HAS_META && i + 2 < n_wg) f4_meta_stage(p, cx, b, kt0 + i + 2, (i + 2) % 3);
        };
        PosCursor kc = cursor_begin(kt_);
        auto rel_of = [&](int rel, int (&r)[2]) {       // mask offsets of tile `rel` for the two query blocks
            const int krow0 = (kt0 + rel) * kF4BK;
            cursor_seek(km, kc, krow0);                 // (the walk ascends: a compare while the tile is inside the piece)
            const int64_t k_pos0 = kc.base + krow0;
            for (int qb = 0; qb < 2; ++qb) {
                const int64_t d = p.causal ? (q_pos[qb] - k_pos0) : (int64_t)kF4BK;
                r[qb] = d > kF4BK ? kF4BK : (d < -1 ? -1 : (int)d);
            }
        };
        // a tile needs the mask code when its last key lies beyond the wave's first query (or keys can be masked for
        // another reason than causality): from tile first_mask on -- one integer compare per half step in the loop
        int first_mask = 0x7fffffff;
        if (p.causal) {
            const int ft = tiles_below(kt_, kF4BK, nkt_all, wq_min) - kt0;     // the tiles before it lie wholly at or below wq_min
            first_mask = ft < 0 ? 0 : ft;
        }
        auto needs_causal = [&](int rel) -> bool { return rel >= first_mask; };
        // Packed sequences: the wave's 64 queries of one segment?  Then a key tile whose 64 staged segment words all carry
        // it needs no segment test (attn_common.h, seg_step_uniform); rows past Sq are never stored and do not count.
        const int32_t own_seg = wave_uniform(seg_q[0]);
        const bool own_uniform = HAS_META && !wave_any((q_ok[0] && seg_q[0] != own_seg) || (q_ok[1] && seg_q[1] != own_seg));
        auto meta_words = [&](int mbuf) -> lds_t { return cx.lds + kF4OffMeta + mbuf * kF4BK * 4; };
        bool uni_c = false;      // the tile whose masks come next
#define LWM_F4_MASK(KHALF_, s_, rel_, mbuf_, uni_, SETTLE_)                                                          \
    do {                                                                                                            \
        if ((HAS_META && !(uni_)) || needs_causal(rel_)) {                                                          \
            int r_[2];                                                                                              \
            if (SETTLE_) f4_settle_s(s_);      /* first: rel_of may branch, and hipcc copies tuples at merges */       \
            rel_of(rel_, r_);                                                                                       \
            if (HAS_META && !(uni_)) f4_mask<HAS_META, KHALF_>(cx, s_, r_, seg_q, mbuf_);                           \
            else f4_mask<false, KHALF_>(cx, s_, r_, seg_q, mbuf_);                                                  \
        }                                                                                                           \
    } while (0)

        // ---- prologue: K(0), V(0) and the first half of K(1) in flight -> S of half tile 0, its masks, its reference
        stage_k(0, 0);
        stage_k(0, 1);
        stage_v(0);
        if (n_wg > 1) stage_k(1, 0);
        if (HAS_META) {
            f4_meta_stage(p, cx, b, kt0, 0);
            if (n_wg > 1) f4_meta_stage(p, cx, b, kt0 + 1, 1);
        }
        f4_load_agpr_wait(qf);      // vmcnt(0): Q fragments AND the tiles just requested
        block_sync();

        f32x16 sA[2], sB[2];
        float tt[2][16], ps[2][8];
        bf16x8 pb[2][2];
        float mx[2];
        for (int qb = 0; qb < 2; ++qb) {
            sA[qb] = zero_f32x16();
            sB[qb] = zero_f32x16();
            for (int r = 0; r < 16; ++r) tt[qb][r] = 0.0f;
            for (int r = 0; r < 8; ++r) ps[qb][r] = 0.0f;
            for (int t = 0; t < 2; ++t) pb[qb][t] = zero_bf16x8();
        }
        bf16x8 kfr[8], vfr[8];
        for (int j = 0; j < 8; ++j) {
            kfr[j] = zero_bf16x8();
            vfr[j] = zero_bf16x8();
        }
        F4Dma dm = {};
        if (n_w > 0) {
            for (int j = 0; j < kF4Ahead; ++j) kfr[j] = f4_kread<0>(cx, j);
            f4_phase1<0, true, false, -1, -1>(cx, qf, sA, tt, ps, pb, kfr, vfr, st, dm);
            f4_settle_s(sA);       // (no P.V MFMA follows here: the max / exponent fillers read the scores at once)
            uni_c = HAS_META && seg_step_uniform(seg_step_word(meta_words(0), lane), own_uniform, own_seg);
            LWM_F4_MASK(0, sA, 0, 0, uni_c, false);
            f4_phase2<0, false, true, -1, -1>(cx, pb, acc, ps, sA, tt, mx, kfr, vfr, st, dm);
            if (wave_any(mx[0] > cx.thr[0] || mx[1] > cx.thr[1])) f4_rescale(cx, mx, acc, tt);
        }
        block_sync();      // every wave has read K(0)[0..31] before K(2)[0..31] may land on it

        // Tile iteration i.  The fragment addresses in cx point at the buffers of tile i and are toggled to the other
        // buffer as the iteration goes (K between the half steps, V at the end): ONE loop body, no unrolling.
        //   half 0   phase 1: S(2i+1) from K(i)[32..63] || finish(2i);     phase 2: P.V(2i)   from V(i)[0..31]  || sums, max / exponents(2i+1)
        //   half 1   phase 1: S(2i+2) from K(i+1)[0..31] || finish(2i+1);  phase 2: P.V(2i+1) from V(i)[32..63] || sums, max / exponents(2i+2)
        //   bottom   wait for the DMA, one barrier.
        // Every phase but the first of an iteration finds its first fragments already requested by its predecessor;
        // in the fast loop the eight LDS-DMA pieces of the tiles ahead go out two per phase, between MFMAs.
        int32_t ktog = kF4TileBytes, vtog = kF4TileBytes;     // +16 KiB now, -16 KiB next time
        auto toggle_k = [&]() {
            for (int s = 0; s < 8; ++s) cx.ka[s] += (uint32_t)ktog;
            ktog = -ktog;
        };
        auto toggle_v = [&]() {
            for (int db = 0; db < 4; ++db) {
                cx.vlo[db] += (uint32_t)vtog;
                cx.vup[db] += (uint32_t)vtog;
            }
            vtog = -vtog;
        };
        // (-DLWM_PROF builds, scripts/micro/fused_bench with LWM_PROF_DUMP=1: s_memtime laps of the tile loop --
        // 0 phase 1 of half 0 (with the wait for its first K fragments), 1 mask + toggle, 2 phase 2 + rescale test,
        // 3 phase 1 of half 1, 4 mask, 5 phase 2 + rescale test, 6 DMA wait, 7 barrier, 8 iterations)
#ifdef LWM_PROF
        unsigned long long f4p[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, f4t = __builtin_amdgcn_s_memtime();
        const unsigned long long f4t0 = f4t;
#define F4_LAP(slot)                                                  \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        f4p[slot] += now_ - f4t;                                      \
        f4t = now_;                                                   \
    } while (0)
#else
#define F4_LAP(slot)
#endif
#define LWM_F4_TILE(i, D0, D1, D2, D3)                                                                             \
    do {                                                                                                            \
        const int32_t segw_ = HAS_META ? seg_step_word(meta_words(((i) + 1) % 3), lane) : 0;     /* the next tile's */  \
        f4_phase1<1, true, true, 0, D0>(cx, qf, sB, tt, ps, pb, kfr, vfr, st, dm);   /* S(2i+1) || finish(2i) */   \
        F4_LAP(0);                                                                                                  \
        LWM_F4_MASK(1, sB, i, (i) % 3, uni_c, true);                                                                \
        uni_c = HAS_META && seg_step_uniform(segw_, own_uniform, own_seg);                                          \
        toggle_k();                                                                                                 \
        F4_LAP(1);                                                                                                  \
        f4_phase2<0, true, true, 0, D1>(cx, pb, acc, ps, sB, tt, mx, kfr, vfr, st, dm);   /* P.V(2i) */            \
        if (wave_any(mx[0] > cx.thr[0] || mx[1] > cx.thr[1])) f4_rescale(cx, mx, acc, tt);                          \
        F4_LAP(2);                                                                                                  \
        f4_phase1<0, true, true, 1, D2>(cx, qf, sA, tt, ps, pb, kfr, vfr, st, dm);   /* S(2i+2) || finish(2i+1) */ \
        F4_LAP(3);                                                                                                  \
        LWM_F4_MASK(0, sA, (i) + 1, ((i) + 1) % 3, uni_c, true);                                                    \
        F4_LAP(4);                                                                                                  \
        f4_phase2<1, true, true, -1, D3>(cx, pb, acc, ps, sA, tt, mx, kfr, vfr, st, dm);   /* P.V(2i+1) */         \
        if (wave_any(mx[0] > cx.thr[0] || mx[1] > cx.thr[1])) f4_rescale(cx, mx, acc, tt);                          \
        toggle_v();                                                                                                 \
        F4_LAP(5);                                                                                                  \
        glds_wait_all();                                                                                            \
        F4_LAP(6);                                                                                                  \
        block_sync();                                                                                               \
        F4_LAP(7);                                                                                                  \
    } while (0)
        // the wave's own tiles but the last: both half steps have a successor
        const int n_hot = n_w > 0 ? n_w - 1 : 0;
        // ... of which those whose staging needs no decision: tiles i+1 and i+2 exist and neither is the ragged one
        int n_fast = n_hot < n_wg - 2 ? n_hot : n_wg - 2;
        if (ragged && n_fast > nkt_all - 1 - kt0 - 2) n_fast = nkt_all - 1 - kt0 - 2;
        if (n_fast < 0) n_fast = 0;
        int i = 0;
        for (; i < n_fast; ++i) {
            for (int j = 0; j < kF4Ahead; ++j) kfr[j] = f4_kread<1>(cx, j);
            dm.k_lo_src = st.kb + (kt0 + i + 2) * st.ktile_bytes;
            dm.k_up_src = st.kb + (kt0 + i + 1) * st.ktile_bytes;
            dm.v_src = st.vb + (kt0 + i + 1) * st.vtile_bytes;
            dm.k_lo_dst = kdst0 + (i & 1) * kF4TileBytes;
            dm.k_up_dst = kdst0 + ((i + 1) & 1) * kF4TileBytes + 8192;
            dm.v_dst = vdst0 + ((i + 1) & 1) * kF4TileBytes;
            if (HAS_META) f4_meta_stage(p, cx, b, kt0 + i + 2, (i + 2) % 3);
            LWM_F4_TILE(i, 0, 1, 2, -1);
#ifdef LWM_PROF
            f4p[8] += 1;
#endif
        }
#ifdef LWM_PROF
        if (!HAS_META && hb == 0 && qt == nqt - 1 && lane == 0 && p.out_acc) {      // the longest q tile of head 0
            f4p[9] = __builtin_amdgcn_s_memtime() - f4t0;
            for (int j = 0; j < 10; ++j) ((unsigned long long*)p.out_acc)[wave * 10 + j] = f4p[j];
        }
#endif
        for (; i < n_hot; ++i) {
            for (int j = 0; j < kF4Ahead; ++j) kfr[j] = f4_kread<1>(cx, j);
            stage_iter(i);
            LWM_F4_TILE(i, -1, -1, -1, -1);
        }
#undef LWM_F4_TILE
#undef F4_LAP
        // its last tile: the second half step has no successor
        if (n_w > 0) {
            for (int j = 0; j < kF4Ahead; ++j) kfr[j] = f4_kread<1>(cx, j);
            stage_iter(i);
            f4_phase1<1, true, true, 0, -1>(cx, qf, sB, tt, ps, pb, kfr, vfr, st, dm);
            LWM_F4_MASK(1, sB, i, i % 3, uni_c, true);
#undef LWM_F4_MASK
            f4_phase2<0, true, true, -1, -1>(cx, pb, acc, ps, sB, tt, mx, kfr, vfr, st, dm);
            if (wave_any(mx[0] > cx.thr[0] || mx[1] > cx.thr[1])) f4_rescale(cx, mx, acc, tt);
            f4_phase1<0, false, true, 1, -1>(cx, qf, sA, tt, ps, pb, kfr, vfr, st, dm);
            f4_phase2<1, true, false, -1, -1>(cx, pb, acc, ps, sA, tt, mx, kfr, vfr, st, dm);
            glds_wait_all();
            block_sync();
            ++i;
        }
        // tiles of the workgroup beyond this wave's diagonal: staging only
        for (; i < n_wg; ++i) {
            stage_iter(i);
            glds_wait_all();
            block_sync();
        }
    }

    if (n_wg <= 0) f4_load_agpr_wait(qf);      // (no walk: the fragments are unused, their loads are not left in flight)
    // ---- epilogue: normalise, merge with the ring carry, store (per query block)
    f4_settle_acc(acc);
    const bool carry = p.carry_in != 0, fin = p.final_out != 0;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = cx.lsum[qb] + xhalf(cx.lsum[qb]);
        float inv = 0.0f, lse_b = -INFINITY;
        if (l_tot > 0.0f) {
            inv = 1.0f / l_tot;
            lse_b = cx.mref[qb] * p.scale + logf(l_tot);      // (the reference is a raw q.k)
        }
        float w_a = 0.0f, w_b = 1.0f, lse_new = lse_b;
        const int64_t lse_idx = ((int64_t)b * p.H + h) * p.Sq + q_row[qb];
        if (carry && q_ok[qb]) {
            const float lse_a = p.lse_acc[lse_idx];
            const float mxl = fmaxf(lse_a, lse_b);
            if (mxl == -INFINITY) {
                lse_new = -INFINITY;
                w_a = 0.0f;
                w_b = 0.0f;
            } else {
                const float ea = expf(lse_a - mxl), eb = expf(lse_b - mxl);
                lse_new = mxl + logf(ea + eb);
                w_a = ea / (ea + eb);
                w_b = eb / (ea + eb);
            }
        }
        if (q_ok[qb]) {
            // (what the 16 stores share is read once and the common case is its own loop: left inside one loop, hipcc
            // re-read carry_in / final_out behind a wait per store -- attn_bwd64.h, d4_store_tiles)
            const float sc = inv * w_b;
            bf16_t* const op = p.out + (int64_t)b * p.o_sb + (int64_t)q_row[qb] * p.o_ss + (int64_t)h * p.o_sh + 4 * hi;
            float* const ap = p.out_acc + (((int64_t)b * p.Sq + q_row[qb]) * p.H + h) * kHeadDim + 4 * hi;   // the f32 carry is dense [B,Sq,H,D]
            if (fin && !carry) {
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq)
                        global_store_b64(op + 32 * db + 8 * rq,
                                         u32x2{pack_bf16x2(acc[qb][db][4 * rq + 0] * sc, acc[qb][db][4 * rq + 1] * sc),
                                               pack_bf16x2(acc[qb][db][4 * rq + 2] * sc, acc[qb][db][4 * rq + 3] * sc)});
            } else {
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int d0 = 32 * db + 8 * rq;
                        float o0 = acc[qb][db][4 * rq + 0] * sc, o1 = acc[qb][db][4 * rq + 1] * sc;
                        float o2 = acc[qb][db][4 * rq + 2] * sc, o3 = acc[qb][db][4 * rq + 3] * sc;
                        if (carry) {
                            const f32x4 a = global_load_f32x4(ap + d0);
                            o0 += a[0] * w_a; o1 += a[1] * w_a; o2 += a[2] * w_a; o3 += a[3] * w_a;
                        }
                        if (fin) global_store_b64(op + d0, u32x2{pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)});
                        else global_store_f32x4(ap + d0, f32x4{o0, o1, o2, o3});
                    }
            }
            if (hi == 0) {
                if (fin) p.lse[lse_idx] = lse_new;
                else p.lse_acc[lse_idx] = lse_new;
            }
        }
    }
}

LWM_KERNEL(kF4Threads) void attn_fwd64_kernel(AttnParams p) { attn_fwd64_body<false>(p); }
LWM_KERNEL(kF4Threads) void attn_fwd64_meta_kernel(AttnParams p) { attn_fwd64_body<true>(p); }

}  // namespace lwm
