// attn_bwd64.h -- blockwise attention backward, ONE WAVE PER SIMD (the structure of attn_fwd64.h): the dK/dV kernel
// and (second half of the file) the dQ kernel of the two-kernel backward for one (q block, kv block) ring step on
// gfx950.  Requires wave_ops.h, attn_common.h, attn_fwd.h, attn_fwd64.h (f4_* instruction helpers) and attn_bwd.h (the
// delta kernel, the layout of the row statistics).
//
// dK/dV: replaces the dk / dv part of the custom-VJP backward of `ringattention` (call site lwm/llama.py:539-569;
// SURVEY.md Appendix A.1): p from the saved LSE, dv += p^T do, dp = do v^T, ds = p * (dp - rowsum(do * o)),
// dk += ds^T q * scale.
//
// Workgroup = 4 waves = 128 keys; a wave owns 32 keys and the SIMD's whole register file: its K and V fragments (B
// operands, 32 + 32 registers) and the f32 dK^T / dV^T accumulators (64 + 64) live in the accumulator file -- the
// forward's budget of 192 AGPRs -- so the only LDS traffic of the tile loop is the streamed operand: Q / dO row
// fragments for S and dP, Q^T / dO^T transposed fragments for dK and dV, one fragment per MFMA, every address
// register-resident (no v_xor).  Q / dO arrive by LDS-DMA in steps of 64 queries through a ring of four 32-KiB
// slots: the pieces of step i+2 are issued in the first phase of step i and have landed, for every wave, at the
// barrier that ends step i -- a whole step before their first reader, which is what lets the last phase of step
// i+1 request the first fragments of step i+2 across the barrier.  The row statistics (attn_bwd.h: -lse * log2 e,
// -delta, padded) travel the same way, one dword per lane.
//
// A unit = 32 queries x the wave's 32 keys = two phases of 16 MFMAs, written as `asm volatile` statements in program
// order (attn_fwd64.h explains why), with the vector work placed in the gaps:
//
//   phase   matrix pipe (consecutive MFMAs never share an accumulator)      vector pipe (same wave)
//   X(u)    S(u) = Q(u) K^T  and  dP'(u) = dO(u) V^T - delta, alternating   P(u-1) -> bf16, dS(u-1) = p dP'(u-1) -> bf16
//   Y(u)    dV^T += dO(u-1)^T P(u-1), dK^T += Q(u-1)^T dS(u-1), 8 tuples    t = S(u) c - lse2, p(u) = exp2(t)
//
// The products lag one unit, so every phase has fillers -- 32 VALU per 16 MFMAs -- and S needs one register tile (dP'
// two: the multiply of unit u-1 runs beside the chain of unit u).  -delta enters as the initial value of the dP
// accumulator (the wave that carries the statistics negates it on its way to LDS; the unit before requests it into
// the tuple), so the chain leaves dP' = dP - delta and dS is ONE multiply: 64 VALU per 32 MFMAs.
//
// LDS map: slot 0..3 = [Q tile 64 rows | dO tile 64 rows] (16 KiB each) | stats 0..3 = [nl2 64 | -delta 64 | seg_q 64]
#pragma once

namespace lwm {

constexpr int kD4BK = 128;      // keys per workgroup
constexpr int kD4BQ = 64;       // queries per step (two units of 32)
constexpr int kD4Threads = 256;
constexpr int kD4Slots = 4;
constexpr int kD4TileBytes = kD4BQ * kRowBytes;          // 16 KiB
constexpr int kD4SlotBytes = 2 * kD4TileBytes;           // Q | dO
constexpr int kD4OffStat = kD4Slots * kD4SlotBytes;      // 128 KiB
constexpr int kD4StatBytes = 3 * kD4BQ * 4;              // nl2 | delta | seg_q
constexpr int kD4LdsBytes = kD4OffStat + kD4Slots * kD4StatBytes;
// LDS fragments are requested kD4Ahead MFMAs before the MFMA that consumes them, through ONE register ring of eight
// that all four phases share (fragment m of a unit lives in ring[m % 8]).
#ifndef LWM_D4_AHEAD
#define LWM_D4_AHEAD 6
#endif
constexpr int kD4Ahead = LWM_D4_AHEAD;
static_assert(kD4Ahead >= 1 && kD4Ahead <= 7, "prefetch distance in fragments");
// Fragments are requested and consumed in PAIRS (the two chains of X, the two products of Y): both requests go out in
// the even gap, and the pair's YOUNGER fragment is consumed first -- the s_waitcnt hipcc puts in front of that MFMA
// covers the older one too, so there is one wait per two MFMAs.
static_assert((kD4Ahead % 2) == 0, "paired requests need an even distance");

struct D4Ctx {
    uint32_t qa[8];             // Q row-fragment addresses (d step s), rows 0..31 of the CURRENT step's Q tile
    uint32_t tlo[4], tup[4];    // transposed-fragment addresses (d block), rows 0..15 of the current step's Q tile
    uint32_t plo[4], pup[4];    // the same in the PREVIOUS step's slot (the lagging dK of a step's first unit)
    uint32_t stat;              // this step's statistics + 16 * hi
    float c;                    // scale * log2(e)
    int hi;
};

#ifdef LWM_EMU
LWM_DEVICE void d4_mfma_p_first(f32x16& d, bf16x8 a, bf16x8 b) { d = mfma_32x32x16(a, b, d); }
LWM_DEVICE void d4_mfma_o_first(f32x16& d, bf16x8 a, bf16x8 b) { d = mfma_32x32x16(a, b, zero_f32x16()); }
LWM_DEVICE float d4_mul(float a, float b) { return a * b; }
LWM_DEVICE void d4_dma_b32(uint32_t voff, const char* src, lds_t dst) { glds_load_b32(src + voff, dst); }
LWM_DEVICE void d4_settle_t(f32x16&) {}
LWM_DEVICE void d4_settle_acc(f32x16 (&)[4], f32x16 (&)[4]) {}
LWM_DEVICE void d4_settle_acc4(f32x16 (&)[4]) {}
#else
// What hipcc does not know about an asm MFMA (cdna_hip_programming.md section 5.7 item 2): a register it has just
// written itself -- a copy that moves a tuple into place, a zero it materialises late -- needs two wait states before
// an MFMA reads it.  The two statements below are used where that can happen:
//   * the first MFMA of a dP chain: its accumulator was filled with -delta by compiler-visible LDS loads, which hipcc
//     may route through copies (it does in the prologue): the wait states stand in front of the MFMA;
//   * the first product of the walk into each dK^T / dV^T tuple: C = 0 and an OUTPUT-only operand, so that no
//     compiler-made zero is read at all.
LWM_DEVICE void d4_mfma_p_first(f32x16& d, bf16x8 a, bf16x8 b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
LWM_DEVICE void d4_mfma_o_first(f32x16& d, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(d) : "v"(a), "v"(b));
}
LWM_DEVICE float d4_mul(float a, float b) {
    float y;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b));
    return y;
}
// one LDS-DMA dword per lane (64 floats of row statistics), M0 written bare as f4_dma1 does
LWM_DEVICE void d4_dma_b32(uint32_t voff, const char* src, lds_t dst) {
    const uint64_t a = (uint64_t)src;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(u), "s"(dst) : "memory");
}
LWM_DEVICE void d4_settle_t(f32x16& s) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(s)); }
LWM_DEVICE void d4_settle_acc(f32x16 (&a)[4], f32x16 (&b)[4]) {
    asm volatile("s_nop 7\n\ts_nop 7"
                 : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(b[0]), "+a"(b[1]), "+a"(b[2]), "+a"(b[3]));
}
LWM_DEVICE void d4_settle_acc4(f32x16 (&a)[4]) { asm volatile("s_nop 7\n\ts_nop 7" : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3])); }
#endif

// per-lane byte offsets of the wave's four pieces of a 64-row tile (piece = 4 rows = 1 KiB; wave w moves pieces
// w, w+4, w+8, w+12: rows 4w + 16j + (l>>4)) relative to the step's first row, rows clamped to Sq-1 (rows past Sq
// get nl2 = -inf through the statistics).  Lane l writes physical slot l&15 of its row, so it fetches logical slot
// (l&15) ^ swz(row) -- the XOR swizzle applied on the SOURCE column.
LWM_DEVICE void d4_stage_offsets(const AttnParams& p, int wave, int lane, int st, uint32_t (&vq)[4], uint32_t (&vdo)[4]) {
    const int slot = lane & 15;
    for (int j = 0; j < 4; ++j) {
        const int row = 4 * (wave + 4 * j) + (lane >> 4);
        int qrow = st * kD4BQ + row;
        qrow = qrow < p.Sq ? qrow : p.Sq - 1;
        const int rel = qrow - st * kD4BQ;
        const int col = (slot ^ swz(row)) << 3;
        vq[j] = (uint32_t)(((int64_t)rel * p.q_ss + col) * 2);
        vdo[j] = (uint32_t)(((int64_t)rel * p.do_ss + col) * 2);
    }
}

// fragment f of a unit (f = MFMA index 0..31; 32.. = the first fragments of the unit that follows):
//   0..15   phase X: d step f>>1; even = Q row fragment (S), odd = dO row fragment (dP)
//   16..31  phase Y: j = f-16, (q step (j>>3)&1, d block (j>>1)&3) of the PREVIOUS unit; even = dO^T (dV), odd = Q^T (dK)
// cx.qa points at the step's slot for HALF = 0 and for the X phase of HALF = 1; it has moved on to the next step's slot
// when the Y phase of HALF = 1 requests f >= 32.
template <int HALF>
LWM_DEVICE bf16x8 d4_frag(const D4Ctx& cx, int f) {
    if (f >= 32) {
        const int g = f - 32;
        return lds_read_b128(cx.qa[g >> 1] + (g & 1) * kD4TileBytes + (HALF == 0 ? 32 * kRowBytes : 0));
    }
    if (f < 16) return lds_read_b128(cx.qa[f >> 1] + (f & 1) * kD4TileBytes + HALF * 32 * kRowBytes);
    const int j = f - 16, t = (j >> 3) & 1, db = (j >> 1) & 3;
    // the previous unit: the second half of the previous step's tiles for HALF = 0, the first half of this step's
    const uint32_t alo = HALF == 0 ? cx.plo[db] : cx.tlo[db];
    const uint32_t aup = HALF == 0 ? cx.pup[db] : cx.tup[db];
    const uint32_t off = ((j & 1) ? 0 : kD4TileBytes) + (HALF == 0 ? 32 * kRowBytes : 0) + 16 * t * kRowBytes;
    bf16x4 lo = lds_read_tr16(alo + off);
    bf16x4 up = lds_read_tr16(aup + off);
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = up[0]; o[5] = up[1]; o[6] = up[2]; o[7] = up[3];
    return o;
}

// ---- where the vector work of a unit u goes.  Gaps are numbered over the two phases that host it: G = 0..15 = the
// gaps of Y(u) (its S and dP' tiles are complete), G = 16..31 = the gaps of X(u+1).  Per element e (16 per lane) /
// per pair i (8 per lane):
//   F(e)  t = S c - lse2          needs S(u): G >= 1 (the chains have left the pipe), G <= 15 (X(u+1) overwrites S)
//   E(e)  p = exp2(t)             >= 1 gap behind F(e) (and a transcendental's result is not read in the next slot)
//   M(e)  dS = p dP'              >= 1 gap behind E(e)
//   P(i)  P words -> bf16         >= 1 gap behind E(2i), E(2i+1); the products of Y(u) still read the OLD P: word i of
//   D(i)  dS words -> bf16           the first half (i < 4) from G = 8, of the second from G = 16; D(i) likewise, >= 1 gap
//                                    behind M(2i), M(2i+1); everything done by G = 29 (Y(u+1) reads them at G = 32)
// d4_sched_ok() checks the table at compile time.
struct D4Sched {
    int F[16], E[16], M[16], P[8], D[8];
};
constexpr bool d4_sched_ok(const D4Sched& c) {
    for (int e = 0; e < 16; ++e) {
        if (c.F[e] < 1 || c.F[e] > 15) return false;
        if (c.E[e] <= c.F[e] || c.M[e] <= c.E[e] || c.M[e] > 29) return false;
    }
    for (int i = 0; i < 8; ++i) {
        const int lo = i < 4 ? 8 : 16;
        if (c.P[i] < lo || c.P[i] > 29 || c.P[i] <= c.E[2 * i] || c.P[i] <= c.E[2 * i + 1]) return false;
        if (c.D[i] < lo || c.D[i] > 29 || c.D[i] <= c.M[2 * i] || c.D[i] <= c.M[2 * i + 1]) return false;
    }
    return true;
}
// exponentials in Y (beside the transposed reads), multiplies and packs in X (three other placements measured within
// the run-to-run noise: profiles/r04_backward.md)
constexpr D4Sched kD4Sched = {
    /* F */ {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8},
    /* E */ {2, 3, 4, 4, 5, 6, 7, 7, 8, 9, 9, 10, 10, 11, 11, 12},
    /* M */ {16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 21, 21, 22, 22, 23, 23},
    /* P */ {16, 17, 18, 19, 20, 21, 22, 23},
    /* D */ {24, 25, 26, 27, 28, 29, 29, 29}};
static_assert(d4_sched_ok(kD4Sched), "filler schedule violates a dependency");

// the LDS-DMA pieces one step issues (all wave-uniform but the offsets): Q and dO of the step two ahead
struct D4Dma {
    const char* q_src;
    const char* do_src;
    lds_t dst;          // the slot's Q tile + wave * 1024
};

// Live state of a wave across units.
struct D4Regs {
    f32x16 s;               // S tile (MFMA result, read-only for the vector pipe)
    f32x16 dp[2];           // dP' tiles by unit parity: preloaded with -delta (in C/D register order), then the dP chain
    float nl[16];           // -lse * log2(e) of the unit's rows, C/D register order
    float t[16];            // exponents, then p
    float ds[16];           // dS = p * dP'
    bf16x8 pb[2], dsb[2];   // P / dS as B operands (q steps of 16)
    uint32_t pw[8], dw[8];  // their words as they are packed (a half is handed over when its fourth word is in)
    bf16x8 fr[8];           // the fragment ring
};

// the vector work scheduled in gap G; dpu = the dP' tile of the unit being finished
template <int G>
LWM_DEVICE void d4_fillers(const D4Ctx& cx, D4Regs& rg, const f32x16& dpu) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (kD4Sched.F[e] == G) rg.t[e] = f4_fma(rg.s[e], cx.c, rg.nl[e]);
        if (kD4Sched.E[e] == G) rg.t[e] = f4_exp2(rg.t[e]);
        if (kD4Sched.M[e] == G) rg.ds[e] = d4_mul(rg.t[e], dpu[e]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (kD4Sched.P[i] == G) rg.pw[i] = f4_cvt_pk(rg.t[2 * i], rg.t[2 * i + 1]);
        if (kD4Sched.D[i] == G) rg.dw[i] = f4_cvt_pk(rg.ds[2 * i], rg.ds[2 * i + 1]);
    }
    // hand a half over behind its last word
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int lp = 0, ld = 0;
        for (int i = 4 * h; i < 4 * h + 4; ++i) {
            lp = kD4Sched.P[i] > lp ? kD4Sched.P[i] : lp;
            ld = kD4Sched.D[i] > ld ? kD4Sched.D[i] : ld;
        }
        if (lp == G) rg.pb[h] = __builtin_bit_cast(bf16x8, u32x4{rg.pw[4 * h], rg.pw[4 * h + 1], rg.pw[4 * h + 2], rg.pw[4 * h + 3]});
        if (ld == G) rg.dsb[h] = __builtin_bit_cast(bf16x8, u32x4{rg.dw[4 * h], rg.dw[4 * h + 1], rg.dw[4 * h + 2], rg.dw[4 * h + 3]});
    }
}
// dispatch a loop index to the compile-time gap (the loops are fully unrolled: the chain folds to one call)
template <int G0, int N>
LWM_DEVICE void d4_fill_at(int g, const D4Ctx& cx, D4Regs& rg, const f32x16& dpu) {
    if constexpr (N > 0) {
        if (g == G0) d4_fillers<G0>(cx, rg, dpu);
        else d4_fill_at<G0 + 1, N - 1>(g, cx, rg, dpu);
    }
}

// -delta of a unit's rows -> the dP tuple that unit's chain will accumulate into (stat = statistics slot + 16 hi)
template <int HALF>
LWM_DEVICE void d4_load_ndelta(uint32_t stat, f32x16& dp, int g) {
    const f32x4 v = lds_read_f32x4(stat + kD4BQ * 4 + HALF * 32 * 4 + 8 * g * 4);
    dp[4 * g + 0] = v[0]; dp[4 * g + 1] = v[1]; dp[4 * g + 2] = v[2]; dp[4 * g + 3] = v[3];
}

// Phase X of unit u (16 MFMAs): S(u) = Q K^T and dP'(u) = dO V^T - delta, the two chains ALTERNATING -- a dependent
// MFMA that does not follow its predecessor back to back waits for the whole chain before it (MI355X_MICROARCH.md,
// per-instruction constants: +43 cycles for the first instruction in between), an independent MFMA in between hides
// that -- || the second half of unit u-1's vector work: P -> bf16, dS = p dP', dS -> bf16.
// The first kD4Ahead fragments are ALREADY in the ring; the dP tuple holds -delta.
template <int HALF, bool HAS_PREV, bool DMA>
LWM_DEVICE void d4_x(const D4Ctx& cx, D4Regs& rg, const bf16x8 (&kf)[8], const bf16x8 (&vf)[8], const uint32_t (&vq)[4],
                     const uint32_t (&vdo)[4], const D4Dma& dm) {
    f32x16& dpn = rg.dp[HALF];            // this unit's dP'
    const f32x16& dpo = rg.dp[HALF ^ 1];  // the previous unit's
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        if ((m & 1) == 0) {
            rg.fr[(m + kD4Ahead) & 7] = d4_frag<HALF>(cx, m + kD4Ahead);
            rg.fr[(m + kD4Ahead + 1) & 7] = d4_frag<HALF>(cx, m + kD4Ahead + 1);
        }
        if ((m & 3) == 2) {      // the unit's -lse * log2 e (needed from gap 1 of phase Y)
            const int g = m >> 2;
            const f32x4 v = lds_read_f32x4(cx.stat + HALF * 32 * 4 + 8 * g * 4);
            rg.nl[4 * g + 0] = v[0]; rg.nl[4 * g + 1] = v[1]; rg.nl[4 * g + 2] = v[2]; rg.nl[4 * g + 3] = v[3];
        }
        sched_fence();
        {
            const int f = m ^ 1;      // the fragment this gap's MFMA consumes: even = Q (S), odd = dO (dP')
            if (f == 0) f4_mfma_s_first(rg.s, rg.fr[0], kf[0]);
            else if ((f & 1) == 0) f4_mfma_s(rg.s, rg.fr[f & 7], kf[f >> 1]);
            else if (f == 1) d4_mfma_p_first(dpn, rg.fr[1], vf[0]);
            else f4_mfma_s(dpn, rg.fr[f & 7], vf[f >> 1]);
        }
        if (DMA && (m & 1)) {         // the wave's 4 Q and 4 dO pieces of the step two ahead, all in the step's first phase
            const int j = m >> 2;
            if ((m & 2) == 0) f4_dma1(vq[j], dm.q_src, dm.dst + 4096 * j);
            else f4_dma1(vdo[j], dm.do_src, dm.dst + kD4TileBytes + 4096 * j);
        }
        if (HAS_PREV) d4_fill_at<16, 16>(16 + m, cx, rg, dpo);
        sched_fence();
    }
}

// masks of a unit on its scores (lwm/llama.py:572-592): query row (r&3) + 8 (r>>2) + 4 hi of the unit sees this lane's
// key iff it is not before it (rel = key position - position of the unit's row 0 - 4 hi, clamped) and, with key meta,
// shares its segment (kseg = kSegInvalid for a padded / out-of-range key)
template <int HALF, bool HAS_META>
LWM_DEVICE void d4_mask(const D4Ctx& cx, D4Regs& rg, int rel, int32_t kseg) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (HAS_META) {
            const u32x4 sg = lds_read_u32x4(cx.stat + 2 * kD4BQ * 4 + HALF * 32 * 4 + 8 * g * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool vis = ((int32_t)sg[j] == kseg) && (8 * g + j >= rel);
                rg.s[4 * g + j] = vis ? rg.s[4 * g + j] : -INFINITY;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) rg.s[4 * g + j] = (8 * g + j >= rel) ? rg.s[4 * g + j] : -INFINITY;
        }
    }
}

// Phase Y of unit u (16 MFMAs): dV^T += dO(u-1)^T P(u-1) and dK^T += Q(u-1)^T dS(u-1), eight accumulators in turn  ||
// the first half of unit u's vector work: t = S c - lse2 (from gap 1: the chains have left the pipe), p = exp2(t).
// The last gaps request the first fragments of the unit that follows and its -delta (statn = the statistics slot
// of that unit + 16 hi).
// INIT: these are the first products of the walk (C = 0: the tuples are defined here).
template <int HALF, bool HAS_PREV, bool NEXT, bool INIT>
LWM_DEVICE void d4_y(const D4Ctx& cx, D4Regs& rg, f32x16 (&dk)[4], f32x16 (&dv)[4], uint32_t statn) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int m = 16 + j;
        if ((j & 1) == 0 && (m + kD4Ahead < 32 || NEXT)) {
            rg.fr[(m + kD4Ahead) & 7] = d4_frag<HALF>(cx, m + kD4Ahead);
            rg.fr[(m + kD4Ahead + 1) & 7] = d4_frag<HALF>(cx, m + kD4Ahead + 1);
        }
        if (NEXT && j >= 12) d4_load_ndelta<HALF ^ 1>(statn, rg.dp[HALF ^ 1], j - 12);
        sched_fence();
        if (HAS_PREV) {
            const int jf = j ^ 1;      // the fragment consumed: even = dO^T (dV), odd = Q^T (dK)
            if (INIT && jf < 8) {
                if ((jf & 1) == 0) d4_mfma_o_first(dv[(jf >> 1) & 3], rg.fr[(16 + jf) & 7], rg.pb[0]);
                else d4_mfma_o_first(dk[(jf >> 1) & 3], rg.fr[(16 + jf) & 7], rg.dsb[0]);
            } else {
                if ((jf & 1) == 0) f4_mfma_o(dv[(jf >> 1) & 3], rg.fr[(16 + jf) & 7], rg.pb[jf >> 3]);
                else f4_mfma_o(dk[(jf >> 1) & 3], rg.fr[(16 + jf) & 7], rg.dsb[jf >> 3]);
            }
        }
        d4_fill_at<0, 16>(j, cx, rg, rg.dp[HALF]);
        sched_fence();
    }
}

// the pipeline's tail: the second half of the last unit's vector work, then its dV / dK products (no fillers left).
// PAR = the parity of the last unit (its dP' tuple); its tiles are addressed as the "previous unit" of a HALF = 0 unit.
template <int PAR>
LWM_DEVICE void d4_drain(const D4Ctx& cx, D4Regs& rg, f32x16 (&dk)[4], f32x16 (&dv)[4]) {
#pragma unroll
    for (int g = 16; g < 32; ++g) d4_fill_at<16, 16>(g, cx, rg, rg.dp[PAR]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rg.fr[j] = d4_frag<0>(cx, 16 + 8 * h + j);
        sched_fence();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int jj = 8 * h + j;
            if ((jj & 1) == 0) f4_mfma_o(dv[(jj >> 1) & 3], rg.fr[j], rg.pb[jj >> 3]);
            else f4_mfma_o(dk[(jj >> 1) & 3], rg.fr[j], rg.dsb[jj >> 3]);
        }
        sched_fence();
    }
}

// The epilogue of a key block: the wave's dK^T (scaled) and dV^T tiles, rows row0 .. row0 + 31 of (b, h), through the
// wave's LDS staging tile at tb (attn_common.h, "epilogue staging"; free of other readers and writers: the caller's
// barrier), merged with the ring carries, stored as whole rows -- bf16 results or f32 partials.  Everything the 32 store
// instructions share is read and computed ONCE (kernel arguments, the lane's first address of each buffer, the
// strides): left inside the row loop, hipcc re-read carry_in / final_out / Sk behind a wait per row and rebuilt the
// 64-bit address products -- 15 k cycles per key block (s_memtime), five times the data movement.
LWM_DEVICE void d4_store_tiles(const AttnParams& p, lds_t tb, const f32x16 (&dk)[4], const f32x16 (&dv)[4], int b, int h,
                               int row0, int lane) {
    const int l31 = lane & 31, hi = lane >> 5, col = l31 * 4;
    const int Sk = p.Sk;
    const bool carry = p.carry_in != 0, fin = p.final_out != 0;
    const int rows = Sk - (row0 + hi);                              // this lane stores rows row0 + hi + 2 i while 2 i < rows
    const int64_t acc0 = ((((int64_t)b * Sk + row0 + hi) * p.H + h) * kHeadDim + col) * 4;       // bytes
    const int64_t acc_step = (int64_t)2 * p.H * kHeadDim * 4;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (which) wave_lds_fence();        // (dK^T's rows have been read)
        epi_tile_write(tb, which ? dv : dk, which ? 1.0f : p.scale, l31, hi);
        wave_lds_fence();
        char* ap = (char*)(which ? p.dv_acc : p.dk_acc) + acc0;
        const int64_t o_ss = which ? p.dv_ss : p.dk_ss;
        char* op = (char*)((which ? p.dv : p.dk) + (int64_t)b * (which ? p.dv_sb : p.dk_sb) + (int64_t)(row0 + hi) * o_ss +
                           (int64_t)h * (which ? p.dv_sh : p.dk_sh) + col);
        const int64_t out_step = 2 * o_ss * 2;
        if (fin && !carry) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 x = epi_tile_read(tb, i, lane);
                if (2 * i < rows) global_store_b64(op + i * out_step, u32x2{pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3])});
            }
        } else if (!fin && !carry) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 x = epi_tile_read(tb, i, lane);
                if (2 * i < rows) global_store_f32x4((float*)(ap + i * acc_step), x);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                f32x4 x = epi_tile_read(tb, i, lane);
                if (2 * i < rows) {
                    const f32x4 c = global_load_f32x4((const float*)(ap + i * acc_step));
                    x[0] += c[0]; x[1] += c[1]; x[2] += c[2]; x[3] += c[3];
                    if (fin) global_store_b64(op + i * out_step, u32x2{pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3])});
                    else global_store_f32x4((float*)(ap + i * acc_step), x);
                }
            }
        }
    }
}

template <bool HAS_META>
LWM_DEVICE void attn_bwd_dkdv4_body(const AttnParams& p) {
#ifdef LWM_PROF
    // (dump slots 11..13: entry -> last store issued, entry -> first step of the walk, entry -> first store of the epilogue)
    const unsigned long long d4_entry = __builtin_amdgcn_s_memtime();
    unsigned long long d4_head = 0, d4_epi = 0;
#endif
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = wave_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;

    // ---- block -> (key block, head, batch): all key blocks of one (b,h) on one XCD, longest walks first
    const int nkb = (p.Sk + kD4BK - 1) / kD4BK;
    const int HB = p.H * p.B;
    int lin = block_idx_x(), kbi, hb;
    if ((HB & 7) == 0) {
        int xcd = lin & 7, i = lin >> 3;
        hb = xcd + 8 * (i / nkb);
        kbi = i % nkb;
    } else {
        hb = lin / nkb;
        kbi = lin % nkb;
    }
    const int b = hb / p.H, h = hb % p.H;
    const bf16_t* qb = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* kb = p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const bf16_t* vb = p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const bf16_t* dob = p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;

    // ---- this lane's key: fragments straight into the accumulator file (a row past Sk re-reads the last row: its
    // results are never stored)
    const int k_row = kbi * kD4BK + wave * 32 + l31;
    const bool k_ok = k_row < p.Sk;
    const int kr = k_ok ? k_row : p.Sk - 1;
    bf16x8 kf[8], vf[8];
    for (int s = 0; s < 8; ++s) kf[s] = f4_load_agpr(kb + (int64_t)kr * p.k_ss + 16 * s + 8 * hi);
    for (int s = 0; s < 8; ++s) vf[s] = f4_load_agpr(vb + (int64_t)kr * p.v_ss + 16 * s + 8 * hi);
    const PosMap qm = q_map(p);
    const PosTab qt_ = load_postab(qm);                           // (register copies of the tables: prologue only)
    const int64_t k_base = pos_base(load_postab(k_map(p)), kbi * kD4BK);       // position of key row r of this workgroup = k_base + r
    const int64_t k_pos = k_base + k_row;
    int32_t kseg = 0;
    if (HAS_META) {
        kseg = kSegInvalid;
        if (k_ok) {
            const bool valid = p.key_valid ? (p.key_valid[(int64_t)b * p.Sk + k_row] != 0) : true;
            if (valid) kseg = p.seg_k ? p.seg_k[(int64_t)b * p.Sk + k_row] : 0;
        }
    }

    // ---- step range of the walk (steps of 64 queries; causal: skip steps wholly before this key block)
    int nst = (p.Sq + kD4BQ - 1) / kD4BQ;
    int st0 = 0;
    if (p.causal) st0 = tiles_below(qt_, kD4BQ, nst, k_base + (int64_t)kbi * kD4BK - 1);   // steps wholly before the block's key 0
    if (HAS_META && p.segb_q && p.segb_k && st0 < nst) {     // packed sequences: skip other documents' query steps
        const int nbq = (p.Sq + 31) >> 5, nbk = (p.Sk + 31) >> 5;
        int smin, smax, lo, hi2;
        seg_own_range(p.segb_k + (int64_t)b * nbk * 2, nbk, kbi * (kD4BK / 32), kD4BK / 32, smin, smax);
        seg_narrow<kD4Threads>(p.segb_q + (int64_t)b * nbq * 2, nbq, kD4BQ / 32, st0, nst, smin, smax, lds + kD4OffStat, tid, lo, hi2);
        st0 = lo;
        nst = hi2;
    }
    const int n = nst > st0 ? nst - st0 : 0;       // steps of the walk; walk index i -> step nst - 1 - i (descending:
                                                   // the key blocks resident on an XCD read the same tile at the same time)

    f32x16 dk[4], dv[4];       // defined by the first products of the walk (d4_y<.., INIT>), or zero when there is no walk
    if (n == 0)
        for (int i = 0; i < 4; ++i) {
            dk[i] = zero_f32x16();
            dv[i] = zero_f32x16();
        }
    if (n == 0) {       // (no walk: the fragments are unused, their loads are not left in flight)
        f4_load_agpr_wait8(kf);
        f4_load_agpr_wait8(vf);
    }

    if (n > 0) {
        D4Ctx cx;
        cx.hi = hi;
        cx.c = p.scale * kLog2e;
        for (int s = 0; s < 8; ++s) cx.qa[s] = lds + tile_off(l31, 2 * s + hi);
        {
            const TrFragAddr t = frag_tr_addr(lds, lane);
            for (int db = 0; db < 4; ++db) {
                cx.tlo[db] = t.lo[db];
                cx.tup[db] = t.up[db];
                cx.plo[db] = t.lo[db];
                cx.pup[db] = t.up[db];
            }
        }
        cx.stat = lds + kD4OffStat + 16 * hi;

        const int64_t q_step_bytes = (int64_t)kD4BQ * p.q_ss * 2, do_step_bytes = (int64_t)kD4BQ * p.do_ss * 2;
        uint32_t vq[4], vdo[4];
        d4_stage_offsets(p, wave, lane, 0, vq, vdo);      // full steps: nothing is clamped
        // The row statistics travel like the tiles: ONE LDS-DMA dword instruction moves the 64 floats of a step's nl2 row
        // (wave 0), of its nd row (wave 1) and -- with key meta -- of its query segment ids (wave 2), first thing in the
        // step, so that the counted wait at the bottom (all but the 8 tile pieces of the step) covers them.
        const int64_t Sqp = bwd_stat_pad(p.Sq);
        const char* stat_src = (const char*)(p.delta + bwd_stat_row((int64_t)b * p.H + h, Sqp) + (wave == 1 ? Sqp : 0));
        bool stat_on = wave < 2;
        if (HAS_META && wave == 2 && p.seg_q) {
            stat_src = (const char*)(p.seg_q + (int64_t)b * p.Sq);
            stat_on = true;
        }
        const lds_t stat_dst = lds + kD4OffStat + (uint32_t)(wave < 3 ? wave : 0) * kD4BQ * 4;
        const uint32_t vstat = (uint32_t)lane * 4;
        if (HAS_META && !p.seg_q)                          // no query segments: every slot's ids read as 0
            for (int sl = 0; sl < kD4Slots; ++sl)
                if (tid < kD4BQ) lds_write_i32(lds + kD4OffStat + sl * kD4StatBytes + 2 * kD4BQ * 4 + tid * 4, 0);
        // staging of walk index i from the slow path (prologue; the ragged top step has its own offsets, and its
        // segment ids stop at Sq: the statistics rows are padded, the caller's ids are not)
        auto stage_step = [&](int i) {
            const int st = nst - 1 - i;
            const lds_t dst = lds + (i & 3) * kD4SlotBytes + (uint32_t)wave * 1024;
            const char* qs = (const char*)qb + st * q_step_bytes;
            const char* ds = (const char*)dob + st * do_step_bytes;
            if (stat_on) {
                int row = st * kD4BQ + lane;
                if (HAS_META && wave == 2) row = row < p.Sq ? row : p.Sq - 1;
                glds_load_b32(stat_src + (int64_t)row * 4, stat_dst + (i & 3) * kD4StatBytes);
            }
            if (st * kD4BQ + kD4BQ > p.Sq) {
                uint32_t v2[4], d2[4];
                d4_stage_offsets(p, wave, lane, st, v2, d2);
                f4_dma<4>(v2, qs, dst);
                f4_dma<4>(d2, ds, dst + kD4TileBytes);
            } else {
                f4_dma<4>(vq, qs, dst);
                f4_dma<4>(vdo, ds, dst + kD4TileBytes);
            }
        };

        // ---- prologue: steps 0 and 1 in flight; the K / V fragments are waited for together with them (one memory
        // latency per workgroup, not two in a row: 2-3 us of the ~50 us a 32-step walk of a ring shard takes)
        stage_step(0);
        if (n > 1) stage_step(1);
        f4_load_agpr_wait8(kf);      // vmcnt(0): everything requested so far
        f4_load_agpr_wait8(vf);
        block_sync();

        // A unit needs the mask code when its first query lies before the wave's last key (or keys can be masked for
        // another reason than causality).  Units are named by the row of their first query inside the q block,
        // ub = 64 * step + 32 * half (steps descend, the halves of a step ascend); everything is clamped to 32 bits once.
        auto clamp32 = [](int64_t x) -> int { return x > (1 << 30) ? (1 << 30) : (x < -(1 << 30) ? -(1 << 30) : (int)x); };
        // positions relative to q_start; a unit's first query (row ub of the q block) sits at rel_pos(qm, ub)
        const int k_rel = clamp32(k_pos - p.q_start) - 4 * hi;                                     // this lane's key
        PosCursor qc = cursor_begin(qt_);
        auto q_rel_of = [&](int ub) -> int {             // (the walk descends: a compare while the unit is inside the piece)
            cursor_seek(qm, qc, ub);
            return clamp32(qc.base + ub - p.q_start);
        };
        // positions ascend with the row: the units that begin before the wave's last key -- the ones that need the mask
        // code -- are the first mask_end rows of the q block (one compare per unit in the walk)
        const int mask_end = p.causal ? 32 * tiles_reaching(qt_, 32, (p.Sq + 31) / 32, k_base + (int64_t)kbi * kD4BK + wave * 32 + 31 - 1) : 0;
        auto needs_causal = [&](int ub) -> bool { return ub < mask_end; };
        // the wave's 32 keys all valid and of one segment?  (then a step whose 64 queries carry it needs no segment test)
        const int32_t own_seg = wave_uniform(kseg);
        const bool own_uniform = HAS_META && !wave_any(kseg != own_seg || kseg == kSegInvalid);
        auto rel_of = [&](int ub) -> int {
            if (!p.causal) return -64;
            const int64_t d = (int64_t)k_rel - q_rel_of(ub);
            return d > 64 ? 64 : (d < -64 ? -64 : (int)d);
        };

        D4Regs rg;
        rg.s = zero_f32x16();
        rg.dp[0] = zero_f32x16();
        rg.dp[1] = zero_f32x16();
        for (int r = 0; r < 16; ++r) {
            rg.nl[r] = 0.0f;
            rg.t[r] = 0.0f;
            rg.ds[r] = 0.0f;
        }
        for (int t = 0; t < 2; ++t) {
            rg.pb[t] = zero_bf16x8();
            rg.dsb[t] = zero_bf16x8();
        }
        for (int j = 0; j < 8; ++j) rg.fr[j] = zero_bf16x8();
        D4Dma dm = {};
        // the first unit's fragments and -delta (every later unit finds them requested by the unit before it)
        for (int j = 0; j < kD4Ahead; ++j) rg.fr[j] = d4_frag<0>(cx, j);
        for (int g = 0; g < 4; ++g) d4_load_ndelta<0>(cx.stat, rg.dp[0], g);

        // Step i of the walk.  Its tiles are in slot i & 3; the address registers in cx point there and move on to the slot
        // of step i+1 (wrapping) as the step ends: the row-fragment addresses behind the last X phase (the Y phase after
        // it already requests the next step's first fragments: that step's tiles became visible at the barrier before
        // this one), the others behind the barrier.  Top: the statistics piece and, inside the first X phase, the
        // LDS-DMA pieces of step i+2 -> slot (i+2) & 3 (last read by the lagging products at the head of step i-1).
        // Bottom: this wave's pieces have landed (they were issued >= 2000 cycles ago), one barrier.
#define LWM_D4_MASK(HALF_, HAS_PREV_, ub_)                                                                          \
    do {                                                                                                            \
        /* (the wait states sit in ONE statement on every path that needs them: with two, hipcc merges the tuples */ \
        /* of the two paths by copies placed right behind the last MFMA)                                         */ \
        if (!(HAS_PREV_)) d4_settle_t(rg.s);     /* no MFMA stands between the S chain and its first reader */       \
        if ((HAS_META && !uni_) || needs_causal(ub_)) {                                                             \
            if (HAS_PREV_) d4_settle_t(rg.s);                                                                       \
            if (HAS_META && !uni_) d4_mask<HALF_, HAS_META>(cx, rg, rel_of(ub_), kseg);                             \
            else d4_mask<HALF_, false>(cx, rg, rel_of(ub_), kseg);                                                  \
        }                                                                                                           \
    } while (0)
        // (-DLWM_PROF builds, scripts/micro/attn_bench with LWM_PROF_DUMP=1: s_memtime laps of the pipelined steps of the
        // longest key block of head 0 -- 0 X(0), 1 mask, 2 Y(0), 3 X(1), 4 mask, 5 Y(1), 6 counted wait, 7 barrier,
        // 8 address registers; 9 steps, 10 total.  LWM_PROF = 1 stamps only 5..8: a stamp drains lgkmcnt, i.e. the
        // fragments requested across a phase boundary)
#ifdef LWM_PROF
        unsigned long long d4p[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d4t = __builtin_amdgcn_s_memtime();
        const unsigned long long d4t0 = d4t;
        d4_head = d4t - d4_entry;
#define D4_LAP(slot)                                                  \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        d4p[slot] += now_ - d4t;                                      \
        d4t = now_;                                                   \
    } while (0)
#if LWM_PROF > 1
#define D4_LAP2(slot) D4_LAP(slot)
#else
#define D4_LAP2(slot)
#endif
#else
#define D4_LAP(slot)
#define D4_LAP2(slot)
#endif
        // `ub` = first row of the running step, `src` pointers = the step two ahead (they walk down with the steps)
        int ub = (nst - 1) * kD4BQ;
        const char* q_src2 = (const char*)qb + (int64_t)(nst - 3) * q_step_bytes;
        const char* do_src2 = (const char*)dob + (int64_t)(nst - 3) * do_step_bytes;
        const char* st_src2 = stat_src + (int64_t)(nst - 3) * kD4BQ * 4;
#define LWM_D4_STEP(i, FIRST, PIPE)                                                                                 \
    do {                                                                                                            \
        if (PIPE) {                                                                                                 \
            if (stat_on) d4_dma_b32(vstat, st_src2, stat_dst + (((i) + 2) & 3) * kD4StatBytes);        \
            dm.q_src = q_src2;                                                                                      \
            dm.do_src = do_src2;                                                                                    \
            dm.dst = lds + (((i) + 2) & 3) * kD4SlotBytes + (uint32_t)wave * 1024;                                  \
            q_src2 -= q_step_bytes;                                                                                 \
            do_src2 -= do_step_bytes;                                                                               \
            st_src2 -= kD4BQ * 4;                                                                                   \
        }                                                                                                           \
        const uint32_t d_ = (((i) & 3) == 3) ? (uint32_t)(-3 * kD4SlotBytes) : (uint32_t)kD4SlotBytes;             \
        const uint32_t e_ = (((i) & 3) == 3) ? (uint32_t)(-3 * kD4StatBytes) : (uint32_t)kD4StatBytes;             \
        const int32_t segw_ = HAS_META ? seg_step_word(cx.stat - 16 * hi + 2 * kD4BQ * 4, lane) : 0;                \
        d4_x<0, !(FIRST), PIPE>(cx, rg, kf, vf, vq, vdo, dm);                                                       \
        const bool uni_ = HAS_META && seg_step_uniform(segw_, own_uniform, own_seg);                                \
        D4_LAP2(0);                                                                                                 \
        LWM_D4_MASK(0, !(FIRST), ub);                                                                               \
        D4_LAP2(1);                                                                                                 \
        d4_y<0, !(FIRST), true, false>(cx, rg, dk, dv, cx.stat);                                                           \
        D4_LAP2(2);                                                                                                 \
        d4_x<1, true, false>(cx, rg, kf, vf, vq, vdo, dm);                                                           \
        for (int s_ = 0; s_ < 8; ++s_) cx.qa[s_] += d_;                                                             \
        D4_LAP2(3);                                                                                                 \
        LWM_D4_MASK(1, true, ub + 32);                                                                              \
        D4_LAP2(4);                                                                                                 \
        d4_y<1, true, true, FIRST>(cx, rg, dk, dv, cx.stat + e_);                                                          \
        D4_LAP(5);                                                                                                  \
        ub -= kD4BQ;                                                                                                \
        /* wherever control flow merges behind a step (loop entry, loop exit, the short-walk branch) hipcc may */  \
        /* reconcile the accumulator tuples of the two paths by copies: they must find the last MFMAs retired  */  \
        if ((FIRST) || !(PIPE)) d4_settle_acc(dk, dv);                                                              \
        glds_wait_all();                                                                                            \
        D4_LAP(6);                                                                                                  \
        block_sync_lds();                                                                                           \
        D4_LAP(7);                                                                                                  \
        for (int db_ = 0; db_ < 4; ++db_) {                                                                         \
            cx.plo[db_] = cx.tlo[db_];                                                                              \
            cx.pup[db_] = cx.tup[db_];                                                                              \
            cx.tlo[db_] += d_;                                                                                      \
            cx.tup[db_] += d_;                                                                                      \
        }                                                                                                           \
        cx.stat += e_;                                                                                              \
        D4_LAP(8);                                                                                                  \
    } while (0)

        int i = 0;
        if (n > 2) {
            LWM_D4_STEP(0, true, true);
#ifdef LWM_PROF
            for (int j = 0; j < 12; ++j) d4p[j] = 0;
            d4t = __builtin_amdgcn_s_memtime();
#endif
            for (i = 1; i + 2 < n; ++i) {
                LWM_D4_STEP(i, false, true);
#ifdef LWM_PROF
                d4p[9] += 1;
#endif
            }
#ifdef LWM_PROF
            if (!HAS_META && hb == 0 && kbi == 0 && lane == 0 && p.out_acc) {
                d4p[10] = __builtin_amdgcn_s_memtime() - d4t0;
                for (int j = 0; j < 11; ++j) ((unsigned long long*)p.out_acc)[wave * 16 + j] = d4p[j];
            }
#endif
            for (; i < n; ++i) LWM_D4_STEP(i, false, false);
        } else {
            LWM_D4_STEP(0, true, false);
            if (n > 1) LWM_D4_STEP(1, false, false);
        }
#undef LWM_D4_STEP
#undef D4_LAP
#undef D4_LAP2
#undef LWM_D4_MASK
        // the last unit's products: its tiles are the second half of the PREVIOUS slot now (the registers moved on)
        d4_drain<1>(cx, rg, dk, dv);
        d4_settle_acc(dk, dv);      // before the paths merge (hipcc reconciles the tuples by copies at the merge)
    }

    // ---- epilogue: scale, merge with the ring carries, store
    d4_settle_acc(dk, dv);
#ifdef LWM_PROF
    d4_epi = __builtin_amdgcn_s_memtime() - d4_entry;
#endif
    // The tiles leave as whole rows (attn_common.h, "epilogue staging"): the walk is over for every wave behind the
    // barrier, so the tile slots are free; a wave passes dK^T, then dV^T through its own 16 KiB of them.
    block_sync_lds();
    d4_store_tiles(p, lds + (uint32_t)wave * kEpiTileBytes, dk, dv, b, h, kbi * kD4BK + wave * 32, lane);
#ifdef LWM_PROF
    if (!HAS_META && hb == 0 && kbi == 0 && lane == 0 && p.out_acc) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the stores have left)
        unsigned long long* o_ = (unsigned long long*)p.out_acc + wave * 16;
        o_[11] = __builtin_amdgcn_s_memtime() - d4_entry;
        o_[12] = d4_head;
        o_[13] = d4_epi;
    }
#endif
}


LWM_KERNEL(kD4Threads) void attn_bwd_dkdv4_kernel(AttnParams p) { attn_bwd_dkdv4_body<false>(p); }
LWM_KERNEL(kD4Threads) void attn_bwd_dkdv4_meta_kernel(AttnParams p) { attn_bwd_dkdv4_body<true>(p); }


// ====================================================================================================== dQ
// The dQ kernel of the two-kernel backward in the same structure.  Replaces the dq part of the custom VJP (see the
// file header): p from the saved LSE, dp = do v^T, ds = p * (dp - delta), dq += ds k * scale.  Same contract,
// operands, masks, carries and segment-block hints as the dK/dV kernel above.
//
// Workgroup = 4 waves = 128 queries; a wave owns 32 query rows: their Q and dO fragments (B operands, 32 + 32
// registers) and the f32 dQ^T accumulator (64) live in the accumulator file.  Everything is computed transposed, as
// in the forward: a lane owns one query column, so the row statistics are two scalars per lane and -delta, the initial
// value of the dP chain, is ONE constant tuple for the whole launch (no per-unit statistics traffic at all).  K / V
// arrive by LDS-DMA in steps of 64 keys through the same ring of four 32-KiB slots [K tile | V tile]; a unit = 32 keys:
//
//   phase   matrix pipe                                                   vector pipe (same wave)
//   X(u)    S^T(u) = K(u) Q^T  and  dP'^T(u) = V(u) dO^T - delta, 16      the vector work of unit u-1 continues
//   Y(u)    dQ^T += K(u-1)^T dS^T(u-1), 8 MFMAs over 4 tuples             t = S c - lse2, p = exp2(t), dS = p dP' -> bf16
//
// S and dP' have two register tiles each (by unit parity): the 56 VALU of a unit spread over the 24 gaps of Y(u) and
// X(u+1).  With key meta (segment ids, padded keys, a ragged last tile) the 64 meta words of a step are staged through
// LDS by the threads themselves, as the forward does.
//
// LDS map: slot 0..3 = [K tile 64 rows | V tile 64 rows] (16 KiB each) | key meta 0..3 (64 words each)
constexpr int kQ4BQ = 128;      // queries per workgroup
constexpr int kQ4BK = 64;       // keys per step (two units of 32)
constexpr int kQ4OffMeta = kD4Slots * kD4SlotBytes;
constexpr int kQ4LdsBytes = kQ4OffMeta + kD4Slots * kQ4BK * 4;

// G = 0..7 = the gaps of Y(u), G = 8..23 = the gaps of X(u+1).  F >= 1; E behind F, M behind E, D behind its two M; the
// product of Y(u) still reads the OLD dS words: the first half (i < 4) may be rewritten from G = 4, the second from G = 8;
// everything done by G = 21.
struct Q4Sched {
    int F[16], E[16], M[16], D[8];
};
constexpr bool q4_sched_ok(const Q4Sched& c) {
    for (int e = 0; e < 16; ++e)
        if (c.F[e] < 1 || c.E[e] <= c.F[e] || c.M[e] <= c.E[e] || c.M[e] > 21) return false;
    for (int i = 0; i < 8; ++i)
        if (c.D[i] < (i < 4 ? 4 : 8) || c.D[i] > 21 || c.D[i] <= c.M[2 * i] || c.D[i] <= c.M[2 * i + 1]) return false;
    return true;
}
constexpr Q4Sched kQ4Sched = {      // two per gap beside the transposed reads of Y, three beside the row reads of X
    /* F */ {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8},
    /* E */ {8, 9, 9, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12, 13, 13, 13},
    /* M */ {14, 14, 14, 15, 15, 15, 16, 16, 16, 17, 17, 17, 18, 18, 18, 19},
    /* D */ {19, 19, 20, 20, 20, 21, 21, 21}};
static_assert(q4_sched_ok(kQ4Sched), "filler schedule violates a dependency");

struct Q4Ctx {
    uint32_t ka[8];             // K row-fragment addresses (d step s), rows 0..31 of the CURRENT step's K tile
    uint32_t tlo[4], tup[4];    // K transposed-fragment addresses (d block), rows 0..15 of the current step's K tile
    uint32_t plo[4], pup[4];    // the same in the PREVIOUS step's slot
    uint32_t meta;              // this step's key meta + 16 * hi
    float c, nl;                // scale * log2(e); -lse * log2(e) of this lane's query (-inf: p = 0)
};

struct Q4Regs {
    f32x16 s[2], dp[2];         // S^T / dP'^T tiles by unit parity (MFMA results, read-only for the vector pipe)
    f32x16 ndl;                 // -delta of this lane's query in every element: the C operand that starts each dP chain
    float t[16], ds[16];
    bf16x8 dsb[2];
    uint32_t dw[8];
    bf16x8 fr[8];
};

#ifdef LWM_EMU
LWM_DEVICE void q4_mfma_c_first(f32x16& d, bf16x8 a, bf16x8 b, const f32x16& c) { d = mfma_32x32x16(a, b, c); }
LWM_DEVICE void q4_opaque(f32x16&) {}
#else
// first MFMA of a dP chain: C = the -delta tuple (two wait states in front, see d4_mfma_p_first)
LWM_DEVICE void q4_mfma_c_first(f32x16& d, bf16x8 a, bf16x8 b, const f32x16& c) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
// sixteen registers that hold the same value: hipcc must not know (it would keep one and copy it into a tuple in
// front of every chain)
LWM_DEVICE void q4_opaque(f32x16& x) { asm volatile("" : "+v"(x)); }
#endif

// fragment f of a unit (MFMA index 0..23; 24.. = the first fragments of the unit that follows):
//   0..15   phase X: d step f>>1; even = K row fragment (S^T), odd = V row fragment (dP'^T)
//   16..23  phase Y: K^T of the PREVIOUS unit, (key step (f>>2)&1, d block f&3)
template <int HALF>
LWM_DEVICE bf16x8 q4_frag(const Q4Ctx& cx, int f) {
    if (f >= 24) {
        const int g = f - 24;
        return lds_read_b128(cx.ka[g >> 1] + (g & 1) * kD4TileBytes + (HALF == 0 ? 32 * kRowBytes : 0));
    }
    if (f < 16) return lds_read_b128(cx.ka[f >> 1] + (f & 1) * kD4TileBytes + HALF * 32 * kRowBytes);
    const int j = f - 16, t = (j >> 2) & 1, db = j & 3;
    const uint32_t alo = HALF == 0 ? cx.plo[db] : cx.tlo[db];
    const uint32_t aup = HALF == 0 ? cx.pup[db] : cx.tup[db];
    const uint32_t off = (HALF == 0 ? 32 * kRowBytes : 0) + 16 * t * kRowBytes;
    bf16x4 lo = lds_read_tr16(alo + off);
    bf16x4 up = lds_read_tr16(aup + off);
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = up[0]; o[5] = up[1]; o[6] = up[2]; o[7] = up[3];
    return o;
}

// the vector work scheduled in gap G for the unit of parity PAR
template <int G, int PAR>
LWM_DEVICE void q4_fillers(const Q4Ctx& cx, Q4Regs& rg) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (kQ4Sched.F[e] == G) rg.t[e] = f4_fma(rg.s[PAR][e], cx.c, cx.nl);
        if (kQ4Sched.E[e] == G) rg.t[e] = f4_exp2(rg.t[e]);
        if (kQ4Sched.M[e] == G) rg.ds[e] = d4_mul(rg.t[e], rg.dp[PAR][e]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (kQ4Sched.D[i] == G) rg.dw[i] = f4_cvt_pk(rg.ds[2 * i], rg.ds[2 * i + 1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int ld = 0;
        for (int i = 4 * h; i < 4 * h + 4; ++i) ld = kQ4Sched.D[i] > ld ? kQ4Sched.D[i] : ld;
        if (ld == G) rg.dsb[h] = __builtin_bit_cast(bf16x8, u32x4{rg.dw[4 * h], rg.dw[4 * h + 1], rg.dw[4 * h + 2], rg.dw[4 * h + 3]});
    }
}
template <int G0, int N, int PAR>
LWM_DEVICE void q4_fill_at(int g, const Q4Ctx& cx, Q4Regs& rg) {
    if constexpr (N > 0) {
        if (g == G0) q4_fillers<G0, PAR>(cx, rg);
        else q4_fill_at<G0 + 1, N - 1, PAR>(g, cx, rg);
    }
}

// Phase X of unit u (parity HALF): the two chains alternating, fragments requested and consumed in pairs (attn d4_x);
// fillers: gaps 8..23 of unit u-1.
template <int HALF, bool HAS_PREV, bool DMA>
LWM_DEVICE void q4_x(const Q4Ctx& cx, Q4Regs& rg, const bf16x8 (&qf)[8], const bf16x8 (&dof)[8], const uint32_t (&vk)[4],
                     const uint32_t (&vv)[4], const D4Dma& dm) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        if ((m & 1) == 0) {
            rg.fr[(m + kD4Ahead) & 7] = q4_frag<HALF>(cx, m + kD4Ahead);
            rg.fr[(m + kD4Ahead + 1) & 7] = q4_frag<HALF>(cx, m + kD4Ahead + 1);
        }
        sched_fence();
        {
            const int f = m ^ 1;      // even = K (S^T), odd = V (dP'^T)
            if (f == 0) f4_mfma_s_first(rg.s[HALF], rg.fr[0], qf[0]);
            else if ((f & 1) == 0) f4_mfma_s(rg.s[HALF], rg.fr[f & 7], qf[f >> 1]);
            else if (f == 1) q4_mfma_c_first(rg.dp[HALF], rg.fr[1], dof[0], rg.ndl);
            else f4_mfma_s(rg.dp[HALF], rg.fr[f & 7], dof[f >> 1]);
        }
        if (DMA && (m & 1)) {
            const int j = m >> 2;
            if ((m & 2) == 0) f4_dma1(vk[j], dm.q_src, dm.dst + 4096 * j);
            else f4_dma1(vv[j], dm.do_src, dm.dst + kD4TileBytes + 4096 * j);
        }
        if (HAS_PREV) q4_fill_at<8, 16, HALF ^ 1>(8 + m, cx, rg);
        sched_fence();
    }
}

// masks of a unit on its transposed scores (lwm/llama.py:572-592): key row (r&3) + 8 (r>>2) + 4 hi of the unit is visible to
// this lane's query iff it is not after it (rel = query position - position of the unit's key 0 - 4 hi, clamped) and, with
// key meta, carries the query's segment (kSegInvalid: a padded / out-of-range key)
template <int HALF, bool HAS_META>
LWM_DEVICE void q4_mask(const Q4Ctx& cx, f32x16& s, int rel, int32_t seg_q) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (HAS_META) {
            const u32x4 sg = lds_read_u32x4(cx.meta + HALF * 32 * 4 + 8 * g * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool vis = ((int32_t)sg[j] == seg_q) && (8 * g + j <= rel);
                s[4 * g + j] = vis ? s[4 * g + j] : -INFINITY;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[4 * g + j] = (8 * g + j <= rel) ? s[4 * g + j] : -INFINITY;
        }
    }
}

// Phase Y of unit u: dQ^T += K(u-1)^T dS^T(u-1)  ||  gaps 0..7 of unit u.  INIT: the first products of the walk.
template <int HALF, bool HAS_PREV, bool INIT>
LWM_DEVICE void q4_y(const Q4Ctx& cx, Q4Regs& rg, f32x16 (&dq)[4]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = 16 + j;
        if ((j & 1) == 0) {
            rg.fr[(m + kD4Ahead) & 7] = q4_frag<HALF>(cx, m + kD4Ahead);
            rg.fr[(m + kD4Ahead + 1) & 7] = q4_frag<HALF>(cx, m + kD4Ahead + 1);
        }
        sched_fence();
        if (HAS_PREV) {
            const int jf = j ^ 1;      // (pairs: the younger fragment first -- one wait per two MFMAs)
            if (INIT && jf < 4) d4_mfma_o_first(dq[jf & 3], rg.fr[(16 + jf) & 7], rg.dsb[0]);
            else f4_mfma_o(dq[jf & 3], rg.fr[(16 + jf) & 7], rg.dsb[jf >> 2]);
        }
        q4_fill_at<0, 8, HALF>(j, cx, rg);
        sched_fence();
    }
}

template <int PAR>
LWM_DEVICE void q4_drain(const Q4Ctx& cx, Q4Regs& rg, f32x16 (&dq)[4]) {
#pragma unroll
    for (int g = 8; g < 24; ++g) q4_fill_at<8, 16, PAR>(g, cx, rg);
#pragma unroll
    for (int j = 0; j < 8; ++j) rg.fr[j] = q4_frag<0>(cx, 16 + j);
    sched_fence();
#pragma unroll
    for (int j = 0; j < 8; ++j) f4_mfma_o(dq[j & 3], rg.fr[j], rg.dsb[j >> 2]);
}

// per-lane byte offsets of the wave's four pieces of a 64-row K / V tile, rows clamped to Sk-1 (rows past Sk are masked
// through the key meta)
LWM_DEVICE void q4_stage_offsets(const AttnParams& p, int wave, int lane, int st, uint32_t (&vk)[4], uint32_t (&vv)[4]) {
    const int slot = lane & 15;
    for (int j = 0; j < 4; ++j) {
        const int row = 4 * (wave + 4 * j) + (lane >> 4);
        int krow = st * kQ4BK + row;
        krow = krow < p.Sk ? krow : p.Sk - 1;
        const int rel = krow - st * kQ4BK;
        const int col = (slot ^ swz(row)) << 3;
        vk[j] = (uint32_t)(((int64_t)rel * p.k_ss + col) * 2);
        vv[j] = (uint32_t)(((int64_t)rel * p.v_ss + col) * 2);
    }
}

template <bool HAS_META>
LWM_DEVICE void attn_bwd_dq4_body(const AttnParams& p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = wave_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;

    // ---- block -> (q block, head, batch): all q blocks of one (b,h) on one XCD, longest walks (the last blocks) first
    const int nqb = (p.Sq + kQ4BQ - 1) / kQ4BQ;
    const int HB = p.H * p.B;
    int lin = block_idx_x(), qbi, hb;
    if ((HB & 7) == 0) {
        int xcd = lin & 7, i = lin >> 3;
        hb = xcd + 8 * (i / nqb);
        qbi = nqb - 1 - (i % nqb);
    } else {
        hb = lin / nqb;
        qbi = nqb - 1 - (lin % nqb);
    }
    const int b = hb / p.H, h = hb % p.H;
    const bf16_t* qb = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* kb = p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const bf16_t* vb = p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const bf16_t* dob = p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;

    // ---- this lane's query: fragments straight into the accumulator file (a row past Sq re-reads the last row: its
    // results are never stored), its statistics
    const int q_row = qbi * kQ4BQ + wave * 32 + l31;
    const bool q_ok = q_row < p.Sq;
    const int qr = q_ok ? q_row : p.Sq - 1;
    bf16x8 qf[8], dof[8];
    for (int s = 0; s < 8; ++s) qf[s] = f4_load_agpr(qb + (int64_t)qr * p.q_ss + 16 * s + 8 * hi);
    for (int s = 0; s < 8; ++s) dof[s] = f4_load_agpr(dob + (int64_t)qr * p.do_ss + 16 * s + 8 * hi);
    Q4Ctx cx;
    Q4Regs rg;
    {
        const int64_t Sqp = bwd_stat_pad(p.Sq), srow = bwd_stat_row((int64_t)b * p.H + h, Sqp);
        cx.nl = q_ok ? p.delta[srow + qr] : -INFINITY;
        const float nd = p.delta[srow + Sqp + qr];
        for (int r = 0; r < 16; ++r) rg.ndl[r] = nd;
        q4_opaque(rg.ndl);
    }
    cx.c = p.scale * kLog2e;
    const int32_t seg_q = (HAS_META && q_ok && p.seg_q) ? p.seg_q[(int64_t)b * p.Sq + q_row] : 0;
    const PosMap km = k_map(p);
    const PosTab kt_ = load_postab(km);                           // (register copies of the tables: prologue only)
    const int64_t q_base = pos_base(load_postab(q_map(p)), qbi * kQ4BQ);       // position of query row r of this workgroup = q_base + r

    // ---- key step range of the walk (steps of 64 keys; causal: up to the diagonal of the workgroup's last query)
    const int nst_all = (p.Sk + kQ4BK - 1) / kQ4BK;
    int nst = nst_all, st0 = 0;
    const int q_last = (qbi * kQ4BQ + kQ4BQ < p.Sq ? qbi * kQ4BQ + kQ4BQ : p.Sq) - 1;
    if (p.causal) nst = tiles_reaching(kt_, kQ4BK, nst_all, q_base + q_last);      // up to the step of the last visible key
    if (HAS_META && p.segb_q && p.segb_k && nst > 0) {     // packed sequences: skip other documents' key steps
        const int nbq = (p.Sq + 31) >> 5, nbk = (p.Sk + 31) >> 5;
        int smin, smax, lo, hi2;
        seg_own_range(p.segb_q + (int64_t)b * nbq * 2, nbq, qbi * (kQ4BQ / 32), kQ4BQ / 32, smin, smax);
        seg_narrow<kD4Threads>(p.segb_k + (int64_t)b * nbk * 2, nbk, kQ4BK / 32, 0, nst, smin, smax, lds + kQ4OffMeta, tid, lo, hi2);
        st0 = lo;
        nst = hi2;
    }
    const int n = nst > st0 ? nst - st0 : 0;       // walk index i -> key step st0 + i (ascending)

    f32x16 dq[4];
    if (n == 0)
        for (int i = 0; i < 4; ++i) dq[i] = zero_f32x16();
    if (n == 0) {
        f4_load_agpr_wait8(qf);
        f4_load_agpr_wait8(dof);
    }

    if (n > 0) {
        for (int s = 0; s < 8; ++s) cx.ka[s] = lds + tile_off(l31, 2 * s + hi);
        {
            const TrFragAddr t = frag_tr_addr(lds, lane);
            for (int db = 0; db < 4; ++db) {
                cx.tlo[db] = t.lo[db];
                cx.tup[db] = t.up[db];
                cx.plo[db] = t.lo[db];
                cx.pup[db] = t.up[db];
            }
        }
        cx.meta = lds + kQ4OffMeta + 16 * hi;
        const int64_t k_step_bytes = (int64_t)kQ4BK * p.k_ss * 2, v_step_bytes = (int64_t)kQ4BK * p.v_ss * 2;
        uint32_t vk[4], vv[4];
        q4_stage_offsets(p, wave, lane, 0, vk, vv);       // full steps: nothing is clamped
        // key meta of walk index i -> meta slot i & 3: segment id, or kSegInvalid for padded / out-of-range keys
        auto meta_stage = [&](int i) {
            if (HAS_META && tid < kQ4BK) {
                const int krow = (st0 + i) * kQ4BK + tid;
                const int kr = krow < p.Sk ? krow : p.Sk - 1;
                const uint8_t kvalid = p.key_valid ? p.key_valid[(int64_t)b * p.Sk + kr] : (uint8_t)1;
                const int32_t kseg = p.seg_k ? p.seg_k[(int64_t)b * p.Sk + kr] : 0;
                lds_write_i32(lds + kQ4OffMeta + (i & 3) * kQ4BK * 4 + tid * 4, (krow < p.Sk && kvalid != 0) ? kseg : kSegInvalid);
            }
        };
        auto stage_step = [&](int i) {
            const int st = st0 + i;
            const lds_t dst = lds + (i & 3) * kD4SlotBytes + (uint32_t)wave * 1024;
            const char* ks = (const char*)kb + st * k_step_bytes;
            const char* vs = (const char*)vb + st * v_step_bytes;
            if (st * kQ4BK + kQ4BK > p.Sk) {
                uint32_t v2[4], d2[4];
                q4_stage_offsets(p, wave, lane, st, v2, d2);
                f4_dma<4>(v2, ks, dst);
                f4_dma<4>(d2, vs, dst + kD4TileBytes);
            } else {
                f4_dma<4>(vk, ks, dst);
                f4_dma<4>(vv, vs, dst + kD4TileBytes);
            }
            meta_stage(i);
        };
        // the ragged last step of the K/V block may only be staged from this slow path: the pipelined steps stop before it
        const bool ragged = (p.Sk % kQ4BK) != 0;
        const int n_pipe_end = (ragged && nst == nst_all) ? n - 1 : n;      // walk indices >= this are never issued in-loop

        // ---- prologue: steps 0 and 1 in flight, waited for together with the Q / dO fragments
        stage_step(0);
        if (n > 1) stage_step(1);
        f4_load_agpr_wait8(qf);      // vmcnt(0): everything requested so far
        f4_load_agpr_wait8(dof);
        block_sync();

        auto clamp32 = [](int64_t x) -> int { return x > (1 << 30) ? (1 << 30) : (x < -(1 << 30) ? -(1 << 30) : (int)x); };
        // positions relative to k_start; a unit's first key (row ub of the K/V block) sits at rel_pos(km, ub)
        const int q_rel = clamp32(q_base + q_row - p.k_start) - 4 * hi;                         // this lane's query
        PosCursor kc = cursor_begin(kt_);
        auto k_rel_of = [&](int ub) -> int {             // (the walk ascends)
            cursor_seek(km, kc, ub);
            return clamp32(kc.base + ub - p.k_start);
        };
        // a unit needs the mask code when its last key lies after the wave's first query: positions ascend with the row,
        // so those are the units from row mask_from of the K/V block on (one compare per unit in the walk)
        const int mask_from = p.causal ? 32 * tiles_below(kt_, 32, (p.Sk + 31) / 32, q_base + (int64_t)qbi * kQ4BQ + wave * 32) : 0x7fffffff;
        auto needs_causal = [&](int ub) -> bool { return ub >= mask_from; };
        // the wave's 32 queries of one segment?  (then a step whose 64 keys carry it needs no segment test; rows past Sq
        // are never stored and do not count)
        const int32_t own_seg = wave_uniform(seg_q);
        const bool own_uniform = HAS_META && !wave_any(q_ok && seg_q != own_seg);
        auto rel_of = [&](int ub) -> int {
            if (!p.causal) return 64;
            const int64_t d = (int64_t)q_rel - k_rel_of(ub);
            return d > 64 ? 64 : (d < -64 ? -64 : (int)d);
        };

        for (int par = 0; par < 2; ++par) {
            rg.s[par] = zero_f32x16();
            rg.dp[par] = zero_f32x16();
        }
        for (int r = 0; r < 16; ++r) {
            rg.t[r] = 0.0f;
            rg.ds[r] = 0.0f;
        }
        for (int t = 0; t < 2; ++t) rg.dsb[t] = zero_bf16x8();
        for (int j = 0; j < 8; ++j) rg.fr[j] = zero_bf16x8();
        D4Dma dm = {};
        for (int j = 0; j < kD4Ahead; ++j) rg.fr[j] = q4_frag<0>(cx, j);

#define LWM_Q4_MASK(HALF_, HAS_PREV_, ub_)                                                                          \
    do {                                                                                                            \
        if (!(HAS_PREV_)) d4_settle_t(rg.s[HALF_]);     /* no MFMA stands between the S chain and its first reader */ \
        if ((HAS_META && !uni_) || needs_causal(ub_)) {                                                             \
            if (HAS_PREV_) d4_settle_t(rg.s[HALF_]);                                                                \
            if (HAS_META && !uni_) q4_mask<HALF_, HAS_META>(cx, rg.s[HALF_], rel_of(ub_), seg_q);                   \
            else q4_mask<HALF_, false>(cx, rg.s[HALF_], rel_of(ub_), seg_q);                                        \
        }                                                                                                           \
    } while (0)
        int ub = st0 * kQ4BK;
        const char* k_src2 = (const char*)kb + (int64_t)(st0 + 2) * k_step_bytes;
        const char* v_src2 = (const char*)vb + (int64_t)(st0 + 2) * v_step_bytes;
#define LWM_Q4_STEP(i, FIRST, PIPE)                                                                                 \
    do {                                                                                                            \
        if (PIPE) {                                                                                                 \
            dm.q_src = k_src2;                                                                                      \
            dm.do_src = v_src2;                                                                                     \
            dm.dst = lds + (((i) + 2) & 3) * kD4SlotBytes + (uint32_t)wave * 1024;                                  \
            k_src2 += k_step_bytes;                                                                                 \
            v_src2 += v_step_bytes;                                                                                 \
            meta_stage((i) + 2);                                                                                    \
        }                                                                                                           \
        const uint32_t d_ = (((i) & 3) == 3) ? (uint32_t)(-3 * kD4SlotBytes) : (uint32_t)kD4SlotBytes;             \
        const uint32_t e_ = (((i) & 3) == 3) ? (uint32_t)(-3 * kQ4BK * 4) : (uint32_t)(kQ4BK * 4);                 \
        const int32_t segw_ = HAS_META ? seg_step_word(cx.meta - 16 * hi, lane) : 0;                               \
        q4_x<0, !(FIRST), PIPE>(cx, rg, qf, dof, vk, vv, dm);                                                       \
        const bool uni_ = HAS_META && seg_step_uniform(segw_, own_uniform, own_seg);                                \
        LWM_Q4_MASK(0, !(FIRST), ub);                                                                               \
        q4_y<0, !(FIRST), false>(cx, rg, dq);                                                                       \
        q4_x<1, true, false>(cx, rg, qf, dof, vk, vv, dm);                                                          \
        for (int s_ = 0; s_ < 8; ++s_) cx.ka[s_] += d_;                                                             \
        LWM_Q4_MASK(1, true, ub + 32);                                                                              \
        q4_y<1, true, FIRST>(cx, rg, dq);                                                                           \
        ub += kQ4BK;                                                                                                \
        if ((FIRST) || !(PIPE)) d4_settle_acc4(dq);                                                                 \
        glds_wait_all();                                                                                            \
        block_sync_lds();                                                                                           \
        for (int db_ = 0; db_ < 4; ++db_) {                                                                         \
            cx.plo[db_] = cx.tlo[db_];                                                                              \
            cx.pup[db_] = cx.tup[db_];                                                                              \
            cx.tlo[db_] += d_;                                                                                      \
            cx.tup[db_] += d_;                                                                                      \
        }                                                                                                           \
        cx.meta += e_;                                                                                              \
    } while (0)

        int i = 0;
        if (n_pipe_end > 2) {
            LWM_Q4_STEP(0, true, true);
            for (i = 1; i + 2 < n_pipe_end; ++i) LWM_Q4_STEP(i, false, true);
            // the ragged last step (if any) was not issued by the loop: stage it now, two steps ahead of its use
            if (n_pipe_end < n) {
                stage_step(n - 1);
                glds_wait_all();
                block_sync_lds();
            }
            for (; i < n; ++i) LWM_Q4_STEP(i, false, false);
        } else {
            if (n > 2) {                 // (n == 3 with a ragged last step: nothing was pipelined)
                stage_step(2);
                glds_wait_all();
                block_sync_lds();
            }
            LWM_Q4_STEP(0, true, false);
            for (i = 1; i < n; ++i) LWM_Q4_STEP(i, false, false);
        }
#undef LWM_Q4_STEP
#undef LWM_Q4_MASK
        // the last unit's product: its K tile is the second half of the PREVIOUS slot now (the registers moved on)
        q4_drain<1>(cx, rg, dq);
        d4_settle_acc4(dq);      // before the paths merge
    }

    // ---- epilogue: scale, merge with the ring carry, store (one query row per lane)
    d4_settle_acc4(dq);
    // (what the 16 stores share is read once and the three cases are three loops: left inside one loop, hipcc re-read
    // carry_in / final_out behind a wait per store -- see d4_store_tiles)
    if (q_ok) {
        const bool carry = p.carry_in != 0, fin = p.final_out != 0;
        const float sc = p.scale;
        bf16_t* const op = p.dq + (int64_t)b * p.dq_sb + (int64_t)q_row * p.dq_ss + (int64_t)h * p.dq_sh + 4 * hi;
        float* const ap = p.dq_acc + (int64_t)b * p.dqa_sb + (int64_t)q_row * p.dqa_ss + (int64_t)h * p.dqa_sh + 4 * hi;
        if (fin && !carry) {
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    global_store_b64(op + 32 * db + 8 * rq, u32x2{pack_bf16x2(dq[db][4 * rq + 0] * sc, dq[db][4 * rq + 1] * sc),
                                                                  pack_bf16x2(dq[db][4 * rq + 2] * sc, dq[db][4 * rq + 3] * sc)});
        } else {
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d0 = 32 * db + 8 * rq;
                    float o0 = dq[db][4 * rq + 0] * sc, o1 = dq[db][4 * rq + 1] * sc;
                    float o2 = dq[db][4 * rq + 2] * sc, o3 = dq[db][4 * rq + 3] * sc;
                    if (carry) {
                        const f32x4 a = global_load_f32x4(ap + d0);
                        o0 += a[0]; o1 += a[1]; o2 += a[2]; o3 += a[3];
                    }
                    if (fin) global_store_b64(op + d0, u32x2{pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)});
                    else global_store_f32x4(ap + d0, f32x4{o0, o1, o2, o3});
                }
        }
    }
}

LWM_KERNEL(kD4Threads) void attn_bwd_dq4_kernel(AttnParams p) { attn_bwd_dq4_body<false>(p); }
LWM_KERNEL(kD4Threads) void attn_bwd_dq4_meta_kernel(AttnParams p) { attn_bwd_dq4_body<true>(p); }

}  // namespace lwm
