// attn_common.h -- shared tile geometry for the blockwise attention kernels.
//
// Requires a wave_ops.h (product: lwm_amd/csrc/wave_ops.h) to be included
// first.  Restates, MI355X-first, the inner update of the reference's
// blockwise ring attention: lwm/llama.py:539-569 calls
// ringattention(q,k,v,bias,segment_ids, float32_logits=True,
// causal_block_size=1, query/key chunk sizes) -- the arithmetic lives in the
// un-vendored `ringattention` package; the in-tree mask specification is
// lwm/llama.py:572-592.
//
// Geometry (all kernels): head_dim D = 128 (LWM-7B: lwm/llama.py:70-81),
// bf16 operands, f32 logits/softmax/accumulators, v_mfma_f32_32x32x16_bf16.
//
// LDS tile image: rows of 128 bf16 = 256 B = 16 slots of 16 B.  Slot s of row r
// is stored at physical slot  s ^ swz(r),  swz(r) = ((r&3)<<2) | ((r>>2)&3).
//   * row-fragment reads (ds_read_b128, 16 lanes = 16 distinct r&15 at one
//     logical slot) hit 16 distinct physical slots      -> conflict-free;
//   * transpose reads (ds_read_b64_tr_b16, a half-wave = 4 consecutive rows x
//     64 contiguous logical bytes) land in 4 distinct 64-B chunks, because the
//     chunk index is XORed with r&3                     -> conflict-free.
#pragma once

namespace lwm {

constexpr int kHeadDim = 128;
constexpr int kRowBytes = kHeadDim * 2;  // 256
constexpr int kMaxPieces = 8;            // LWM_MAX_PIECES of include/lwm_hip.h

struct AttnParams {
    // forward operands
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    bf16_t* out;        // final bf16 output (final_out != 0)
    float* lse;         // final natural-log LSE [B,H,Sq]
    float* out_acc;     // f32 carry [B,Sq,H,D], normalised partial output
    float* lse_acc;     // f32 carry [B,H,Sq]
    // backward operands
    const bf16_t* dout;
    const float* delta;  // backward row statistics (attn_bwd.h): per (b,h) [-lse*log2e | -rowsum(dO*O)], rows padded to 64
    bf16_t* dq;
    bf16_t* dk;
    bf16_t* dv;
    float* dq_acc;      // f32 carry [B,Sq,H,D]
    float* dk_acc;      // f32 carry [B,Sk,H,D]
    float* dv_acc;
    // masks
    const int32_t* seg_q;      // [B,Sq] or null
    const int32_t* seg_k;      // [B,Sk] or null
    const uint8_t* key_valid;  // [B,Sk] or null (0 = padded key)
    // element strides (D is contiguous)
    int64_t q_sb, q_ss, q_sh;
    int64_t k_sb, k_ss, k_sh;
    int64_t v_sb, v_ss, v_sh;
    int64_t o_sb, o_ss, o_sh;    // out, dout, dq share q's logical shape
    int64_t do_sb, do_ss, do_sh;
    int64_t dq_sb, dq_ss, dq_sh;
    int64_t dk_sb, dk_ss, dk_sh;
    int64_t dv_sb, dv_ss, dv_sh;
    int32_t B, H, Sq, Sk;
    int64_t q_start, k_start;  // global token position of row 0 (ring offset)
    // Piecewise position maps (a shard under zigzag / balanced ownership is a few runs of consecutive positions; so is the
    // K/V a rank gathers from its peers): piece i = rows [row[i], row[i+1]) at positions pos[i] + (r - row[i]); row[0] = 0,
    // pos[0] = the operand's start; rows and positions ascend, pieces do not overlap; q_np / k_np >= 1 pieces are in use.
    // Cuts are multiples of 256 rows, so no workgroup tile straddles one.  (api.inc fills these from LwmAttnArgs.)
    int32_t q_np, k_np;
    int32_t q_row[kMaxPieces], k_row[kMaxPieces];
    int64_t q_pos[kMaxPieces], k_pos[kMaxPieces];
    float scale;               // softmax scale, 1/sqrt(D)
    int32_t causal;
    int32_t carry_in;          // merge with *_acc before writing
    int32_t final_out;         // write bf16 results (else f32 *_acc)
    int64_t dqa_sb, dqa_ss, dqa_sh;   // element strides of dq_acc: [B,Sq,H,D] or head-major [B,H,Sq,D]
    // forward only: dense boolean mask and split-K (see include/lwm_hip.h)
    const uint8_t* dense_mask;
    int64_t msk_sb, msk_sq;
    int32_t k_splits;
    // block-sparsity hints (packed sequences), see include/lwm_hip.h
    const int32_t* segb_q;
    const int32_t* segb_k;
};

// ---- position maps (see AttnParams): everything below is wave-uniform scalar arithmetic over the piece tables in the
// kernel arguments, read with scalar loads where they are needed (prologue; the rare masked tile) -- nothing of them
// stays in registers across the tile loops.
struct PosMap {
    const int32_t* row;
    const int64_t* pos;
    int32_t n, S;
};
LWM_DEVICE PosMap q_map(const AttnParams& p) { return PosMap{p.q_row, p.q_pos, p.q_np, p.Sq}; }
LWM_DEVICE PosMap k_map(const AttnParams& p) { return PosMap{p.k_row, p.k_pos, p.k_np, p.Sk}; }
// The prologue's copy of a table: every entry in scalar registers, loaded by ONE clause of kernel-argument loads behind
// one wait (the empty asm pins all 24 values at this point; left to itself hipcc loads each entry where it is first used,
// behind its own wait and often its own branch: 46 scalar loads, 31 waits and 34 branches before a workgroup's first
// barrier -- 3000 to 5000 cycles, 3-5 % of a 32-step walk).  Dead after the prologue: nothing of it is live in a tile loop.
struct PosTab {
    int32_t row[kMaxPieces];
    int64_t pos[kMaxPieces];
    int32_t S;
};
LWM_DEVICE PosTab load_postab(const PosMap& m) {
    PosTab t;
    t.S = m.S;
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
        t.row[i] = m.row[i];
        t.pos[i] = m.pos[i];
    }
#ifndef LWM_EMU
    asm volatile("" : "+s"(t.row[0]), "+s"(t.row[1]), "+s"(t.row[2]), "+s"(t.row[3]), "+s"(t.row[4]), "+s"(t.row[5]), "+s"(t.row[6]),
                      "+s"(t.row[7]), "+s"(t.pos[0]), "+s"(t.pos[1]), "+s"(t.pos[2]), "+s"(t.pos[3]), "+s"(t.pos[4]), "+s"(t.pos[5]),
                      "+s"(t.pos[6]), "+s"(t.pos[7]));
#endif
    return t;
}
// position of row r = pos_base(t, row0) + r for every row r of a tile that begins at row0 (unused entries: row = INT32_MAX)
LWM_DEVICE int64_t pos_base(const PosTab& t, int row0) {
    int64_t base = t.pos[0];
#pragma unroll
    for (int i = 1; i < kMaxPieces; ++i) base = row0 >= t.row[i] ? t.pos[i] - t.row[i] : base;
    return base;
}
// How many LEADING tiles (of `per` rows; n_tiles = ceil(S / per)) begin at a position <= P -- positions ascend with the
// row, so those are the tiles a query at P can see a key of (or, mirrored, the query steps that lie wholly before a key
// at P + 1 when asked with P = key - per).  Branch-free over the register copy.
LWM_DEVICE int tiles_reaching(const PosTab& t, int per, int n_tiles, int64_t P) {
    int total = 0;
    bool open = true;       // every piece so far was reached to its end
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
        const int r0 = i == 0 ? 0 : t.row[i];
        const int rn = i + 1 < kMaxPieces ? t.row[i + 1] : 0x7fffffff;
        const bool live = r0 < t.S;
        const int r1 = rn < t.S ? rn : t.S;
        const int t_i = live ? (r1 - r0 + per - 1) / per : 0;
        const int64_t d = P - t.pos[i];
        const int64_t c = d < 0 ? 0 : d / per + 1;
        const int take = (open & live) ? (int)(c < t_i ? c : t_i) : 0;
        total += take;
        open = open & (take == t_i);
    }
    return total < n_tiles ? total : n_tiles;
}
// ... whose EVERY row lies at a position <= P
LWM_DEVICE int tiles_below(const PosTab& t, int per, int n_tiles, int64_t P) { return tiles_reaching(t, per, n_tiles, P - (per - 1)); }
// A cursor over the pieces for a walk that moves monotonically through the rows: rows [lo, hi) sit at base + row.
// seek() is called where a position is needed (a masked tile); it costs a compare when the row is still inside the piece.
struct PosCursor {
    int64_t base;
    int32_t lo, hi, i;
};
LWM_DEVICE PosCursor cursor_begin(const PosTab& t) { return PosCursor{t.pos[0], 0, t.row[1] < t.S ? t.row[1] : t.S, 0}; }
LWM_DEVICE void cursor_seek(const PosMap& m, PosCursor& c, int row) {
    while (row >= c.hi && c.i + 1 < m.n) {
        ++c.i;
        c.lo = m.row[c.i];
        c.hi = c.i + 1 < m.n ? m.row[c.i + 1] : m.S;
        c.base = m.pos[c.i] - c.lo;
    }
    while (row < c.lo && c.i > 0) {
        --c.i;
        c.hi = c.lo;
        c.lo = m.row[c.i];
        c.base = m.pos[c.i] - c.lo;
    }
}

LWM_DEVICE int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
LWM_DEVICE uint32_t tile_off(int row, int slot) {
    return (uint32_t)(row * kRowBytes + ((slot ^ swz(row)) << 4));
}

// ---- per-lane fragment addresses ------------------------------------------
// All are relative to a tile's LDS base and depend only on the lane, so a
// kernel computes them ONCE (8 + 8 VGPRs) and every read in the tile loop is
// `base + compile-time constant` = one ds_read with an immediate offset:
// adding a multiple of 16 rows (4096 B) to the row never changes swz().

// Row fragment (ds_read_b128): the MFMA operand's non-contracted index is the
// tile row, the contracted index is d.  Lane l, k-step s reads
// tile[row0 + (l&31)][16*s + 8*(l>>5) + 0..7]  at  frag_rows_addr(...)[s] + row0*256
// (row0 a multiple of 16).
struct RowFragAddr { uint32_t a[8]; };
LWM_DEVICE RowFragAddr frag_rows_addr(lds_t base, int row_in_16x, int l31, int hi) {
    RowFragAddr r;
    for (int s = 0; s < 8; ++s) r.a[s] = base + tile_off(row_in_16x + l31, 2 * s + hi);
    return r;
}

// Compact form: k-step s of a row fragment differs from step 0 only in address
// bits 5..7 (tile bases are 256-byte aligned, so the swizzled slot is an XOR on
// the low byte): a[s] == a[0] ^ (s << 5); likewise lo/up[db] == lo/up[0] ^ (db << 6).
// One register + one v_xor per read instead of 8 registers per fragment family.
LWM_DEVICE uint32_t row_frag_at(uint32_t a0, int s) { return a0 ^ (uint32_t)(s << 5); }
LWM_DEVICE bf16x8 read_tr_frag_x(uint32_t lo0, uint32_t up0, int db, uint32_t const_off) {
    bf16x4 lo = lds_read_tr16((lo0 ^ (uint32_t)(db << 6)) + const_off);
    bf16x4 up = lds_read_tr16((up0 ^ (uint32_t)(db << 6)) + const_off);
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = up[0]; o[5] = up[1]; o[6] = up[2]; o[7] = up[3];
    return o;
}

// Transposed fragment (2 x ds_read_b64_tr_b16): the operand's non-contracted
// index is d, the contracted index is the tile row.  Lane l (d = 32*db + (l&31))
// gets, for j = 0..7, tile[row0 + 4*(l>>5) + (j&3) + 8*(j>>2)][d] -- exactly the
// order in which a 32x32 C/D fragment holds its rows in registers 8t..8t+7, so
// a C/D fragment converted to bf16 is the other operand with no lane traffic.
// lo[db] addresses rows row0+4hi+0..3, up[db] rows row0+8+4hi+0..3; row0 a
// multiple of 16 is added as row0*256.
struct TrFragAddr { uint32_t lo[4], up[4]; };
LWM_DEVICE TrFragAddr frag_tr_addr(lds_t base, int lane) {
    TrFragAddr r;
    const int g = lane >> 4, i = lane & 15, hi = g >> 1;
    const int row = 4 * hi + (i >> 2);
    for (int db = 0; db < 4; ++db) {
        const int d = 32 * db + 16 * (g & 1) + 4 * (i & 3);
        r.lo[db] = base + tile_off(row, d >> 3) + (d & 7) * 2;
        r.up[db] = base + tile_off(row + 8, d >> 3) + (d & 7) * 2;
    }
    return r;
}
LWM_DEVICE bf16x8 read_tr_frag(const TrFragAddr& t, int db, uint32_t const_off) {
    bf16x4 lo = lds_read_tr16(t.lo[db] + const_off);
    bf16x4 up = lds_read_tr16(t.up[db] + const_off);
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = up[0]; o[5] = up[1]; o[6] = up[2]; o[7] = up[3];
    return o;
}

// Row index (within a 32-row block) that register r of a C/D fragment holds.
LWM_DEVICE int cd_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

LWM_DEVICE bf16x8 zero_bf16x8() {
    bf16x8 z;
    for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.0f;
    return z;
}

LWM_DEVICE f32x16 zero_f32x16() {
    f32x16 z;
    for (int j = 0; j < 16; ++j) z[j] = 0.0f;
    return z;
}

// 8 f32 (regs base..base+7 of a C/D fragment) -> bf16x8 operand.
LWM_DEVICE bf16x8 cvt_frag(const f32x16& x, int base) {
    bf16x8 o;
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)x[base + j];
    return o;
}

// ---- epilogue staging.  A wave's 32 x 128 f32 result tile sits in its C/D fragments with one ROW per lane pair: lane
// (l31, hi) holds, for d block db and register quad rq, the four columns 32 db + 8 rq + 4 hi + 0..3 of row l31 -- stored
// from there a wave instruction touches 32 rows x 32 bytes (8.9 k cycles for the 128 KiB of a dK/dV workgroup's f32
// partials).  Through LDS the tile leaves as whole rows: written at a row stride of 528 bytes (16 bytes of padding: the
// 16 lanes of a ds_write_b128 pass hit 16 x 4 distinct banks), read back two rows per instruction -- lanes 0..31 the 512
// contiguous bytes of row 2 i, lanes 32..63 of row 2 i + 1.  Same values, same roundings: only the order of the stores.
constexpr int kEpiRowBytes = kHeadDim * 4 + 16;
constexpr int kEpiTileBytes = 32 * kEpiRowBytes;       // 16 896 B per wave
LWM_DEVICE void epi_tile_write(lds_t tb, const f32x16 (&acc)[4], float scale, int l31, int hi) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            lds_write_f32x4(tb + (uint32_t)(l31 * kEpiRowBytes + (32 * db + 8 * rq + 4 * hi) * 4),
                            f32x4{acc[db][4 * rq + 0] * scale, acc[db][4 * rq + 1] * scale, acc[db][4 * rq + 2] * scale,
                                  acc[db][4 * rq + 3] * scale});
}
// read instruction i (0..15): row 2 i + (lane >> 5) of the tile, columns 4 (lane & 31) + 0..3
LWM_DEVICE f32x4 epi_tile_read(lds_t tb, int i, int lane) {
    return lds_read_f32x4(tb + (uint32_t)((2 * i + (lane >> 5)) * kEpiRowBytes + (lane & 31) * 16));
}

// ---- packed sequences: narrow a tile loop to the tiles whose segment range can meet
// the workgroup's own.  `blk` = (min,max) per 32-row block of the OTHER operand (one batch
// row), a tile = `per` consecutive blocks; the workgroup's own range is [smin, smax].
// All NT threads scan [t0, t1) cooperatively; returns the smallest enclosing [lo, hi).
// Tiles inside [lo, hi) may still be fully masked (non-monotone segment ids): the
// per-element mask stays authoritative, this only removes work that cannot contribute.
template <int NT>
LWM_DEVICE void seg_narrow(const int32_t* blk, int nblk, int per, int t0, int t1, int smin, int smax,
                           lds_t scratch, int tid, int& lo, int& hi) {
    int mylo = 0x7fffffff, myhi = -1;
    for (int t = t0 + tid; t < t1; t += NT) {
        int kmin = 0x7fffffff, kmax = (int)0x80000000;
        for (int j = 0; j < per; ++j) {
            const int bi = t * per + j;
            if (bi < nblk) {
                const int a = blk[2 * bi], b = blk[2 * bi + 1];
                kmin = a < kmin ? a : kmin;
                kmax = b > kmax ? b : kmax;
            }
        }
        if (kmax >= smin && kmin <= smax) {
            mylo = t < mylo ? t : mylo;
            myhi = t > myhi ? t : myhi;
        }
    }
    for (int m = 1; m < 64; m <<= 1) {
        const int a = shfl_xor_i(mylo, m), b = shfl_xor_i(myhi, m);
        mylo = a < mylo ? a : mylo;
        myhi = b > myhi ? b : myhi;
    }
    if ((tid & 63) == 0) {
        lds_write_i32(scratch + (tid >> 6) * 8, mylo);
        lds_write_i32(scratch + (tid >> 6) * 8 + 4, myhi);
    }
    block_sync();
    lo = 0x7fffffff;
    hi = -1;
    for (int w = 0; w < NT / 64; ++w) {
        const int a = lds_read_i32(scratch + w * 8), b = lds_read_i32(scratch + w * 8 + 4);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    block_sync();
    hi = hi + 1;            // exclusive; lo > hi-1 means "nothing"
    if (lo >= hi) { lo = t0; hi = t0; }
}

// Packed sequences, per step of 64 staged rows of the OTHER operand: do all of them carry the wave's own segment?  Then
// the per-element segment test of the step (4 LDS reads + 32 compare / select per 32 x 32 unit) is skipped and only the
// causal test remains where the unit touches the diagonal -- deep inside a document a packed batch then costs what a
// dense one costs.  `words` = the step's 64 staged segment words (kSegInvalid marks padded / out-of-range rows and
// never matches); own_uniform = all rows the wave owns carry `own_seg`.  Wave-uniform answer.
// (two parts, so that a phase of MFMAs can stand between the LDS read and the vote)
LWM_DEVICE int32_t seg_step_word(lds_t words, int lane) { return lds_read_i32(words + (uint32_t)lane * 4); }
LWM_DEVICE bool seg_step_uniform(int32_t word, bool own_uniform, int32_t own_seg) { return own_uniform && !wave_any(word != own_seg); }

// (min, max) over blocks [b0, b0+n) of a (min,max) block table
LWM_DEVICE void seg_own_range(const int32_t* blk, int nblk, int b0, int n, int& smin, int& smax) {
    smin = 0x7fffffff;
    smax = (int)0x80000000;
    for (int j = 0; j < n; ++j)
        if (b0 + j < nblk) {
            const int a = blk[2 * (b0 + j)], b = blk[2 * (b0 + j) + 1];
            smin = a < smin ? a : smin;
            smax = b > smax ? b : smax;
        }
}

// ---- per-phase cycle accounting (only in -DLWM_PROF builds: scripts/build_prof.sh).
// PROF_T(i) stamps s_memtime into slot i; PROF_ADD(dst, a, b) accumulates t[b]-t[a].  Lane 0 of
// every wave of ONE chosen workgroup writes its sums to AttnParams::out_acc (unused by the
// instrumented launch).  s_memtime drains lgkmcnt, so the stamps perturb the schedule by
// ~10 %; the numbers rank phases, they are not a clock.
#ifdef LWM_PROF
struct ProfAcc { unsigned long long v[8]; };
#define PROF_DECL(n) unsigned long long prof_t[n]
#define PROF_T(i) prof_t[i] = __builtin_amdgcn_s_memtime()
#define PROF_ADD(acc, slot, a, b) (acc).v[slot] += prof_t[b] - prof_t[a]
#define PROF_KEEP(x) asm volatile("" ::"v"(x))
#else
struct ProfAcc {};
#define PROF_DECL(n)
#define PROF_T(i)
#define PROF_ADD(acc, slot, a, b)
#define PROF_KEEP(x)
#endif

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int32_t kSegInvalid = (int32_t)0x80000000;

}  // namespace lwm
