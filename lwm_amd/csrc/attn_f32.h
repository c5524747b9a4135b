// attn_f32.h -- blockwise attention forward + backward with FLOAT32 operands on gfx950: the `--dtype=fp32` flavour
// of the training op (the reference's default, lwm/train.py:36; BASELINE configs[0] runs the model in fp32; SURVEY.md
// section 8a: "q,k,v,out dtype = --dtype: bf16 in configs 2-5, fp32 in #1").  Requires wave_ops.h + attn_common.h +
// attn_bwd.h (the layout of the backward's row statistics).
//
// Same op, same masks, same carries as the bf16 kernels (attn_fwd64.h / attn_bwd64.h; call site lwm/llama.py:539-569,
// mask specification lwm/llama.py:572-592), with every contraction on the exact-f32 matrix
// instruction v_mfma_f32_32x32x2_f32 (one rounding per product: wave_ops.h) -- no operand is ever rounded to bf16, which
// is what lets the parity tests hold this path to 1e-5 of the fp64 oracle instead of the bf16 path's 8e-3.
//
// Geometry (all three kernels): workgroup = 4 waves; a wave OWNS 32 rows of the stationary operand, held in registers
// in MFMA B-operand form (64 floats per lane and tensor); the other operand streams through LDS in tiles of 32 rows
// (rows padded to 528 bytes: the 16 lanes of a ds_read_b128 pass then hit 64 distinct banks), fetched one tile ahead
// into registers while the current tile is computed.  All products are computed TRANSPOSED where that makes the
// C/D fragment of one product the B operand of the next with no lane traffic (the trick of attn_common.h's
// "transposed fragment", here with k = 2: register i of a C/D fragment holds row 8 (i / 4) + 4 hi + i % 4, and the
// MFMA that consumes it contracts over exactly that row for lane half hi):
//
//   forward  (owns 32 queries/wave, streams K,V)   S^T = K Q^T;  P^T = exp2(S^T c - m);  O^T += V^T P^T
//   dQ       (owns 32 queries/wave, streams K,V)   S^T = K Q^T;  dP'^T = V dO^T - delta;  dS^T = P^T dP'^T;  dQ^T += K^T dS^T
//   dK/dV    (owns 32 keys/wave,  streams Q,dO)    S = Q K^T;    dP' = dO V^T - delta;    dS = P dP';  dV^T += dO^T P;  dK^T += Q^T dS
//
// Roofline: MFMA, f32 (157.3 TF/s dense: 256 FLOP per cycle and CU).  Algorithmic work as the bf16 path's (SURVEY.md
// section 8d); executed GEMM units 2 / 3 / 4.  This is the plumbing / parity flavour, not the headline: no LDS-DMA,
// no hand-ordered instruction stream -- hipcc schedules it.  Single-piece position maps only (q_start / k_start);
// dense_mask, k_splits and the block-sparsity hints are not read.
#pragma once

namespace lwm {

constexpr int kX32Threads = 256;
constexpr int kX32Own = 128;                         // stationary rows per workgroup (4 waves x 32)
constexpr int kX32BT = 32;                           // streamed rows per tile
constexpr int kX32RowBytes = kHeadDim * 4 + 16;      // 528: padded LDS row of 128 floats
constexpr int kX32TileBytes = kX32BT * kX32RowBytes; // 16 896
constexpr int kX32OffMeta = 2 * kX32TileBytes;       // after the two streamed tiles: 3 x 32 words of row meta
constexpr int kX32MetaBytes = 3 * kX32BT * 4;
constexpr int kX32LdsBytes = 4 * kEpiTileBytes > kX32OffMeta + kX32MetaBytes ? 4 * kEpiTileBytes : kX32OffMeta + kX32MetaBytes;
static_assert(kEpiRowBytes == kX32RowBytes, "the epilogue staging of attn_common.h uses the same padded row");

// the value must EXIST here (its LDS read issued, not sunk below the MFMAs that follow): an empty statement hipcc cannot move
#ifdef LWM_EMU
LWM_DEVICE void x32_pin(f32x4&) {}
#else
LWM_DEVICE void x32_pin(f32x4& v) { asm volatile("" : "+v"(v)); }
#endif

// Workgroup -> (tile, head, batch row), TILE-major: under a causal mask a workgroup's walk grows with its query tile
// (shrinks with its key block) from a few steps to S / 32, and the heaviest workgroup alone is a quarter of a launch at
// S = 8192 -- handed out head by head, the last head's long walks start when most of the chip has run dry (measured:
// 63 % of the wave slots filled on average).  All heads' longest walks go first, the short ones fill the tail.
LWM_DEVICE void x32_block_to_tile(int bid, int n_tiles, int H, int B, bool descending, int& tile, int& h, int& b) {
    const int nbh = H * B, order = bid / nbh, bh = bid % nbh;
    tile = descending ? n_tiles - 1 - order : order;
    h = bh % H;
    b = bh / H;
}

struct X32Frag { float v[64]; };       // B-operand form of 32 rows x 128: v[4 j + c] = row (lane & 31), column 8 j + 4 hi + c

// rows [row0, row0 + 32) of a [.., S, .., 128] f32 tensor as a wave's B-operand fragments (rows >= S: zeros)
LWM_DEVICE void x32_load_frag(X32Frag& f, const float* base, int64_t stride_s, int row, int S, int hi) {
    const bool ok = row < S;
    const float* r = base + (int64_t)(ok ? row : 0) * stride_s + 4 * hi;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        f32x4 t = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (ok) t = global_load_f32x4(r + 8 * j);
        f.v[4 * j + 0] = t[0]; f.v[4 * j + 1] = t[1]; f.v[4 * j + 2] = t[2]; f.v[4 * j + 3] = t[3];
    }
}

// One streamed tile (32 rows x 128 floats) per tensor: thread t fetches the 16-byte pieces t, t + 256, t + 512, t + 768
// (piece i = row i / 32, columns 4 (i % 32) ..), rows past S as zeros (a masked score times a stale NaN is still a NaN).
struct X32Pre { f32x4 a[4], b[4]; };
LWM_DEVICE void x32_fetch(X32Pre& pre, const float* ta, int64_t sa, const float* tb, int64_t sb, int row0, int S, int tid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j;
        const int row = row0 + (i >> 5), c4 = (i & 31) * 4;
        pre.a[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        pre.b[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (row < S) {
            pre.a[j] = global_load_f32x4(ta + (int64_t)row * sa + c4);
            pre.b[j] = global_load_f32x4(tb + (int64_t)row * sb + c4);
        }
    }
}
LWM_DEVICE void x32_commit(const X32Pre& pre, lds_t lds, int tid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j;
        const uint32_t off = (uint32_t)((i >> 5) * kX32RowBytes + (i & 31) * 16);
        lds_write_f32x4(lds + off, pre.a[j]);
        lds_write_f32x4(lds + kX32TileBytes + off, pre.b[j]);
    }
}

// acc(32 x 32) += tile rows (A operand, from LDS: row = lane & 31, 4 consecutive columns per read) x the wave's fragments.
// The read of step j + 1 is issued before the four MFMAs of step j (left to itself hipcc puts each ds_read directly in
// front of its first MFMA with a full wait in between: the LDS latency, once per 128 cycles of matrix pipe).
LWM_DEVICE f32x16 x32_rows_times_frag(lds_t tile, const X32Frag& f, f32x16 acc, int l31, int hi) {
    const lds_t a0 = tile + (uint32_t)(l31 * kX32RowBytes + hi * 16);
    f32x4 t = lds_read_f32x4(a0), n1 = lds_read_f32x4(a0 + 32u);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        f32x4 n2 = n1;
        if (j + 2 < 16) n2 = lds_read_f32x4(a0 + (uint32_t)((j + 2) * 32));
        acc = mfma_32x32x2_f32(t[0], f.v[4 * j + 0], acc);
        acc = mfma_32x32x2_f32(t[1], f.v[4 * j + 1], acc);
        acc = mfma_32x32x2_f32(t[2], f.v[4 * j + 2], acc);
        acc = mfma_32x32x2_f32(t[3], f.v[4 * j + 3], acc);
        x32_pin(n1);
        t = n1;
        n1 = n2;
    }
    return acc;
}

// out^T(128 x 32) += tile^T x w, w a C/D fragment over (tile row, owned row): register i of w multiplies tile row
// cd_row(i, hi); the A operand is column 32 db + (lane & 31) of that tile row (32 consecutive floats per half wave).
// Reads run two steps ahead of the MFMAs, as above.
LWM_DEVICE f32x4 x32_col4(lds_t ar) {
    return f32x4{lds_read_f32(ar), lds_read_f32(ar + 128u), lds_read_f32(ar + 256u), lds_read_f32(ar + 384u)};
}
LWM_DEVICE void x32_tileT_times_cd(lds_t tile, const f32x16& w, f32x16 (&out)[4], int l31, int hi) {
    const lds_t a0 = tile + (uint32_t)(4 * hi * kX32RowBytes + l31 * 4);
    f32x4 t = x32_col4(a0), n1 = x32_col4(a0 + (uint32_t)kX32RowBytes);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        f32x4 n2 = n1;
        if (i + 2 < 16) n2 = x32_col4(a0 + (uint32_t)((((i + 2) & 3) + 8 * ((i + 2) >> 2)) * kX32RowBytes));
#pragma unroll
        for (int db = 0; db < 4; ++db) out[db] = mfma_32x32x2_f32(t[db], w[i], out[db]);
        x32_pin(n1);
        t = n1;
        n1 = n2;
    }
}

// the 16 words of a 32-word LDS row table that belong to the rows of a C/D fragment's registers (row = cd_row(i, hi))
LWM_DEVICE void x32_cd_words_i(lds_t tab, int hi, int32_t (&o)[16]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const u32x4 t = lds_read_u32x4(tab + (uint32_t)((8 * a + 4 * hi) * 4));
        o[4 * a + 0] = (int32_t)t[0]; o[4 * a + 1] = (int32_t)t[1]; o[4 * a + 2] = (int32_t)t[2]; o[4 * a + 3] = (int32_t)t[3];
    }
}
LWM_DEVICE void x32_cd_words_f(lds_t tab, int hi, float (&o)[16]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x4 t = lds_read_f32x4(tab + (uint32_t)((8 * a + 4 * hi) * 4));
        o[4 * a + 0] = t[0]; o[4 * a + 1] = t[1]; o[4 * a + 2] = t[2]; o[4 * a + 3] = t[3];
    }
}

// segment word of a key / query row with everything that can hide the row folded in: kSegInvalid never matches
LWM_DEVICE int32_t x32_seg_word(const int32_t* seg, const uint8_t* valid, int64_t brow, int row, int S) {
    if (row >= S) return kSegInvalid;
    if (valid && valid[brow + row] == 0) return kSegInvalid;
    return seg ? seg[brow + row] : 0;
}

// a wave's 128 x 32 transposed result, leaving as whole rows through its epilogue tile: value * scale (+ carry) -> dst
LWM_DEVICE void x32_store_rows(lds_t tb, const f32x16 (&acc)[4], float scale, const float* carry, int64_t c_ss, float* dst,
                               int64_t d_ss, int row0, int S, int lane) {
    epi_tile_write(tb, acc, scale, lane & 31, lane >> 5);
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = row0 + 2 * i + (lane >> 5);
        f32x4 t = epi_tile_read(tb, i, lane);
        if (row < S) {
            if (carry) {
                const f32x4 c = global_load_f32x4(carry + (int64_t)row * c_ss + (lane & 31) * 4);
                t[0] += c[0]; t[1] += c[1]; t[2] += c[2]; t[3] += c[3];
            }
            global_store_f32x4(dst + (int64_t)row * d_ss + (lane & 31) * 4, t);
        }
    }
}

// ------------------------------------------------------------------ forward
LWM_KERNEL(kX32Threads) void attn_fwd_f32_kernel(AttnParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx(), wave = wave_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int nqt = (p.Sq + kX32Own - 1) / kX32Own;
    int qt, h, b;
    x32_block_to_tile(block_idx_x(), nqt, p.H, p.B, p.causal != 0, qt, h, b);     // the longest walks (last queries) first
    const int q0 = qt * kX32Own + wave * 32, row = q0 + l31;
    const bool row_ok = row < p.Sq;
    const float* Q = (const float*)p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const float* K = (const float*)p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const float* V = (const float*)p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const int64_t bh = (int64_t)b * p.H + h;
    const int64_t acc_ss = (int64_t)p.H * kHeadDim;
    float* acc_base = p.out_acc ? p.out_acc + ((int64_t)b * p.Sq * p.H + h) * kHeadDim : nullptr;

    X32Frag qf;
    x32_load_frag(qf, Q, p.q_ss, row, p.Sq, hi);
    f32x16 o[4];
    for (int db = 0; db < 4; ++db) o[db] = zero_f32x16();
    float m = -INFINITY, l = 0.0f;
    if (p.carry_in && row_ok) {      // the carry is a normalised partial output and its natural-log lse: mass 1 at m = lse
        const float lc = p.lse_acc[bh * p.Sq + row];
        if (lc != -INFINITY) {
            m = lc * kLog2e;
            l = 1.0f;
            for (int db = 0; db < 4; ++db)
                for (int a = 0; a < 4; ++a) {
                    const f32x4 t = global_load_f32x4(acc_base + (int64_t)row * acc_ss + 32 * db + 8 * a + 4 * hi);
                    o[db][4 * a + 0] = t[0]; o[db][4 * a + 1] = t[1]; o[db][4 * a + 2] = t[2]; o[db][4 * a + 3] = t[3];
                }
        }
    }
    const int64_t q_pos = p.q_start + row;
    const int32_t seg_own = x32_seg_word(p.seg_q, nullptr, (int64_t)b * p.Sq, row, p.Sq);
    const float c2 = p.scale * kLog2e;

    // keys the workgroup can see
    int n_kt = (p.Sk + kX32BT - 1) / kX32BT;
    const int wg_last = (qt * kX32Own + kX32Own - 1 < p.Sq - 1) ? qt * kX32Own + kX32Own - 1 : p.Sq - 1;
    const int wave_last = (q0 + 31 < p.Sq - 1) ? q0 + 31 : p.Sq - 1;
    if (p.causal) {
        const int64_t d = p.q_start + wg_last - p.k_start;       // last visible key row
        const int64_t e = d < 0 ? 0 : d / kX32BT + 1;
        n_kt = e < n_kt ? (int)e : n_kt;
    }
    X32Pre pre;
    if (n_kt > 0) x32_fetch(pre, K, p.k_ss, V, p.v_ss, 0, p.Sk, tid);
    for (int kt = 0; kt < n_kt; ++kt) {
        block_sync();
        x32_commit(pre, lds, tid);
        if (tid < kX32BT)
            lds_write_i32(lds + kX32OffMeta + (uint32_t)tid * 4,
                          x32_seg_word(p.seg_k, p.key_valid, (int64_t)b * p.Sk, kt * kX32BT + tid, p.Sk));
        block_sync();
        if (kt + 1 < n_kt) x32_fetch(pre, K, p.k_ss, V, p.v_ss, (kt + 1) * kX32BT, p.Sk, tid);
        const int64_t kpos0 = p.k_start + (int64_t)kt * kX32BT;
        if (q0 >= p.Sq || (p.causal && kpos0 > p.q_start + wave_last)) continue;   // wave-uniform: nothing of this tile is visible
        f32x16 s = x32_rows_times_frag(lds, qf, zero_f32x16(), l31, hi);            // S^T[key][q]
        int32_t segw[16];
        x32_cd_words_i(lds + kX32OffMeta, hi, segw);
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = segw[i] == seg_own && (!p.causal || kpos0 + cd_row(i, hi) <= q_pos);
            s[i] = ok ? s[i] * c2 : -INFINITY;
            mx = s[i] > mx ? s[i] : mx;
        }
        const float mo = xhalf(mx);
        mx = mo > mx ? mo : mx;
        const float m_new = mx > m ? mx : m;
        const float m_safe = m_new == -INFINITY ? 0.0f : m_new;
        const float alpha = fast_exp2(m - m_safe);
        float rs = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = fast_exp2(s[i] - m_safe);
            rs += s[i];
        }
        rs += xhalf(rs);
        l = l * alpha + rs;
        m = m_new;
        if (wave_any(alpha != 1.0f)) {      // (a running maximum that did not move leaves alpha = 1 exactly: nothing to rescale)
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
        }
        x32_tileT_times_cd(lds + kX32TileBytes, s, o, l31, hi);                      // O^T += V^T P^T
    }
    block_sync();       // the tiles are dead: the epilogue reuses them
    if (q0 < p.Sq) {
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        const float lse = l > 0.0f ? (m + fast_log2(l)) * kLn2 : -INFINITY;
        float* dst;
        int64_t d_ss;
        if (p.final_out) {
            dst = (float*)p.out + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh;
            d_ss = p.o_ss;
            if (row_ok && hi == 0) p.lse[bh * p.Sq + row] = lse;
        } else {
            dst = acc_base;
            d_ss = acc_ss;
            if (row_ok && hi == 0) p.lse_acc[bh * p.Sq + row] = lse;
        }
        x32_store_rows(lds + (uint32_t)(wave * kEpiTileBytes), o, inv, nullptr, 0, dst, d_ss, q0, p.Sq, lane);
    }
}

// ------------------------------------------------------------------ backward: row statistics (the layout of attn_bwd.h)
LWM_KERNEL(kDeltaThreads) void attn_bwd_delta_f32_kernel(AttnParams p, float* stats) {
    const int tid = thread_idx();
    const int64_t Sqp = bwd_stat_pad(p.Sq);
    const int64_t rows = (int64_t)p.B * p.H * Sqp;
    const int part = tid & 15;
    int64_t row = (int64_t)block_idx_x() * (kDeltaThreads / 16) + (tid >> 4);
    const int64_t row_step = (int64_t)grid_dim_x() * (kDeltaThreads / 16);
    const int64_t iters = (rows + row_step - 1) / row_step;
    const float* O = (const float*)p.out;
    const float* dO = (const float*)p.dout;
    for (int64_t it = 0; it < iters; ++it, row += row_step) {
        float s = 0.0f;
        const int64_t q = row % Sqp, bh = row / Sqp;
        const bool ok = row < rows && q < p.Sq;
        if (ok) {
            const int64_t h = bh % p.H, b = bh / p.H;
            const float* o = O + b * p.o_sb + q * p.o_ss + h * p.o_sh + part * 8;
            const float* d = dO + b * p.do_sb + q * p.do_ss + h * p.do_sh + part * 8;
            const f32x4 o0 = global_load_f32x4(o), o1 = global_load_f32x4(o + 4);
            const f32x4 d0 = global_load_f32x4(d), d1 = global_load_f32x4(d + 4);
            for (int j = 0; j < 4; ++j) s = fmaf(o0[j], d0[j], s);
            for (int j = 0; j < 4; ++j) s = fmaf(o1[j], d1[j], s);
        }
        s += shfl_xor_f(s, 1);
        s += shfl_xor_f(s, 2);
        s += shfl_xor_f(s, 4);
        s += shfl_xor_f(s, 8);
        if (row < rows && part == 0) {
            float nl2 = -INFINITY;
            if (ok) {
                const float lg = p.lse[bh * p.Sq + q];
                nl2 = (lg == -INFINITY) ? -INFINITY : -lg * kLog2e;
            }
            stats[bwd_stat_row(bh, Sqp) + q] = nl2;
            stats[bwd_stat_row(bh, Sqp) + Sqp + q] = ok ? -s : 0.0f;
        }
    }
}

// ------------------------------------------------------------------ backward: dQ (a workgroup owns 128 queries)
LWM_KERNEL(kX32Threads) void attn_bwd_dq_f32_kernel(AttnParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx(), wave = wave_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int nqt = (p.Sq + kX32Own - 1) / kX32Own;
    int qt, h, b;
    x32_block_to_tile(block_idx_x(), nqt, p.H, p.B, p.causal != 0, qt, h, b);     // the longest walks (last queries) first
    const int q0 = qt * kX32Own + wave * 32, row = q0 + l31;
    const float* Q = (const float*)p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const float* K = (const float*)p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const float* V = (const float*)p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const float* dO = (const float*)p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;
    const int64_t bh = (int64_t)b * p.H + h;
    const int64_t Sqp = bwd_stat_pad(p.Sq);

    X32Frag qf, dof;
    x32_load_frag(qf, Q, p.q_ss, row, p.Sq, hi);
    x32_load_frag(dof, dO, p.do_ss, row, p.Sq, hi);
    float nl2 = -INFINITY, nd = 0.0f;          // rows past Sq (inside the padding of the statistics) read -inf / 0 anyway
    if (row < Sqp) {
        nl2 = p.delta[bwd_stat_row(bh, Sqp) + row];
        nd = p.delta[bwd_stat_row(bh, Sqp) + Sqp + row];
    }
    f32x16 dq[4];
    for (int db = 0; db < 4; ++db) dq[db] = zero_f32x16();
    const int64_t q_pos = p.q_start + row;
    const int32_t seg_own = x32_seg_word(p.seg_q, nullptr, (int64_t)b * p.Sq, row, p.Sq);
    const float c2 = p.scale * kLog2e;

    int n_kt = (p.Sk + kX32BT - 1) / kX32BT;
    const int wg_last = (qt * kX32Own + kX32Own - 1 < p.Sq - 1) ? qt * kX32Own + kX32Own - 1 : p.Sq - 1;
    const int wave_last = (q0 + 31 < p.Sq - 1) ? q0 + 31 : p.Sq - 1;
    if (p.causal) {
        const int64_t d = p.q_start + wg_last - p.k_start;
        const int64_t e = d < 0 ? 0 : d / kX32BT + 1;
        n_kt = e < n_kt ? (int)e : n_kt;
    }
    X32Pre pre;
    if (n_kt > 0) x32_fetch(pre, K, p.k_ss, V, p.v_ss, 0, p.Sk, tid);
    for (int kt = 0; kt < n_kt; ++kt) {
        block_sync();
        x32_commit(pre, lds, tid);
        if (tid < kX32BT)
            lds_write_i32(lds + kX32OffMeta + (uint32_t)tid * 4,
                          x32_seg_word(p.seg_k, p.key_valid, (int64_t)b * p.Sk, kt * kX32BT + tid, p.Sk));
        block_sync();
        if (kt + 1 < n_kt) x32_fetch(pre, K, p.k_ss, V, p.v_ss, (kt + 1) * kX32BT, p.Sk, tid);
        const int64_t kpos0 = p.k_start + (int64_t)kt * kX32BT;
        if (q0 >= p.Sq || (p.causal && kpos0 > p.q_start + wave_last)) continue;
        f32x16 s = x32_rows_times_frag(lds, qf, zero_f32x16(), l31, hi);                 // S^T[key][q]
        f32x16 dp;
        for (int i = 0; i < 16; ++i) dp[i] = nd;                                         // dP'^T = V dO^T - delta
        dp = x32_rows_times_frag(lds + kX32TileBytes, dof, dp, l31, hi);
        int32_t segw[16];
        x32_cd_words_i(lds + kX32OffMeta, hi, segw);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = segw[i] == seg_own && (!p.causal || kpos0 + cd_row(i, hi) <= q_pos);
            const float pr = ok ? fast_exp2(fmaf(s[i], c2, nl2)) : 0.0f;
            s[i] = pr * dp[i];                                                           // dS^T (the scale joins at the end)
        }
        x32_tileT_times_cd(lds, s, dq, l31, hi);                                         // dQ^T += K^T dS^T
    }
    block_sync();
    if (q0 < p.Sq) {
        const float* carry = nullptr;
        if (p.carry_in) carry = p.dq_acc + (int64_t)b * p.dqa_sb + (int64_t)h * p.dqa_sh;
        float* dst;
        int64_t d_ss;
        if (p.final_out) {
            dst = (float*)p.dq + (int64_t)b * p.dq_sb + (int64_t)h * p.dq_sh;
            d_ss = p.dq_ss;
        } else {
            dst = p.dq_acc + (int64_t)b * p.dqa_sb + (int64_t)h * p.dqa_sh;
            d_ss = p.dqa_ss;
        }
        x32_store_rows(lds + (uint32_t)(wave * kEpiTileBytes), dq, p.scale, carry, p.dqa_ss, dst, d_ss, q0, p.Sq, lane);
    }
}

// ------------------------------------------------------------------ backward: dK, dV (a workgroup owns 128 keys)
LWM_KERNEL(kX32Threads) void attn_bwd_dkdv_f32_kernel(AttnParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx(), wave = wave_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int nkt = (p.Sk + kX32Own - 1) / kX32Own;
    int kb, h, b;
    x32_block_to_tile(block_idx_x(), nkt, p.H, p.B, false, kb, h, b);             // the longest walks (first keys) first
    const int k0 = kb * kX32Own + wave * 32, row = k0 + l31;
    const float* Q = (const float*)p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const float* K = (const float*)p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const float* V = (const float*)p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const float* dO = (const float*)p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;
    const int64_t bh = (int64_t)b * p.H + h;
    const int64_t Sqp = bwd_stat_pad(p.Sq);
    const float* st = p.delta + bwd_stat_row(bh, Sqp);

    X32Frag kf, vf;
    x32_load_frag(kf, K, p.k_ss, row, p.Sk, hi);
    x32_load_frag(vf, V, p.v_ss, row, p.Sk, hi);
    f32x16 dk[4], dv[4];
    for (int db = 0; db < 4; ++db) {
        dk[db] = zero_f32x16();
        dv[db] = zero_f32x16();
    }
    const int64_t k_pos = p.k_start + row;
    const int32_t seg_own = x32_seg_word(p.seg_k, p.key_valid, (int64_t)b * p.Sk, row, p.Sk);
    const float c2 = p.scale * kLog2e;

    const int n_qt = (p.Sq + kX32BT - 1) / kX32BT;
    int qt0 = 0;
    if (p.causal) {         // the first query tile that holds a position >= the workgroup's first key
        const int64_t d = p.k_start + (int64_t)kb * kX32Own - p.q_start;
        const int64_t e = d <= 0 ? 0 : d / kX32BT;
        qt0 = e < n_qt ? (int)e : n_qt;
    }
    const lds_t tab_nl2 = lds + kX32OffMeta, tab_nd = tab_nl2 + kX32BT * 4, tab_seg = tab_nd + kX32BT * 4;
    X32Pre pre;
    if (qt0 < n_qt) x32_fetch(pre, Q, p.q_ss, dO, p.do_ss, qt0 * kX32BT, p.Sq, tid);
    for (int qt = qt0; qt < n_qt; ++qt) {
        block_sync();
        x32_commit(pre, lds, tid);
        if (tid < kX32BT) {
            const int qr = qt * kX32BT + tid;          // inside the padded statistics: n_qt * 32 <= Sqp
            lds_write_f32(tab_nl2 + (uint32_t)tid * 4, st[qr]);
            lds_write_f32(tab_nd + (uint32_t)tid * 4, st[Sqp + qr]);
            lds_write_i32(tab_seg + (uint32_t)tid * 4, x32_seg_word(p.seg_q, nullptr, (int64_t)b * p.Sq, qr, p.Sq));
        }
        block_sync();
        if (qt + 1 < n_qt) x32_fetch(pre, Q, p.q_ss, dO, p.do_ss, (qt + 1) * kX32BT, p.Sq, tid);
        const int64_t qpos0 = p.q_start + (int64_t)qt * kX32BT;
        if (k0 >= p.Sk || (p.causal && p.k_start + k0 > qpos0 + kX32BT - 1)) continue;   // every key of the wave lies after the tile
        f32x16 s = x32_rows_times_frag(lds, kf, zero_f32x16(), l31, hi);                 // S[q][key]
        float nl2[16], nd[16];
        int32_t segw[16];
        x32_cd_words_f(tab_nl2, hi, nl2);
        x32_cd_words_f(tab_nd, hi, nd);
        x32_cd_words_i(tab_seg, hi, segw);
        f32x16 dp;
        for (int i = 0; i < 16; ++i) dp[i] = nd[i];                                      // dP' = dO V^T - delta
        dp = x32_rows_times_frag(lds + kX32TileBytes, vf, dp, l31, hi);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = segw[i] == seg_own && (!p.causal || k_pos <= qpos0 + cd_row(i, hi));
            const float pr = ok ? fast_exp2(fmaf(s[i], c2, nl2[i])) : 0.0f;
            s[i] = pr;
            dp[i] = pr * dp[i];
        }
        x32_tileT_times_cd(lds + kX32TileBytes, s, dv, l31, hi);                         // dV^T += dO^T P
        x32_tileT_times_cd(lds, dp, dk, l31, hi);                                        // dK^T += Q^T dS
    }
    block_sync();
    if (k0 < p.Sk) {
        const int64_t acc_ss = (int64_t)p.H * kHeadDim;
        const int64_t acc_off = ((int64_t)b * p.Sk * p.H + h) * kHeadDim;
        const float* ck = p.carry_in ? p.dk_acc + acc_off : nullptr;
        const float* cv = p.carry_in ? p.dv_acc + acc_off : nullptr;
        float *dstk, *dstv;
        int64_t k_ss, v_ss;
        if (p.final_out) {
            dstk = (float*)p.dk + (int64_t)b * p.dk_sb + (int64_t)h * p.dk_sh;
            dstv = (float*)p.dv + (int64_t)b * p.dv_sb + (int64_t)h * p.dv_sh;
            k_ss = p.dk_ss;
            v_ss = p.dv_ss;
        } else {
            dstk = p.dk_acc + acc_off;
            dstv = p.dv_acc + acc_off;
            k_ss = v_ss = acc_ss;
        }
        const lds_t tb = lds + (uint32_t)(wave * kEpiTileBytes);
        x32_store_rows(tb, dk, p.scale, ck, acc_ss, dstk, k_ss, k0, p.Sk, lane);
        wave_lds_fence();
        x32_store_rows(tb, dv, 1.0f, cv, acc_ss, dstv, v_ss, k0, p.Sk, lane);
    }
}

}  // namespace lwm
