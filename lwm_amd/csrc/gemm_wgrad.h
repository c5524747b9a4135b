// gemm_wgrad.h -- the weight gradient of a flax Dense kernel (lwm/llama.py:390-421, :631-655: y = x @ W, W (in, out)):
//
//     dW[K][N] = sum_s x[s][K] * g[s][N]          x, g bf16 with the FEATURE dimension contiguous, f32 accumulation
//
// -- the one GEMM of the LWM-7B step whose reduction dimension (S) is contiguous in NEITHER operand.  hipBLASLt runs it at
// 0.90-1.05 PF/s in this layout (1.1-1.2 once the narrow operand is transposed first, profiles/r06_model_full.md); here
// both operands stay where they are: their tiles arrive in LDS by LDS-DMA in the natural layout and BOTH MFMA operands are
// read as TRANSPOSED fragments (ds_read_b64_tr_b16) -- the read the dK/dV attention kernel uses for Q^T and dO^T
// (attn_common.h: frag_tr_addr; the k order inside a fragment is the C/D row order, the same permutation on both operands).
// Requires wave_ops.h, attn_common.h (tile geometry: 128-column tiles, XOR swizzle), attn_fwd64.h (f4_dma1, f4_mfma_o),
// attn_bwd64.h (d4_settle_acc4).
//
// Workgroup = 8 waves (two per SIMD, NW = 8) = a 256 x 256 tile of dW; wave (wm, wn) owns 64 x 128 of it: 2 x 4 accumulator
// tuples (128 AGPRs).  A stage = 32 rows of S: [x cols 0..127 | x cols 128..255 | g cols 0..127 | g cols 128..255], 8 KiB each;
// FOUR stages are in LDS (128 KiB) and the one barrier of a stage stands in its MIDDLE:
//
//     step 0 of stage i   8 MFMAs; the fragments of step 1 are requested behind them
//     middle              wait for this wave's pieces of stage i + 1 (those of stage i + 2 stay in flight), barrier
//     step 1 of stage i   8 MFMAs; the first fragments of stage i + 1 are requested behind them, and the wave's four
//                         pieces of stage i + 3, one per second MFMA gap, into the slot stage i - 1 used (every wave has
//                         left stage i - 1; four requests back to back hold an in-order wave for 250-700 cycles)
//
// so no wave ever meets the barrier with an empty matrix pipe behind it, and a piece has 1.5 stages (>= 1500 cycles) to
// land.  The request stream has no branch: past the end of its range a stage re-requests the LAST stage into a slot nobody
// reads again, so every stage issues the same instructions and waits on the same count (a uniform branch around each
// request and wait cost 5 %).  The same body serves one wave per SIMD (NW = 4: 128 x 128 per wave, 256 AGPRs, a third
// fewer fragment reads per MFMA -- 3 % behind: every stall of the only wave is the matrix pipe's; LWM_WGRAD_WAVES=4).
// Measured (profiles/r06_wgrad.md): 1.22-1.37 PF/s on the step's four shapes against 0.91-1.21 for the library on the same
// box, mfma_util 0.84, no LDS bank conflicts.  Skeletons of an earlier version on wqkv: 2.78 ms as it was; without the
// waits 2.73, without the barrier 2.69, without the fragment reads 2.38, without the requests 2.27, MFMAs alone 1.87
// (1.76 PF/s: what the clock allows under this load).
//
// Work split: tiles are numbered in bands of 8 tile columns, row-major inside a band, so that the 32 workgroups an XCD runs
// at a time form a 4 x 8 block of tiles (12 operand streams for 32 tiles share that XCD's L2).  The first
// floor(tiles / CUs) * CUs tiles take one workgroup each; the REST (a last, partly filled round: 5.4 rounds for w1|w3, 2.7
// for w2 at 256 CUs) is cut along S into equal stage ranges, one per CU (stream-K): a workgroup whose range is not a whole
// tile leaves its f32 partial in the workspace and wgrad_fixup_kernel adds the partials of a tile in a fixed order
// (deterministic; 66 MB of traffic for w1|w3).
#pragma once

namespace lwm {

#ifndef LWM_EMU
// (the host build brings its own)
LWM_DEVICE void lds_write_bf16(lds_t a, bf16_t v) { *LWM_LDS(bf16_t, a) = v; }
#endif

constexpr int kWgBM = 256, kWgBN = 256, kWgBK = 32;
constexpr int kWgThreads = 512;                     // (the 8-wave form; the 4-wave form runs 256)
constexpr int kWgSubBytes = kWgBK * kRowBytes;       // 8 KiB: a [32 s][128 columns] tile
constexpr int kWgSlotBytes = 4 * kWgSubBytes;        // x lo | x hi | g lo | g hi
constexpr int kWgSlots = 4;
constexpr int kWgLdsBytes = kWgSlots * kWgSlotBytes; // 128 KiB
constexpr int kWgTileFloats = kWgBM * kWgBN;         // one f32 partial

struct WgradParams {
    const bf16_t* x;
    const bf16_t* g;
    bf16_t* dw;
    float* ws;                  // stream-K partials: [2 * sk_blocks][256 * 256] f32 (null when sk_blocks == 0)
    int64_t ldx, ldg, lddw;     // elements
    int32_t S, K, N;
    int32_t tiles_m, tiles_n;   // K / 256, N / 256
    int32_t nst;                // stages per tile = S / 32
    int32_t direct_tiles;       // tiles [0, direct_tiles): one workgroup each (blocks [0, direct_tiles))
    int32_t sk_blocks;          // blocks [direct_tiles, direct_tiles + sk_blocks): stream-K over the remaining tiles
    int32_t sk_q;               // stages per stream-K block
};

// tile number -> tile coordinates: bands of 8 tile columns (the last one narrower), row-major inside a band
LWM_DEVICE void wg_tile_coords(const WgradParams& p, int t, int& tm, int& tn) {
    const int band_tiles = p.tiles_m * 8;
    const int band = t / band_tiles, r = t - band * band_tiles;
    const int rest = p.tiles_n - 8 * band;
    const int w = rest < 8 ? rest : 8;
    tm = r / w;
    tn = 8 * band + r % w;
}

// where thread `tid` keeps element (a, b, r) of its accumulators in an f32 partial: float4 granules, consecutive threads
// consecutive granules (1 KiB per wave instruction)
template <int NW>
LWM_DEVICE int wg_partial_index(int tid, int a, int b, int r4) { return ((((a * 4 + b) * 4 + r4) * (64 * NW)) + tid) * 4; }

// the (a, b, r) element's place in the tile: row 64 wm + 32 a + cd_row(r, hi), column 128 wn + 32 b + l31.  A lane holds ONE
// column of each 32 x 32 block: stored from registers that is 2 bytes per lane and row (288 MB written for a 100 MB wqkv
// gradient, PMC WRITE_SIZE), so the wave's 64 x 128 part goes through its own 16 KiB of LDS (row-major, free once every
// wave has left the loop) and out as 16 bytes per lane: 4 rows x 256 B per instruction.
template <int NW>
LWM_DEVICE void wg_store_tile(const WgradParams& p, lds_t lds, int tid, int tm, int tn, const f32x16 (&acc)[16 / NW][4]) {
    constexpr int AM = 16 / NW, RM = 32 * AM;      // a wave's part: RM rows x 128 columns
    // (opaque: the addresses below are not loop invariants hipcc may compute before the main loop and spill)
    const int wave = tid >> 6, lane = (int)opaque((uint32_t)(tid & 63)), l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const lds_t mine = lds + (uint32_t)wave * (RM * 256);
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            sched_fence();        // (one tuple at a time: hipcc otherwise copies every accumulator out first and spills)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                lds_write_bf16(mine + (uint32_t)((32 * a + cd_row(r, hi)) * 256 + (32 * b + l31) * 2), (bf16_t)acc[a][b][r]);
        }
    wave_lds_fence();
    bf16_t* out = p.dw + ((int64_t)tm * kWgBM + RM * wm) * p.lddw + (int64_t)tn * kWgBN + 128 * wn;
    const bool wide = ((p.lddw & 7) == 0) && (((uintptr_t)p.dw & 15) == 0);
    const int ln = lane;
#pragma unroll
    for (int q = 0; q < RM / 4; ++q) {
        const int row = 4 * q + (ln >> 4), c8 = (ln & 15) * 8;
        const u32x4 v = lds_read_u32x4(mine + (uint32_t)(row * 256 + c8 * 2));
        bf16_t* dst = out + (int64_t)row * p.lddw + c8;
        if (wide) {
            global_store_b128(dst, v);
        } else {
            union { u32x4 v; bf16_t h[8]; } u;
            u.v = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = u.h[e];
        }
    }
}

// stages [s0, s1) of tile (tm, tn); `partial` null: the bf16 tile is written, else the f32 partial
template <int NW>
LWM_DEVICE void wg_segment(const WgradParams& p, lds_t lds, int tid, int tm, int tn, int s0, int s1, float* partial) {
    constexpr int AM = 16 / NW, RM = 32 * AM, NF = AM + 4, NP = 32 / NW;      // x blocks, rows, fragments per step, pieces per stage
    const int wave = wave_uniform(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;      // rows RM wm.., columns 128 wn..
    const int64_t m0 = (int64_t)tm * kWgBM, n0 = (int64_t)tn * kWgBN;

    // ---- LDS-DMA: 32 pieces of 1 KiB (4 rows x 256 B) per stage, 8 per sub-tile; wave w moves pieces w (, w + NW) of each of the
    // four sub-tiles; lane l writes physical slot l & 15 of row 4 w + (l >> 4) (+ 16: the same swizzle) and therefore fetches
    // logical slot (l & 15) ^ swz(row)
    uint32_t voff[4];
    {
        const int row = 4 * wave + (lane >> 4);
        const int c16 = ((lane & 15) ^ swz(row)) << 3;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int64_t ld = sub < 2 ? p.ldx : p.ldg;
            voff[sub] = (uint32_t)(((int64_t)row * ld + ((sub & 1) << 7) + c16) * 2);
        }
    }
    const int64_t xstep = (int64_t)kWgBK * p.ldx * 2, gstep = (int64_t)kWgBK * p.ldg * 2;
    const char* xsrc = (const char*)(p.x + m0) + (int64_t)s0 * xstep;
    const char* gsrc = (const char*)(p.g + n0) + (int64_t)s0 * gstep;
    // piece q of this wave (sub-tile q & 3: x lo, x hi, g lo, g hi; rows + 16 (q >> 2)) of local stage i -> slot `slot` & 3
    auto piece = [&](int q, int i, int slot) {
        const int sub = q & 3, half = q >> 2;
        const lds_t dst = lds + (uint32_t)(slot & 3) * kWgSlotBytes + (uint32_t)wave * 1024 + (uint32_t)sub * kWgSubBytes + (uint32_t)half * 4096;
        const char* src = sub < 2 ? xsrc + (int64_t)i * xstep + (int64_t)half * (xstep >> 1) : gsrc + (int64_t)i * gstep + (int64_t)half * (gstep >> 1);
        f4_dma1(voff[sub], src, dst);
    };
    auto piece_set = [&](int i, int slot) {
#pragma unroll
        for (int q = 0; q < NP; ++q) piece(q, i, slot);
    };
    auto issue = [&](int i) { piece_set(i, i); };
    // ---- fragment addresses (relative to a slot): x sub-tile (RM wm) >> 7, column blocks ((RM wm) & 127) / 32 + a; g sub-tile
    // wn, blocks 0..3 (frag_tr_addr's arithmetic written out: no register array is indexed by a run-time value)
    uint32_t xlo[AM], xup[AM], glo[4], gup[4];
    {
        const int gq = lane >> 4, i15 = lane & 15, h2 = gq >> 1;
        const int row = 4 * h2 + (i15 >> 2);
        const lds_t xb = lds + (uint32_t)((RM * wm) >> 7) * kWgSubBytes, gb = lds + (uint32_t)(2 + wn) * kWgSubBytes;
#pragma unroll
        for (int a = 0; a < AM; ++a) {
            const int dcol = 32 * ((((RM * wm) & 127) >> 5) + a) + 16 * (gq & 1) + 4 * (i15 & 3);
            xlo[a] = xb + tile_off(row, dcol >> 3) + (dcol & 7) * 2;
            xup[a] = xb + tile_off(row + 8, dcol >> 3) + (dcol & 7) * 2;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int dcol = 32 * b + 16 * (gq & 1) + 4 * (i15 & 3);
            glo[b] = gb + tile_off(row, dcol >> 3) + (dcol & 7) * 2;
            gup[b] = gb + tile_off(row + 8, dcol >> 3) + (dcol & 7) * 2;
        }
    }
    auto frag = [&](uint32_t lo_a, uint32_t up_a, uint32_t off) {
        const bf16x4 lo = lds_read_tr16(lo_a + off), up = lds_read_tr16(up_a + off);
        bf16x8 o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
        o[4] = up[0]; o[5] = up[1]; o[6] = up[2]; o[7] = up[3];
        return o;
    };
    // fragment f of a 16-row step: 0..AM-1 = x blocks, AM..AM+3 = g blocks
    auto req = [&](int f, uint32_t off) { return f < AM ? frag(xlo[f], xup[f], off) : frag(glo[f - AM], gup[f - AM], off); };
    // the k-th request of a step, in the order the MFMAs need them: x0, g0, x1 .. x(AM-1), g1, g2, g3
    auto nth = [](int k) { return k == 0 ? 0 : k == 1 ? AM : k <= AM ? k - 1 : k; };

    f32x16 acc[AM][4];
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = zero_f32x16();
#pragma unroll
    for (int a = 0; a < AM; ++a) d4_settle_acc4(acc[a]);       // (compiler-written zeros: two wait states before an asm MFMA reads them)

    // Every stage requests its pieces whether or not a stage i + 3 exists (past the end: the last stage once more, into a
    // slot nobody reads again): no branch stands in the MFMA stream and the counted wait is the same in every stage.
    const int n = s1 - s0, last = n - 1;
    block_sync_lds();             // (a second segment: every wave has left the slots of the first)
    issue(0);
    piece_set(1 < last ? 1 : last, 1);
    piece_set(2 < last ? 2 : last, 2);
    wait_vmem_le<2 * NP>();
    block_sync_lds();

    // MFMA m of a step (0..4 AM - 1) = (g block m / AM, x block m % AM); fr[set][f]: the step's fragments, requested one per
    // gap behind the first NF MFMAs of the step before (program order is kept: the MFMAs are asm statements, sched_fence()
    // pins what stands between them)
    bf16x8 fr[2][NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) fr[0][nth(k)] = req(nth(k), 0);
    for (int i = 0; i < n; ++i) {
        const uint32_t cur = (uint32_t)(i & 3) * kWgSlotBytes;
        const uint32_t nxt = (uint32_t)((i + 1) & 3) * kWgSlotBytes;
#pragma unroll
        for (int m = 0; m < 4 * AM; ++m) {
            sched_fence();
            f4_mfma_o(acc[m % AM][m / AM], fr[0][m % AM], fr[0][AM + m / AM]);
            if (m < NF) fr[1][nth(m)] = req(nth(m), cur + (uint32_t)(16 * kRowBytes));
            sched_fence();
        }
        // (the last stage keeps the shape of the others: its barrier is one too many and the fragments it requests from the
        // next slot are never used -- a branch around them would put the accumulators through a control-flow join, and hipcc
        // then moves them between register files inside the loop)
        wait_vmem_le<NP>();
        block_sync_lds();
        const int ahead = i + 3 < last ? i + 3 : last;

#pragma unroll
        for (int m = 0; m < 4 * AM; ++m) {
            sched_fence();
            f4_mfma_o(acc[m % AM][m / AM], fr[1][m % AM], fr[1][AM + m / AM]);
            if (m < NF) fr[0][nth(m)] = req(nth(m), nxt);
            if (m & 1) piece(m >> 1, ahead, i + 3);
            sched_fence();
        }
    }
#pragma unroll
    for (int a = 0; a < AM; ++a) d4_settle_acc4(acc[a]);

    wait_vmem_le<0>();            // (the requests past the end)
    if (!partial) {
        block_sync_lds();         // every wave has left the slots, every piece has landed
        wg_store_tile<NW>(p, lds, tid, tm, tn, acc);
    } else {
        const int otid = (int)opaque((uint32_t)tid);
#pragma unroll
        for (int a = 0; a < AM; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v;
                    v[0] = acc[a][b][4 * r4]; v[1] = acc[a][b][4 * r4 + 1]; v[2] = acc[a][b][4 * r4 + 2]; v[3] = acc[a][b][4 * r4 + 3];
                    global_store_f32x4(partial + wg_partial_index<NW>(otid, a, b, r4), v);
                }
    }
}

template <int NW>
LWM_KERNEL(64 * NW) void wgrad_bf16_kernel(WgradParams p) {
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int lin = block_idx_x();
    // this block's stages [a, b) of the tiles [base, ...) laid end to end -- a whole tile for the first direct_tiles blocks,
    // stages [w q, (w + 1) q) of the remaining tiles for stream-K block w (q <= nst: at most two tiles)
    int base, w = 0;
    int64_t a, b;
    if (lin < p.direct_tiles) {
        // XCD x runs blocks x, x + 8, ...: 32 consecutive ones of an XCD take 32 consecutive tiles (a 4 x 8 block)
        base = lin;
        if ((p.direct_tiles & 255) == 0) {
            const int xcd = lin & 7, i = lin >> 3;
            base = (((i >> 5) * 8 + xcd) << 5) + (i & 31);
        }
        a = 0;
        b = p.nst;
    } else {
        w = lin - p.direct_tiles;
        base = p.direct_tiles;
        const int64_t total = (int64_t)(p.tiles_m * p.tiles_n - p.direct_tiles) * p.nst;
        a = (int64_t)w * p.sk_q;
        b = a + p.sk_q;
        if (b > total) b = total;
    }
    for (int seg = 0; seg < 2 && a < b; ++seg) {
        const int j = (int)(a / p.nst), s0 = (int)(a - (int64_t)j * p.nst);
        const int64_t tile_end = (int64_t)(j + 1) * p.nst;
        const int64_t e = b < tile_end ? b : tile_end;
        const int s1 = (int)(e - (int64_t)j * p.nst);
        int tm, tn;
        wg_tile_coords(p, base + j, tm, tn);
        const bool whole = s0 == 0 && s1 == p.nst;
        wg_segment<NW>(p, lds, tid, tm, tn, s0, s1, whole ? nullptr : p.ws + (int64_t)(2 * w + seg) * kWgTileFloats);
        a = e;
    }
}

// one workgroup per stream-K tile: the partials of the blocks that cut it, added in block order
template <int NW>
LWM_KERNEL(64 * NW) void wgrad_fixup_kernel(WgradParams p) {
    constexpr int AM = 16 / NW;
    const int tid = thread_idx();
    const int j = block_idx_x();
    const int64_t t0 = (int64_t)j * p.nst, t1 = t0 + p.nst;
    const int w_first = (int)(t0 / p.sk_q), w_last = (int)((t1 - 1) / p.sk_q);
    if (w_first == w_last) return;      // no block boundary inside the tile: one block wrote it whole
    f32x16 acc[AM][4];
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = zero_f32x16();
    for (int w = w_first; w <= w_last; ++w) {
        const int seg = ((int64_t)w * p.sk_q) / p.nst == j ? 0 : 1;      // the block's first tile, or its second
        const float* part = p.ws + (int64_t)(2 * w + seg) * kWgTileFloats;
#pragma unroll
        for (int a = 0; a < AM; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 v = global_load_f32x4(part + wg_partial_index<NW>(tid, a, b, r4));
                    acc[a][b][4 * r4] += v[0]; acc[a][b][4 * r4 + 1] += v[1]; acc[a][b][4 * r4 + 2] += v[2]; acc[a][b][4 * r4 + 3] += v[3];
                }
    }
    int tm, tn;
    wg_tile_coords(p, p.direct_tiles + j, tm, tn);
    wg_store_tile<NW>(p, dyn_lds(), tid, tm, tn, acc);
}

}  // namespace lwm
