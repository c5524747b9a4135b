// vqgan_conv.h -- NHWC f32 convolution as an implicit GEMM on the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32) for the VQGAN encoder/decoder.  Requires wave_ops.h.
//
// Replaces flax nn.Conv as used by lwm/vqgan.py: 3x3 SAME (:155,:163,:172-175,
// :183,:253,:257), 1x1 (:114-115,:262), Downsample = zero pad bottom/right +
// 3x3 stride-2 VALID (:291-300), Upsample = nearest x2 + 3x3 SAME (:312-318),
// the ResnetBlock residual add (:263) and VQGANModel.decode's clip (:141) as
// epilogue options.  Kernel tensor stays in flax's HWIO layout.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = taps x Cin.  Arithmetic
// contract (identical to oracle/vqgan_ref.c, so results are bit-exact):
//   out = ((P_0 + P_1) + ... + P_{T-1}) + bias [+ residual]
//   P_t = fmaf chain over c_in = 0..Cin-1 starting from 0, taps in (kh,kw) order.
// The f32 MFMA IS that chain: each instruction does two fused multiply-adds per
// output element in k order with one rounding each, so walking c_in in order on
// one accumulator per tap reproduces it exactly.  Out-of-image taps contribute
// exact zeros.
//
// Workgroup = 256 threads = WM x WN waves; wave tile = (32*MB) pixels x (32*NB)
// channels = MB*NB accumulators; K is walked in chunks of 32 input channels of
// one tap, double buffered through LDS:
//   A tile [BM pixels][32 cin] f32, rows of 128 + 16 bytes (see "The A operands of one k-quad" below); the staging
//     ds_write_b128 (8 lanes = one row) is a contiguous 128 B.
//   B tile [32 cin][BN cout] f32, linear; fragments are ds_read_b32 rows
//     (32 consecutive floats per half-wave) -> conflict-free.
// A fragment: lane (i, hi) reads elements hi and 2 + hi of 4 consecutive cin of pixel i and feeds k-pair (4u+2t, 4u+2t+1)
// with element 2t+hi, so c_in is consumed in natural order.  This kernel is MFMA-bound (64 cycles per instruction per SIMD,
// 157 TF chip peak); LDS and L2 traffic are far below their limits by construction.
#pragma once

namespace lwm {

#ifndef LWM_EMU      // (the host emulation of the CPU tests brings its own)
// one float to base + voff + soff bytes (the store twin of global_load_f32_at)
LWM_DEVICE void global_store_f32_at(float* base, uint32_t voff, uint32_t soff, float v) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, (int)voff, (int)soff, 0);
}

// this wave's issue priority (0 .. 3), as it stands until the next call: s_setprio ignores EXEC, so a caller that wants it for
// SOME waves branches on a wave-uniform condition
LWM_DEVICE void wave_priority(int p) {
    if (p) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// A range of global memory behind a buffer descriptor: 16 bytes at base + voff, or ZEROS when voff + 16 > bytes (the
// descriptor's range check: an out-of-image tap is an offset past the range, and its zero fill costs no instruction).  The
// descriptor is made ONCE (its base goes through v_readfirstlane: per load that is two vector instructions among the MFMAs).
typedef __amdgpu_buffer_rsrc_t ranged_t;
LWM_DEVICE ranged_t ranged_make(const float* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    // (EVERY word of the descriptor provably uniform: with the byte count left in a vector register hipcc keeps the whole
    //  descriptor there and reads it back lane 0 by lane 0 in front of every load)
    return __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
LWM_DEVICE f32x4 ranged_load_f32x4(ranged_t r, uint32_t voff) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
#endif

struct ConvParams {
    const float* x;
    const float* w;
    const float* bias;  // [Cout] or null
    const float* res;   // [M, Cout] or null
    float* y;           // [M, Cout]
    int32_t B, Hin, Win, Cin, Cout, KH, KW, stride, pad, up_shift, Ho, Wo, clip;
    int64_t M;  // B*Ho*Wo
};

constexpr int kConvKC = 32;  // input channels per K chunk

template <int WM, int WN, int MB, int NB>
struct ConvCfg {
    static constexpr int NT = 64 * WM * WN;
    static constexpr int BM = 32 * MB * WM;
    static constexpr int BN = 32 * NB * WN;
    static constexpr int A_ROW = kConvKC * 4 + 16;   // LDS stride of a staged pixel: its 32 channels + one unused 16-byte slot
    static constexpr int A_BYTES = BM * A_ROW;
    static constexpr int B_BYTES = kConvKC * BN * 4;
    static constexpr int BUF_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static constexpr int APX = NT / 8;         // pixels staged per pass (8 lanes x 16 B per pixel)
    static constexpr int AP = BM / APX;        // A staging passes
    static constexpr int BROWS = NT * 4 / BN;  // B rows staged per pass
    static constexpr int BP = kConvKC / BROWS; // B staging passes
    static_assert(BM % APX == 0 && AP >= 1, "A staging shape");
    static_assert(kConvKC % BROWS == 0 && BP >= 1 && BN / 4 <= NT, "B staging shape");
};

template <int N>
struct IntTag { static constexpr int value = N; };

// The A operands of one k-quad.  MFMA t of the quad (k pair 2t, 2t + 1) takes, from lane half hi, element 2t + hi of the
// lane's channel quad: the two dwords hi and hi + 2 of the quad's 16 bytes -- ONE ds_read2_b32 (offset1 = offset0 + 2), no
// vector instruction.  Rounds 3-6 read the whole quad (ds_read_b128) and picked the two with v_bfi_b32: those two VALU
// instructions per quad, whose results an MFMA reads, held every convolution loop at 0.89-0.94 of the matrix rate; a k-quad
// with the same requests and NO vector instruction runs at 0.99, one wave per SIMD or two (scripts/micro/mfma_aux_rate.cpp,
// profiles/r06_conv_persistent.md).  The address is a lane base that changes per tap (patch kernels) or per chunk (staged
// kernels) plus an immediate: pixel rows carry one unused 16-byte slot (stride = channels * 4 + 16), which spreads pixel p,
// quad q over banks 4 ((p + q) mod 8) + hi.  A dword read reaches only the 8 banks of its position in the quad: 4-way
// conflicts, 16 LDS cycles per instruction, a quarter of the LDS's time at 8 waves per CU -- measured free.

LWM_DEVICE f32x4 zero_f32x4() {
    f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    return z;
}

// VEC: Cin % 4 == 0 and Cout % 4 == 0 (every layer but conv_in / the RGB output
// conv): all staging loads are 16-byte and NOTHING in the staging path branches
// at run time -- a load behind a runtime branch, even a uniform one, makes hipcc
// wait vmcnt(0) before the next load (measured: 4.3k cycles per chunk).
//
// BDIRECT (Cin % 32 == 0, Cout % BN == 0): the B operand never passes through LDS -- a lane's B value of one MFMA is
// ONE float of its column, fetched from L1/L2 into registers a chunk ahead (buffer loads with the row offset in an
// SGPR, as the patch kernels below do); only the A tile is staged.  Half the staging loads and ds_writes, half the
// fragment reads; same arithmetic.
template <int WM, int WN, int MB, int NB, bool VEC, bool BDIRECT = false>
LWM_DEVICE void conv_igemm_body(const ConvParams& p) {
    using Cfg = ConvCfg<WM, WN, MB, NB>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, AP = Cfg::AP, BP = BDIRECT ? 0 : Cfg::BP;
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (p.Cout + BN - 1) / BN;
    const int64_t bm = block_idx_x() / ntn;
    const int bn = block_idx_x() % ntn;
    const int64_t m0 = bm * BM;
    const int n0 = bn * BN;

    // ---- staging roles (fixed for the whole launch)
    const int a_slot = tid & 7;
    int a_oy[AP], a_ox[AP];
    int64_t a_base[AP];  // element offset of (b, 0, 0, 0) in x; -1 = pixel past M
    for (int ps = 0; ps < AP; ++ps) {
        const int px = ps * Cfg::APX + (tid >> 3);
        const int64_t m = m0 + px;
        if (m < p.M) {
            const int ox = (int)(m % p.Wo);
            const int64_t t = m / p.Wo;
            a_ox[ps] = ox * p.stride - p.pad;
            a_oy[ps] = (int)(t % p.Ho) * p.stride - p.pad;
            a_base[ps] = (t / p.Ho) * (int64_t)p.Hin * p.Win * p.Cin;
        } else {
            a_ox[ps] = 0;
            a_oy[ps] = 0;
            a_base[ps] = -1;
        }
    }
    const lds_t a_w = lds + (uint32_t)(tid >> 3) * Cfg::A_ROW + (uint32_t)(a_slot << 4);  // + pass * APX * A_ROW
    const int b_row = tid / (BN / 4);
    const int b_col = (tid % (BN / 4)) * 4;
    const lds_t b_w = lds + Cfg::A_BYTES + (uint32_t)(b_row * BN + b_col) * 4;
    const int Hv = p.Hin << p.up_shift, Wv = p.Win << p.up_shift;
    constexpr bool cin_vec = VEC, cout_vec = VEC;

    const int nch = (p.Cin + kConvKC - 1) / kConvKC;
    const int ntap = p.KH * p.KW;
    const int nit = ntap * nch;

    f32x4 sa[AP], sb[BP > 0 ? BP : 1];
    uint32_t s_ok = 0;  // bit ps: sa[ps] valid; bit 8+ps: sb[ps] valid (else the tile gets zeros)

    // BDIRECT: a staging load is a RANGED buffer load (ranged_load_f32x4) whose offset lies past the tensor for an
    // out-of-image tap -- the descriptor's range check returns the zeros, and a staging pass is one v_add, the load and the
    // ds_write (rounds 1-6: a clamped 64-bit address, a validity bit and four v_cndmask per pass, ~50 vector instructions per
    // 64 MFMAs: every B-direct layer +4-6 % without them, profiles/r06_conv_ranged_ab.txt).  The other forms:
    // loads are UNCONDITIONAL (out-of-image taps / channels read a clamped, valid
    // address) and the zero fill is a select at write time: a load behind a runtime
    // branch makes hipcc wait vmcnt(0) at every join, which serialised the eight
    // staging loads of a chunk (4.3k cycles per chunk, measured).
    // One staging load (index l: 0..AP-1 = A passes, AP..AP+BP-1 = B passes) of chunk `it`.
    // Measured alternatives: issuing them one per k-quad inside the MFMA loop
    // instead of as a burst, or delaying co-resident workgroups against each other, do not
    // help; without any staging the loop runs 125 TF/s, with it 100 TF/s.
    // The chunks are staged in order (0, 1, 2, ...): (tap, chunk) is a running counter, and what depends on the tap only
    // -- where the tap's pixel of each staged row lies in its image, or that it lies outside -- is computed when the tap
    // changes, not per chunk (a_tap: element offset inside the image, -1 = zero fill).
    int ld_kh = 0, ld_kw = 0, ld_ch = 0, ld_tap = 0;
    int a_tap[AP];
    // BDIRECT: the tap's pixel as a BYTE offset, or kOobTap for a tap
    // outside the image / a pixel past M: + the chunk's channel offset it stays past the tensor's end, the ranged load
    // returns zeros, and neither the load nor the LDS write selects anything
    // (offsets are relative to the image of the tile's first pixel -- a tile reaches a few images at most, and the launch
    //  guarantees that span < 0xE0000000 bytes -- so the tensor itself may be of any size)
    constexpr uint32_t kOobTap = 0xF0000000u;
    const int64_t img_el = (int64_t)p.Hin * p.Win * p.Cin;
    const int64_t el0 = m0 / ((int64_t)p.Ho * p.Wo) * img_el;      // first element of that image (workgroup-uniform)
    const float* const x0 = p.x + el0;
    const int64_t x_left = ((int64_t)p.B * img_el - el0) * 4;
    const ranged_t x_range = ranged_make(x0, (uint32_t)(x_left < 0xE0000000LL ? x_left : 0xE0000000LL));
    uint32_t a_tapb[AP];
    auto stage_tap = [&]() {
#pragma unroll
        for (int ps = 0; ps < AP; ++ps) {
            const int vy = a_oy[ps] + ld_kh, vx = a_ox[ps] + ld_kw;
            const bool ok = a_base[ps] >= 0 && vy >= 0 && vy < Hv && vx >= 0 && vx < Wv;
            a_tap[ps] = ok ? ((vy >> p.up_shift) * p.Win + (vx >> p.up_shift)) * p.Cin : -1;
            if constexpr (BDIRECT) a_tapb[ps] = ok ? (uint32_t)(a_base[ps] - el0 + a_tap[ps]) * 4u + (uint32_t)a_slot * 16u : kOobTap;
        }
    };
    stage_tap();
    auto stage_begin = [&](int) { s_ok = 0; };
    auto stage_end = [&]() {           // advance to the next (tap, chunk)
        if (++ld_ch == nch) {
            ld_ch = 0;
            ++ld_tap;
            if (++ld_kw == p.KW) {
                ld_kw = 0;
                ++ld_kh;
            }
            stage_tap();
        }
    };
    auto stage_load_one = [&](int l) {
        if (BDIRECT && l < AP) {
            sa[l] = ranged_load_f32x4(x_range, a_tapb[l] + (uint32_t)ld_ch * (kConvKC * 4u));
        } else if (l < AP) {
            const int ps = l;
            const int c0 = ld_ch * kConvKC + a_slot * 4;
            const bool ok = a_tap[ps] >= 0 && c0 < p.Cin;
            const int64_t off = ok ? a_base[ps] + (int64_t)(a_tap[ps] + c0) : (int64_t)0;
            const float* src = p.x + off;
            if constexpr (cin_vec) {
                sa[ps] = global_load_f32x4(src);
            } else {  // Cin % 4 != 0 (conv_in): element loads, clamped inside the row
                f32x4 v;
                for (int j = 0; j < 4; ++j) v[j] = src[(ok && c0 + j < p.Cin) ? j : 0];
                for (int j = 0; j < 4; ++j) v[j] = (c0 + j < p.Cin) ? v[j] : 0.0f;
                sa[ps] = v;
            }
            s_ok |= ok ? (1u << ps) : 0u;
        } else {
            const int ps = l - AP;
            const float* wt = p.w + (int64_t)ld_tap * p.Cin * p.Cout;
            const int k = ld_ch * kConvKC + ps * Cfg::BROWS + b_row;
            const int col = n0 + b_col;
            const bool ok = k < p.Cin && col < p.Cout;
            const float* src = wt + (ok ? (int64_t)k * p.Cout + col : (int64_t)0);
            if constexpr (cout_vec) {
                sb[ps] = global_load_f32x4(src);
            } else {
                f32x4 v;
                for (int j = 0; j < 4; ++j) v[j] = src[(ok && col + j < p.Cout) ? j : 0];
                for (int j = 0; j < 4; ++j) v[j] = (col + j < p.Cout) ? v[j] : 0.0f;
                sb[ps] = v;
            }
            s_ok |= ok ? (1u << (8 + ps)) : 0u;
        }
    };
    auto stage_load = [&](int it) {      // (calls come in chunk order)
        stage_begin(it);
#pragma unroll
        for (int l = 0; l < AP + BP; ++l) stage_load_one(l);
        stage_end();
    };
    auto stage_write = [&](uint32_t bo) {       // bo = byte offset of the buffer
        for (int ps = 0; ps < AP; ++ps) {
            const int px = ps * Cfg::APX + (tid >> 3);
            lds_write_f32x4(a_w + bo + (uint32_t)ps * Cfg::APX * Cfg::A_ROW, BDIRECT || ((s_ok >> ps) & 1) ? sa[ps] : zero_f32x4());
        }
        for (int ps = 0; ps < BP; ++ps)
            lds_write_f32x4(b_w + bo + (uint32_t)ps * Cfg::BROWS * BN * 4,
                            (s_ok >> (8 + ps)) & 1 ? sb[ps] : zero_f32x4());
    };

    f32x16 acc[MB][NB], acc_tap[MB][NB];
    for (int i = 0; i < MB; ++i)
        for (int j = 0; j < NB; ++j) {
            acc[i][j] = zero_f32x16();
            acc_tap[i][j] = zero_f32x16();
        }

    // fragment addresses (buffer 0)
    // (channel quad u of staged pixel px at px * A_ROW + 16 u: inside a chunk a fragment address is the chunk's lane base +
    //  an immediate; the 16-byte pad per row spreads pixel px, quad u over banks 4 ((px + u) mod 8) + hi, as the XOR swizzle
    //  of rounds 1-6 did -- the 4-way pattern a dword read of one quad position cannot avoid)
    uint32_t a_r[MB];
    for (int i = 0; i < MB; ++i) {
        const int px = (wm * MB + i) * 32 + l31;
        a_r[i] = lds + (uint32_t)px * Cfg::A_ROW + (uint32_t)hi * 4u;
    }
    const lds_t b_r = lds + Cfg::A_BYTES + (uint32_t)(hi * BN + wn * NB * 32 + l31) * 4;

    // BDIRECT: row (it*32 + 4u + 2t + hi) of the [taps*Cin][Cout] kernel matrix, column n0 + wn*NB*32 + j*32 + l31
    const uint32_t b_voff = (uint32_t)(hi * p.Cout + n0 + wn * NB * 32 + l31) * 4u;
    const uint32_t b_rowb = (uint32_t)p.Cout * 4u;
    // two register sets of a quarter chunk each (k-quad pairs): pair q+1 is requested when pair q begins, 16 MFMAs
    // (>= 1024 cycles) ahead of its first use
    float bq[2][2][2][NB];
    auto load_b = [&](int it, int pair) {          // pair 0..3 of chunk `it` -> set pair & 1 (all compile-time after unrolling)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    bq[pair & 1][u][t][j] = global_load_f32_at(p.w, b_voff + (uint32_t)j * 128u,
                                                               (uint32_t)(it * 32 + 8 * pair + 4 * u + 2 * t) * b_rowb);
    };
    if constexpr (BDIRECT) load_b(0, 0);      // (load_b: unconditional -- past the end the last chunk is read again)

    if constexpr (BDIRECT) {
        // A ring of three buffers: chunk it+1 is written when iteration `it` BEGINS (it was loaded an iteration ago),
        // the one barrier of the iteration sits in its middle -- behind it chunk it+1 is visible and chunk it-1 is no
        // longer read by anyone -- and the last k-quad requests the first fragment of chunk it+1: no LDS latency is
        // exposed at a chunk boundary, no ds_write waits for a load.
        constexpr uint32_t AB = Cfg::A_BYTES;
        float afr[2][2][MB];      // [set][t][i]: the A operands straight from LDS
        lds_t a_cur[MB], a_nxt[MB];       // the lane's fragment bases in the current / next ring buffer
        auto set_bases = [&](lds_t (&dst)[MB], uint32_t bo) {
            for (int i = 0; i < MB; ++i) dst[i] = opaque(a_r[i] + bo);
        };
        auto load_a = [&](const lds_t (&base)[MB], int u, int set) {
            for (int i = 0; i < MB; ++i) {
                afr[set][0][i] = lds_read_f32(base[i] + (uint32_t)u * 16u);
                afr[set][1][i] = lds_read_f32(base[i] + (uint32_t)u * 16u + 8u);
            }
        };
        stage_load(0);
        stage_write(0);
        if (nit > 1) stage_load(1);
        block_sync();
        uint32_t cur = 0, nxt = AB, nn = 2 * AB;
        set_bases(a_cur, cur);
        load_a(a_cur, 0, 0);
        int in_tap = 0;
        // Issue order of a k-quad, as in the patch kernels: every other instruction sits behind one of the quad's first four
        // MFMAs -- the next quad's fragment reads behind the first, the two halves of a B request behind the second and third,
        // ONE staging pass behind the fourth (quads 0-3: a ds_write of chunk it + 1, in front of the chunk's barrier; quads
        // 4-7: a global load of chunk it + 2) -- instead of the whole staging burst in front of the chunk's first MFMA and each
        // quad's requests in front of its eight: +2-4 % on every layer of this kernel (profiles/r06_conv_persistent.md).
        static_assert(AP == 4 && MB == 2 && NB == 2, "the chunk below is written out for eight MFMAs per k-quad and four staging passes");
        auto stage_write_one = [&](uint32_t bo, int ps) {
            const int px = ps * Cfg::APX + (tid >> 3);
            lds_write_f32x4(a_w + bo + (uint32_t)ps * Cfg::APX * Cfg::A_ROW, sa[ps]);
        };
        auto load_b_half = [&](int itb, int pair, int u2) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    bq[pair & 1][u2][t][j] = global_load_f32_at(p.w, b_voff + (uint32_t)j * 128u, (uint32_t)(itb * 32 + 8 * pair + 4 * u2 + 2 * t) * b_rowb);
        };
        for (int it = 0; it < nit; ++it) {
            const bool more = it + 1 < nit;
            const bool more2 = it + 2 < nit;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int set = u & 1;
                const int itb = u == 6 ? (more ? it + 1 : it) : it, pair = ((u >> 1) + 1) & 3;
                int k = 0;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j, ++k) {
                            acc_tap[i][j] = mfma_32x32x2_f32(afr[set][t][i], bq[(u >> 1) & 1][u & 1][t][j], acc_tap[i][j]);
                            sched_fence();
                            if (k == 0) {
                                if (u + 1 < 8) load_a(a_cur, u + 1, set ^ 1);
                                else {
                                    set_bases(a_nxt, more ? nxt : cur);
                                    load_a(a_nxt, 0, set ^ 1);
                                }
                            }
                            if (k == 1 && (u & 1) == 0) load_b_half(itb, pair, 0);
                            if (k == 2 && (u & 1) == 0) load_b_half(itb, pair, 1);
                            if (k == 3) {
                                if (u < 4) { if (more) stage_write_one(nxt, u); }
                                else if (more2) {
                                    if (u == 4) stage_begin(it + 2);
                                    stage_load_one(u - 4);
                                    if (u == 7) stage_end();
                                }
                            }
                            if (k < 4) sched_fence();
                        }
                if (u == 3) block_sync_lds();
            }
            if (++in_tap == nch) {  // tap finished: s = s + P_t
                in_tap = 0;
                for (int i = 0; i < MB; ++i)
                    for (int j = 0; j < NB; ++j) {
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + acc_tap[i][j][r];
                        acc_tap[i][j] = zero_f32x16();
                    }
            }
            const uint32_t t3 = cur;
            cur = nxt;
            nxt = nn;
            nn = t3;
            for (int i = 0; i < MB; ++i) a_cur[i] = a_nxt[i];
        }
    }
    if constexpr (!BDIRECT) {
    stage_load(0);
    stage_write(0);
    block_sync();
    auto chunk = [&](int it) {
        const int buf = it & 1;
        const bool more = it + 1 < nit;
        if (more) stage_load(it + 1);
        const uint32_t bo = (uint32_t)buf * Cfg::BUF_BYTES;
        // fragments of k-quad u+1 are fetched while the MFMAs of k-quad u run
        float afr[2][2][MB];
        float bf[2][2][NB];
        auto load_frag = [&](int u, int set) {
            for (int i = 0; i < MB; ++i) {
                afr[set][0][i] = lds_read_f32(a_r[i] + bo + (uint32_t)u * 16u);
                afr[set][1][i] = lds_read_f32(a_r[i] + bo + (uint32_t)u * 16u + 8u);
            }
            for (int t = 0; t < 2; ++t)
                for (int j = 0; j < NB; ++j)
                    bf[set][t][j] = lds_read_f32(b_r + bo + (uint32_t)((4 * u + 2 * t) * BN + j * 32) * 4);
        };
        load_frag(0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int set = u & 1;
            if (u + 1 < 8) load_frag(u + 1, set ^ 1);
            sched_fence();  // keep the prefetch ABOVE this k-quad's MFMAs (hipcc sinks it otherwise)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                for (int i = 0; i < MB; ++i)
                    for (int j = 0; j < NB; ++j)
                        acc_tap[i][j] = mfma_32x32x2_f32(afr[set][t][i], bf[set][t][j], acc_tap[i][j]);
            sched_fence();
        }
        if ((it + 1) % nch == 0) {  // tap finished: s = s + P_t
            for (int i = 0; i < MB; ++i)
                for (int j = 0; j < NB; ++j) {
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + acc_tap[i][j][r];
                    acc_tap[i][j] = zero_f32x16();
                }
        }
        if (more) stage_write((uint32_t)(buf ^ 1) * Cfg::BUF_BYTES);
        block_sync();
    };
    for (int it = 0; it < nit; ++it) chunk(it);
    }

    // ---- epilogue: + bias [+ residual] [clip], one float per (pixel, channel).
    const int64_t tile_base = m0 * p.Cout;
    float* const yb = p.y + tile_base;
    const float* const rb = p.res ? p.res + tile_base : p.y;
    const int64_t left = p.M - m0;                    // >= 1 rows of this tile exist
    const int rows = left < BM ? (int)left : BM;
    if (rows == BM && n0 + BN <= p.Cout) {
        // A whole tile (every layer of the network): each step over all MB x NB accumulators behind ONE uniform branch.  All
        // residual reads of the wave are in flight together -- taken accumulator by accumulator, each of the MB x NB blocks
        // waited a whole HBM round trip in turn, the co-resident workgroup in the same phase -- and an address is the lane's
        // part in a VGPR plus the (i, r) part, wave-uniform, in an SGPR: no vector instruction per load or store.
        const int wu = wave_uniform(wave);
        const int wmu = wu / WN, wnu = wu % WN;
        const uint32_t voff = (uint32_t)(4 * hi * p.Cout + n0 + wnu * NB * 32 + l31) * 4u;
        auto soff = [&](int i, int r) -> uint32_t {
            return (uint32_t)((wmu * MB + i) * 32 + (r & 3) + 8 * (r >> 2)) * (uint32_t)p.Cout * 4u;
        };
        float rv[MB][NB][16];
        if (p.res) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[i][j][r] = global_load_f32_at(rb, voff + (uint32_t)j * 128u, soff(i, r));
        }
        if (p.bias) {
            float bv[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) bv[j] = p.bias[n0 + (wn * NB + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + bv[j];
        }
        if (p.res) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + rv[i][j][r];
        }
        if (p.clip) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];
                        acc[i][j][r] = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
                    }
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) global_store_f32_at(yb, voff + (uint32_t)j * 128u, soff(i, r), acc[i][j][r]);
        return;
    }
    // A ragged tile (rows past M, columns past Cout): 32-bit offsets inside the tile; rows past M are clamped for the residual
    // read and skipped for the store; the 16 residual reads of an accumulator are issued together.
    {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int co = n0 + (wn * NB + j) * 32 + l31;
                const bool col_ok = co < p.Cout;
                const int coc = col_ok ? co : p.Cout - 1;
                const float bv = p.bias ? p.bias[coc] : 0.0f;
                uint32_t off[16];
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (wm * MB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    off[r] = (uint32_t)(ml < rows ? ml : rows - 1) * (uint32_t)p.Cout + (uint32_t)coc;
                }
                if (p.res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = rb[off[r]];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (wm * MB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float v = acc[i][j][r];
                    if (p.bias) v = v + bv;
                    if (p.res) v = v + rv[r];
                    if (p.clip) v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
                    if (col_ok && ml < rows) yb[off[r]] = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// Patch-resident form for the 3x3 stride-1 SAME convolutions with 128 or 256 input channels (with or
// without the nearest x2 upsample folded in) -- 85 % of the VQGAN's FLOPs.  Same arithmetic, same order,
// bit-identical results; what changes is where the operands come from:
//   * A: the workgroup's output tile is TH x 16 pixels of ONE image; its (TH+2) x 18 input halo patch, ALL
//     input channels, is brought into LDS ONCE by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU)
//     and the nine taps read it at shifted addresses.  The generic kernel re-stages the A tile per tap and
//     chunk (9x the traffic, plus the address arithmetic and ds_writes of a register-staged tile).
//     Pixel row = CIN*4 bytes; 16-byte slot L of the patch pixel in patch column x sits at L ^ (x & 15).  A
//     ds_read_b128 is served in four groups of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
//     + 32 (MI355X_MICROARCH.md, LDS): of a wave's two tile rows a group takes columns 0-3 and 12-15 of one and
//     4-11 of the other -- 16 distinct columns, so with the COLUMN as the key every group covers all 16 slots of
//     a 256-byte bank row (rounds 3-6 keyed on the patch pixel's index, 18 per row: two of the 16 collided in
//     every group, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50).  The swizzle is applied on the SOURCE
//     address of the DMA (the LDS side of a DMA is lane-linear).
//     Out-of-image pixels are zero-filled with ds_writes after the DMA has landed.
//   * B: never in LDS.  A lane's B operand of one MFMA is ONE float, w[tap][cin][cout = its column]; the 32
//     lanes of a half-wave read 128 contiguous bytes.  The 16*NB floats of the next (tap, chunk) are
//     fetched into registers (buffer loads, L1/L2 hits: every workgroup walks the same 9*CIN*Cout*4 bytes)
//     while the current ones are consumed.
// After the prologue there is NO barrier: the waves of a workgroup drift apart and fill each other's gaps.
template <int CIN, int TH, int WM, int WN, int NB>
struct PatchCfg {
    static constexpr int NW = WM * WN, NT = 64 * NW, TW = 16;
    static constexpr int BM = TH * TW, BN = 32 * NB * WN;
    static constexpr int MB = BM / (32 * WM);          // 32-pixel blocks per wave
    static constexpr int PW = TW + 2, PH = TH + 2, PPX = PH * PW;
    static constexpr int ROWB = CIN * 4;               // bytes of channels per patch pixel
    static constexpr int SPP = CIN / 4;                // 16-byte channel slots per pixel
    static constexpr int RS = ROWB + 16;               // LDS stride of a patch pixel: its channels + one unused 16-byte slot
    static constexpr int NINS = (PPX * (SPP + 1) + 63) / 64;   // patch DMA wave-instructions (1 KiB of the LDS image each)
    static constexpr int LDS_BYTES = NINS * 1024;
    static constexpr int NCH = CIN / kConvKC;
    static_assert(BM % (32 * WM) == 0 && SPP >= 16 && SPP <= 64 && LDS_BYTES <= 160 * 1024, "patch shape");
    static_assert((NINS + NW - 1) / NW <= 32, "zero-fill mask is 32 bits");
    static_assert(NCH % 2 == 0, "the chunk loop is unrolled by two");
};

// The halo patch of a TH x 16 output tile at (ty0, tx0) of one image (xb), all input channels, into LDS at `lds`
// (every lane fetches SOME valid address; out-of-image slots are overwritten with zeros once the DMA has landed).
// In two halves, so that a persistent workgroup has the next tile's patch in flight while it writes its results:
// conv_patch_request issues the DMA (returns the zero-fill mask), conv_patch_land waits for it, fills the zeros and
// ends with a workgroup barrier.
template <class Cfg>
LWM_DEVICE uint32_t conv_patch_request(const ConvParams& p, const float* xb, int ty0, int tx0, int wave, int lane, lds_t lds) {
    constexpr int NW = Cfg::NW, PW = Cfg::PW, CIN = Cfg::ROWB / 4;
    const int Hv = p.Hin << p.up_shift, Wv = p.Win << p.up_shift;
    uint32_t zmask = 0;
    for (int k = 0; k * NW + wave < Cfg::NINS; ++k) {
        const int g = (k * NW + wave) * 64 + lane;     // 16-byte slot of the LDS image: pixel g / (SPP + 1), channel slot g % (SPP + 1)
        const int ppr = g / (Cfg::SPP + 1), qr = g - ppr * (Cfg::SPP + 1);
        const int pp = ppr < Cfg::PPX ? ppr : Cfg::PPX - 1;                 // (past the image's last pixel: the tail of the last DMA instruction)
        const int lslot = qr < Cfg::SPP ? qr : 0;                           // (the unused slot of a pixel fetches its first one)
        const int py = pp / PW, px = pp - py * PW;
        const int vy = ty0 + py - 1, vx = tx0 + px - 1;
        const bool ok = vy >= 0 && vy < Hv && vx >= 0 && vx < Wv;
        const int sy = ok ? (vy >> p.up_shift) : 0, sx = ok ? (vx >> p.up_shift) : 0;
        glds_load_b128(xb + ((int64_t)sy * p.Win + sx) * CIN + lslot * 4, lds + (uint32_t)(k * NW + wave) * 1024);
        zmask |= ok ? 0u : (1u << k);
    }
    return zmask;
}
template <class Cfg>
LWM_DEVICE void conv_patch_land(uint32_t zmask, int wave, int lane, lds_t lds) {
    constexpr int NW = Cfg::NW;
    glds_wait_all();
    for (int k = 0; k * NW + wave < Cfg::NINS; ++k)
        if ((zmask >> k) & 1) lds_write_f32x4(lds + (uint32_t)((k * NW + wave) * 64 + lane) * 16, zero_f32x4());
    block_sync_lds();
}
template <class Cfg>
LWM_DEVICE void conv_patch_fill(const ConvParams& p, const float* xb, int ty0, int tx0, int wave, int lane, lds_t lds) {
    conv_patch_land<Cfg>(conv_patch_request<Cfg>(p, xb, ty0, tx0, wave, lane, lds), wave, lane, lds);
}

template <int CIN, int TH, int WM, int WN, int NB, bool RES>
LWM_DEVICE void conv_patch_body(const ConvParams& p) {
    using Cfg = PatchCfg<CIN, TH, WM, WN, NB>;
    constexpr int MB = Cfg::MB, BN = Cfg::BN, NCH = Cfg::NCH, PW = Cfg::PW, NW = Cfg::NW;
    constexpr int nit = 9 * NCH;
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = wave_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // Persistent: the grid is one workgroup per CU and a workgroup walks the tiles blockIdx, blockIdx + gridDim, ... --
    // a workgroup of this size (92 - 108 KiB of LDS, 8 waves) has the CU to itself, and between the exit of one and the entry
    // of the next the CU stood empty for 31 k cycles, 9 % of a tile's time (s_memtime stamps, profiles/r06_conv_persistent.md).
    const int ntn = p.Cout / BN;
    const int tiles_x = p.Wo / Cfg::TW, tiles_y = p.Ho / TH;
    const int64_t ntiles = p.M / Cfg::BM * ntn;
    struct TileAt { int n0, tx0, ty0, b; };
    auto tile_at = [&](int64_t tl) {
        TileAt t;
        t.n0 = (int)(tl % ntn) * BN;
        int64_t bm = tl / ntn;
        t.tx0 = (int)(bm % tiles_x) * Cfg::TW;
        bm /= tiles_x;
        t.ty0 = (int)(bm % tiles_y) * TH;
        t.b = (int)(bm / tiles_y);
        return t;
    };
    auto request_patch = [&](const TileAt& t) {
        return conv_patch_request<Cfg>(p, p.x + (int64_t)t.b * p.Hin * p.Win * CIN, t.ty0, t.tx0, wave, lane, lds);
    };
    int64_t tl = block_idx_x();
    if (tl >= ntiles) return;
    uint32_t zmask = request_patch(tile_at(tl));
#pragma unroll 1
    for (;; tl += grid_dim_x()) {
    const TileAt here = tile_at(tl);
    const int n0 = here.n0, tx0 = here.tx0, ty0 = here.ty0, b = here.b;

    // ---- B operands: row (it*32 + 4u + 2t + hi) of the [9*CIN][Cout] kernel matrix, column n0 + wn*NB*32 + j*32 + l31
    const uint32_t b_voff = (uint32_t)(hi * p.Cout + n0 + wn * NB * 32 + l31) * 4u;
    const uint32_t b_rowb = (uint32_t)p.Cout * 4u;
    // A ring of kBRing k-quads, requested kBAhead quads before their use (rows are contiguous across (tap, chunk)s: k-quad g
    // of the whole walk reads rows 4 g + 2 t + hi)
    constexpr int kBRing = 4, kBAhead = 3;
    constexpr int kPrioTap = 4;
    static_assert(8 % kBRing == 0 && kBAhead < kBRing, "B ring");
    float bq[kBRing][2][NB];
    auto load_b_half_at = [&](int it, int uu, int half) {      // k-quad uu (0 .. 8 + kBAhead - 1: it may run into the next (tap, chunk)) of (tap, chunk) `it`, MFMA t = half
#pragma unroll
        for (int j = 0; j < NB; ++j)
            bq[uu % kBRing][half][j] = global_load_f32_at(p.w, b_voff + (uint32_t)j * 128u, (uint32_t)(it * 32 + 4 * uu + 2 * half) * b_rowb);
    };
    auto load_b_half = [&](int g, int half) { load_b_half_at(0, g, half); };
#pragma unroll
    for (int g = 0; g < kBAhead; ++g) {
        load_b_half(g, 0);
        load_b_half(g, 1);
    }

    conv_patch_land<Cfg>(zmask, wave, lane, lds);

    // ---- A fragment addressing
    int pp0[MB];                                        // patch pixel of tap (0, 0) for this lane's pixel
    for (int i = 0; i < MB; ++i) {
        const int px = (wm * MB + i) * 32 + l31;
        pp0[i] = (px / Cfg::TW) * PW + (px % Cfg::TW);
#ifndef LWM_EMU
        asm volatile("" : "+v"(pp0[i]));     // (per tile: hoisted out of the tile loop, the swizzled fragment addresses of all (chunk, k-quad)s spill)
#endif
    }
    float af[2][2][MB];                                 // [quad parity][t][i]
    // (channel quad q of a pixel sits at pixel * RS + 16 q: within a tap every fragment address is the tap's lane base + an
    //  IMMEDIATE -- ds_read2_b32 offset0 = 4 q, offset1 = 4 q + 2 -- and no vector instruction computes an address inside a
    //  (tap, chunk).  The 16-byte pad per pixel row does what the XOR swizzle of rounds 2-6 did: pixel p, quad q lands on banks
    //  4 ((p + q) mod 8) + hi -- the same 4-way pattern a dword read of one quad position cannot avoid.)
    lds_t a_tap[MB];                                    // the lane's fragment base of the current tap
    auto set_tap = [&](int tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
        for (int i = 0; i < MB; ++i) a_tap[i] = opaque(lds + (uint32_t)(pp0[i] + kh * PW + kw) * Cfg::RS + (uint32_t)hi * 4u);
    };
    auto load_a = [&](int ch, int u, int set) {        // k-quad u of chunk ch of the tap set_tap() was last called for
        for (int i = 0; i < MB; ++i) {
            af[set][0][i] = lds_read_f32(a_tap[i] + (uint32_t)(ch * 8 + u) * 16u);
            af[set][1][i] = lds_read_f32(a_tap[i] + (uint32_t)(ch * 8 + u) * 16u + 8u);
        }
    };

    // RES: the residual operand of the epilogue is requested beside the MFMAs of the LAST kResTiles (tap, chunk)s -- where no
    // next B tile is left to fetch -- into registers the odd taps' P_t tile no longer needs (its last sum was taken in tap
    // 8's first chunk).  Requested in the epilogue instead, each (i, j) block of a wave waits a whole HBM round trip with
    // both workgroups of the CU in the same phase: 0.755 against 0.82 of the roof for the same layer without the residual.
    const int64_t tile_base = (((int64_t)b * p.Ho + ty0) * p.Wo + tx0) * p.Cout + n0;
    float* const yb = p.y + tile_base;
    const float* const rb = RES ? p.res + tile_base : p.y;
    // an element of the tile: the lane's part of the byte offset in a VGPR, the (j, i, r) part -- wave-uniform -- in an SGPR,
    // as for the B operand (no vector instruction per address)
    const uint32_t out_voff = (uint32_t)(4 * hi * p.Cout + wn * NB * 32 + l31) * 4u;
    auto out_soff = [&](int i, int r) -> uint32_t {
        const int pxu = (wm * MB + i) * 32 + (r & 3) + 8 * (r >> 2);           // the pixel without the lane half's + 4 hi (same tile row)
        return (uint32_t)((pxu / Cfg::TW) * p.Wo + pxu % Cfg::TW) * (uint32_t)p.Cout * 4u;
    };
    float bv[NB];                                               // (requested here: the epilogue does not wait for it)
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[j] = p.bias ? p.bias[n0 + (wn * NB + j) * 32 + l31] : 0.0f;
    constexpr int kResTiles = 1;
    constexpr int RQ = MB * NB * 16 / (8 * kResTiles);          // residual floats per lane requested per k-quad
    static_assert(RQ * 8 * kResTiles == MB * NB * 16 && RQ % 2 == 0 && kResTiles >= 1 && kResTiles <= 2 && NCH >= 4, "residual prefetch shape");
    float rv[NB][MB][16];
    auto load_res_half = [&](int q, int half) {                 // q: 0 .. 8 kResTiles - 1; half of the quad's RQ requests
#pragma unroll
        for (int k = half * (RQ / 2); k < (half + 1) * (RQ / 2); ++k) {
            const int e = q * RQ + k, j = e / (MB * 16), i = (e / 16) % MB, r = e % 16;
            rv[j][i][r] = global_load_f32_at(rb, out_voff + (uint32_t)j * 128u, out_soff(i, r));
        }
    };

    // P_t has two register tiles, by tap parity: the sum s = s + P_t of a finished tap is spread over the gaps of the
    // NEXT tap's first (tap, chunk) -- the matrix pipe never waits for it -- and the first MFMA of a tap takes C = 0
    // (no tile is ever zeroed).
    f32x16 acc[MB][NB], pt[2][MB][NB];
    for (int i = 0; i < MB; ++i)
        for (int j = 0; j < NB; ++j) acc[i][j] = zero_f32x16();

    // SET: B register set; PSET: P_t tile; FIRST: first (tap, chunk) of a tap; ADD: s = s + P_{t-1} rides along;
    // TAIL: 0 = a (tap, chunk) with a successor, t > 0 = the t-th of the last kResTiles ones (residual requests ride along),
    // the last of them without a successor.
    //
    // Issue order of a k-quad (a wave issues in order, and an MFMA holds the matrix pipe 64 cycles): every other instruction
    // sits BEHIND an MFMA of the quad, in its shadow -- the next quad's fragment read and half of the next tile's B requests
    // behind the first, the other half behind the second, the previous tap's sum behind the third -- so that the next MFMA is the first thing the wave wants when the pipe falls free.  With all of them in front
    // of the quad's four MFMAs (rounds 3-6) a wave needed ~100 cycles between its quads: a wave alone on its SIMD ran the loop
    // at 0.67 of the matrix rate, two at 0.91 (s_memtime stamps per wave, profiles/r06_conv_persistent.md).
    static_assert(MB == 1 && NB == 2, "the k-quad below is written out for four MFMAs");
    // CH: the chunk of the tap (compile time: the fragment offsets are immediates); PSET: P_t tile; ADD: s = s + P_{t-1} rides along
    auto tile = [&](int tap, auto ch_tag, auto pset_tag, auto add_tag, auto tail_tag) {
        constexpr int CH = decltype(ch_tag)::value, PSET = decltype(pset_tag)::value, TAIL = decltype(tail_tag)::value;
        constexpr bool FIRST = CH == 0, ADD = decltype(add_tag)::value != 0 && CH == 0, LAST = TAIL == kResTiles;
        const int it = tap * NCH + CH;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cs = u & 1, ns = cs ^ 1;
            const bool z = FIRST && u == 0;             // a tap's first MFMAs take C = 0
            auto request = [&](int half) {              // half of the quad's global requests: next tile's B floats, or the residual
                if (!LAST || u + kBAhead < 8) load_b_half_at(it, u + kBAhead, half);
                if constexpr (RES && TAIL > 0) load_res_half((TAIL - 1) * 8 + u, half);
            };
            pt[PSET][0][0] = mfma_32x32x2_f32(af[cs][0][0], bq[u % kBRing][0][0], z ? zero_f32x16() : pt[PSET][0][0]);
            sched_fence();
            if (u + 1 < 8) load_a(CH, u + 1, ns);
            else if constexpr (!LAST) {
                if constexpr (CH + 1 == NCH) {          // (the tap's last quad: its own fragment is in registers already)
                    set_tap(tap + 1);
                    load_a(0, 0, ns);
                } else load_a(CH + 1, 0, ns);
            }
            request(0);
            sched_fence();
            pt[PSET][0][1] = mfma_32x32x2_f32(af[cs][0][0], bq[u % kBRing][0][1], z ? zero_f32x16() : pt[PSET][0][1]);
            sched_fence();
            request(1);
            sched_fence();
            pt[PSET][0][0] = mfma_32x32x2_f32(af[cs][1][0], bq[u % kBRing][1][0], pt[PSET][0][0]);
            sched_fence();
            if (ADD && u >= 2 && u < 6) {       // a quarter of the previous tap's sum per k-quad (its last MFMA retired long ago)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 4 * (u - 2); r < 4 * (u - 2) + 4; ++r) acc[0][j][r] = acc[0][j][r] + pt[PSET ^ 1][0][j][r];
            }
            sched_fence();
            pt[PSET][0][1] = mfma_32x32x2_f32(af[cs][1][0], bq[u % kBRing][1][1], pt[PSET][0][1]);
            sched_fence();
        }
    };
    // one tap = NCH (tap, chunk)s: all but the last two, then those (every chunk index a compile-time constant)
    auto tap_head = [&](int tap, auto pset_tag, auto add_tag) {
        tile(tap, IntTag<0>{}, pset_tag, add_tag, IntTag<0>{});
        tile(tap, IntTag<1>{}, pset_tag, add_tag, IntTag<0>{});
        if constexpr (NCH > 4) {
            static_assert(NCH == 4 || NCH == 8, "chunks 2 .. NCH - 3 are written out");
            tile(tap, IntTag<2>{}, pset_tag, add_tag, IntTag<0>{});
            tile(tap, IntTag<3>{}, pset_tag, add_tag, IntTag<0>{});
            tile(tap, IntTag<4>{}, pset_tag, add_tag, IntTag<0>{});
            tile(tap, IntTag<5>{}, pset_tag, add_tag, IntTag<0>{});
        }
    };
    auto tap_tail = [&](int tap, auto pset_tag, auto lasttap_tag) {      // LASTTAP: the kernel's last tap (8)
        constexpr bool LASTTAP = decltype(lasttap_tag)::value != 0;
        tile(tap, IntTag<NCH - 2>{}, pset_tag, IntTag<0>{}, IntTag<(LASTTAP && kResTiles == 2) ? 1 : 0>{});
        tile(tap, IntTag<NCH - 1>{}, pset_tag, IntTag<0>{}, IntTag<LASTTAP ? kResTiles : 0>{});
    };
    set_tap(0);
    load_a(0, 0, 0);
    tap_head(0, IntTag<0>{}, IntTag<0>{});
    tap_tail(0, IntTag<0>{}, IntTag<0>{});
    for (int tap = 1;; tap += 2) {
        tap_head(tap, IntTag<1>{}, IntTag<1>{});
        tap_tail(tap, IntTag<1>{}, IntTag<0>{});
        // The two waves of a SIMD (w and w + NW/2) want the matrix pipe all the time, and the arbiter gives it to the older
        // one first: it ran its tile in 302 k cycles and its partner finished 31 k cycles later, alone on the SIMD, at two
        // thirds of the matrix rate (s_memtime per wave, profiles/r06_conv_persistent.md).  From its tap 4 on the younger
        // half holds priority 1 -- it takes over the older half's share for the rest of the tile and both reach the
        // tile's closing barrier together.
        if (tap + 1 == kPrioTap && wave >= NW / 2) wave_priority(1);
        tap_head(tap + 1, IntTag<0>{}, IntTag<1>{});
        if (tap + 1 == 8) break;                        // (tap 8's tail differs: it stands behind the loop)
        tap_tail(tap + 1, IntTag<0>{}, IntTag<0>{});
    }
    tap_tail(8, IntTag<0>{}, IntTag<1>{});
    for (int i = 0; i < MB; ++i)                        // the last tap (8: parity 0)
        for (int j = 0; j < NB; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + pt[0][i][j][r];

    // ---- epilogue: + bias [+ residual, already in registers] [clip]: each step over the whole tile behind ONE uniform branch
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + bv[j];
    }
    if constexpr (RES) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + rv[j][i][r];
    }
    if (p.clip) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r];
                    acc[i][j][r] = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
                }
    }
    if (wave >= NW / 2) wave_priority(0);
    // every wave has read its last fragment of this patch: the next tile's patch is requested BEFORE the results are
    // written, and lands while they are
    block_sync_lds();
    const bool more = tl + grid_dim_x() < ntiles;
    if (more) zmask = request_patch(tile_at(tl + grid_dim_x()));
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) global_store_f32_at(yb, out_voff + (uint32_t)j * 128u, out_soff(i, r), acc[i][j][r]);
    if (!more) break;
    }
}

// ---------------------------------------------------------------------------------------------------
// The two ends of the network, where the generic kernel pads one GEMM dimension tenfold:
//
// conv_in (3 -> 128 at 256 x 256, lwm/vqgan.py:155; 64-column tiles, two waves per SIMD): K per tap is Cin = 3, not a 32-channel chunk.  One k-quad per
// tap -- two MFMAs per (tap, 32 x 32 block) instead of sixteen, the fourth channel an exact zero -- with the A operand
// read straight from global memory (12 bytes per pixel: the image stays in L1/L2) and the B operand, all 9 x 4 x NB
// floats of a lane's columns, resident in registers.  No LDS, no barrier; bound by the output write.
// Same arithmetic and order as the generic kernel: P_t = the fma chain over c_in from 0 (zeros add nothing),
// s = s + P_t in tap order.
constexpr int kCinPB = 4;        // 32-pixel blocks per wave
template <int NB>
LWM_DEVICE void conv_cin4_body(const ConvParams& p) {
    const int tid = thread_idx();
    const int wave = wave_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int ntn = p.Cout / (32 * NB);
    const int bn = block_idx_x() % ntn;
    const int64_t bm = block_idx_x() / ntn;
    const int n0 = bn * 32 * NB;

    float bq[9][2][NB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ci = 2 * t + hi;
            const bool ok = ci < p.Cin;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float v = p.w[(int64_t)(tap * p.Cin + (ok ? ci : 0)) * p.Cout + n0 + j * 32 + l31];
                bq[tap][t][j] = ok ? v : 0.0f;
            }
        }
    float bv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[j] = p.bias ? p.bias[n0 + j * 32 + l31] : 0.0f;
    const int Hv = p.Hin << p.up_shift, Wv = p.Win << p.up_shift;

#pragma unroll 1
    for (int pb = 0; pb < kCinPB; ++pb) {
        const int64_t mb = (bm * 4 + wave) * (kCinPB * 32) + pb * 32;      // first pixel of the block (uniform)
        if (mb >= p.M) break;
        const int64_t m = mb + l31 < p.M ? mb + l31 : p.M - 1;
        const bool mok = mb + l31 < p.M;
        const int ox = (int)(m % p.Wo);
        const int64_t tq = m / p.Wo;
        const int oy = (int)(tq % p.Ho);
        const float* xb = p.x + (tq / p.Ho) * (int64_t)p.Hin * p.Win * p.Cin;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        float av[9][2];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            const int vy = iy0 + kh, vx = ix0 + kw;
            const bool ok = mok && vy >= 0 && vy < Hv && vx >= 0 && vx < Wv;
            const int off = ok ? ((vy >> p.up_shift) * p.Win + (vx >> p.up_shift)) * p.Cin : 0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ci = 2 * t + hi;
                const float v = xb[off + (ci < p.Cin ? ci : 0)];       // unconditional load, select afterwards
                av[tap][t] = (ok && ci < p.Cin) ? v : 0.0f;
            }
        }
        f32x16 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = zero_f32x16();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            f32x16 pt[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) pt[j] = mfma_32x32x2_f32(av[tap][0], bq[tap][0][j], zero_f32x16());
#pragma unroll
            for (int j = 0; j < NB; ++j) pt[j] = mfma_32x32x2_f32(av[tap][1], bq[tap][1][j], pt[j]);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                for (int r = 0; r < 16; ++r) acc[j][r] = acc[j][r] + pt[j][r];
            sched_fence();      // (the taps are independent: left alone, hipcc issues all 72 MFMAs first -- 36 tuples live)
        }
        float* const yb = p.y + mb * p.Cout + n0;
        const float* const rb = p.res ? p.res + mb * p.Cout + n0 : p.y;
        const int64_t left = p.M - mb;
        const int rows = left < 32 ? (int)left : 32;
        auto store = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value != 0;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const uint32_t off = (uint32_t)((FULL || ml < rows) ? ml : rows - 1) * (uint32_t)p.Cout + (uint32_t)(j * 32 + l31);
                    float v = acc[j][r];
                    if (p.bias) v = v + bv[j];
                    if (p.res) v = v + rb[off];
                    if (p.clip) v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
                    if (FULL || ml < rows) yb[off] = v;
                }
        };
        if (rows == 32) store(IntTag<1>{});
        else store(IntTag<0>{});
    }
}
LWM_KERNEL_OCC(256, 2) void conv_cin4_n64(ConvParams p) { conv_cin4_body<2>(p); }

// conv_out (128 -> 3 at 256 x 256, lwm/vqgan.py:185 / the decoder's last layer): N is Cout = 3, not a 32-column MFMA
// block.  On the vector pipe: a lane owns one pixel of a 4 x 16 tile whose halo patch sits in LDS exactly as for
// conv_patch_c128; the weights of a (tap, c_in) are wave-uniform -- scalar loads, SGPR operands of v_fma_f32 -- and
// the four waves take the taps round robin (P_t are independent chains), leave them in LDS, and wave 0 adds them in
// tap order.  v_fma_f32 is the fma the MFMA performs per k step: same bits.
template <int CIN, int COUT>
struct PatchOutCfg {
    using Patch = PatchCfg<CIN, 4, 2, 2, 1>;                 // 64 pixels, 4 waves (the tile shape is all that is used)
    static constexpr int OFF_PART = Patch::LDS_BYTES;          // [9 taps][64 pixels] x 16 bytes
    static constexpr int LDS_BYTES = OFF_PART + 9 * 64 * 16;
    static_assert(COUT >= 1 && COUT <= 4, "one 16-byte partial per pixel and tap");
};
template <int CIN, int COUT>
LWM_DEVICE void conv_patch_cout_body(const ConvParams& p) {
    using Cfg = PatchOutCfg<CIN, COUT>;
    using PC = typename Cfg::Patch;
    const lds_t lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = wave_uniform(tid >> 6), lane = tid & 63;
    int64_t bm = block_idx_x();
    const int tiles_x = p.Wo / PC::TW, tiles_y = p.Ho / 4;
    const int tx0 = (int)(bm % tiles_x) * PC::TW;
    bm /= tiles_x;
    const int ty0 = (int)(bm % tiles_y) * 4;
    const int b = (int)(bm / tiles_y);
    conv_patch_fill<PC>(p, p.x + (int64_t)b * p.Hin * p.Win * CIN, ty0, tx0, wave, lane, lds);

    const int pp0 = (lane / PC::TW) * PC::PW + (lane % PC::TW);
    for (int tap = wave; tap < 9; tap += 4) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int pp = pp0 + kh * PC::PW + kw;
        const float* wt = p.w + (int64_t)tap * CIN * COUT;          // wave-uniform
        const lds_t row = lds + (uint32_t)pp * PC::RS;
        float s[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) s[co] = 0.0f;
#pragma unroll
        for (int q = 0; q < CIN / 4; ++q) {
            const f32x4 a = lds_read_f32x4(row + (uint32_t)(q << 4));
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int co = 0; co < COUT; ++co) s[co] = fmaf(a[c], uniform_load_f32(wt, (4 * q + c) * COUT + co), s[co]);
        }
        f32x4 o = zero_f32x4();
#pragma unroll
        for (int co = 0; co < COUT; ++co) o[co] = s[co];
        lds_write_f32x4(lds + Cfg::OFF_PART + (uint32_t)(tap * 64 + lane) * 16, o);
    }
    block_sync();
    if (wave == 0) {
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const f32x4 o = lds_read_f32x4(lds + Cfg::OFF_PART + (uint32_t)(tap * 64 + lane) * 16);
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[co] = acc[co] + o[co];
        }
        const int64_t pix = ((int64_t)b * p.Ho + ty0 + lane / PC::TW) * p.Wo + tx0 + lane % PC::TW;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float v = acc[co];
            if (p.bias) v = v + p.bias[co];
            if (p.res) v = v + p.res[pix * COUT + co];
            if (p.clip) v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
            p.y[pix * COUT + co] = v;
        }
    }
}
using PatchOutC128 = PatchOutCfg<128, 3>;
LWM_KERNEL_OCC(256, 2) void conv_patch_c128_out3(ConvParams p) { conv_patch_cout_body<128, 3>(p); }

// (round 4, one workgroup per tile: an 8 x 16-pixel tile with 8 waves for 128 input channels -- 90 KiB, one workgroup per
// CU -- measured 2-3 % slower than two 4 x 16 workgroups; in the persistent form it is the faster one: two waves per SIMD
// of ONE workgroup run the main loop at 0.976 of the matrix rate, one wave per SIMD beside another workgroup's prologue at
// 0.64-0.70.  64 x 128-channel tiles with 8 waves for 256 input channels 12 % slower than 64 x 256; a 2 x 16-pixel x
// 256-channel tile for 512 input channels -- 144 KiB -- no faster than the generic kernel: its B stream needs 16 bytes
// per clock and CU from L2)
using PatchC128 = PatchCfg<128, 8, 4, 2, 2>;   // 128 pixels x 128 channels, 8 waves, 90 KiB
using PatchC256 = PatchCfg<256, 4, 2, 4, 2>;   // 64 pixels x 256 channels, 8 waves, 108 KiB
LWM_KERNEL(512) void conv_patch_c128(ConvParams p) { conv_patch_body<128, 8, 4, 2, 2, false>(p); }
LWM_KERNEL(512) void conv_patch_c256(ConvParams p) { conv_patch_body<256, 4, 2, 4, 2, false>(p); }
// (+ residual: the ResnetBlock's second convolution, lwm/vqgan.py:263)
LWM_KERNEL(512) void conv_patch_c128_res(ConvParams p) { conv_patch_body<128, 8, 4, 2, 2, true>(p); }
LWM_KERNEL(512) void conv_patch_c256_res(ConvParams p) { conv_patch_body<256, 4, 2, 4, 2, true>(p); }
// (256 -> 128 channels with this patch and 4 waves x (32 pixels x 64 channels), one wave per SIMD: 92.9 TF/s against
// 102.3 for the generic kernel and 108 for its B-direct form: dropped)

LWM_KERNEL_OCC(256, 2) void conv_igemm_128x128(ConvParams p) { conv_igemm_body<2, 2, 2, 2, true>(p); }
LWM_KERNEL_OCC(256, 2) void conv_igemm_128x128_bd(ConvParams p) { conv_igemm_body<2, 2, 2, 2, true, true>(p); }
LWM_KERNEL(256) void conv_igemm_128x64(ConvParams p) { conv_igemm_body<4, 1, 1, 2, true>(p); }
LWM_KERNEL(256) void conv_igemm_32x128(ConvParams p) { conv_igemm_body<1, 4, 1, 1, true>(p); }
// generic (scalar staging) forms: Cin or Cout not a multiple of 4
LWM_KERNEL_OCC(256, 2) void conv_igemm_128x128_g(ConvParams p) { conv_igemm_body<2, 2, 2, 2, false>(p); }
LWM_KERNEL(256) void conv_igemm_128x64_g(ConvParams p) { conv_igemm_body<4, 1, 1, 2, false>(p); }
LWM_KERNEL(256) void conv_igemm_128x32_g(ConvParams p) { conv_igemm_body<4, 1, 1, 1, false>(p); }
LWM_KERNEL(256) void conv_igemm_32x128_g(ConvParams p) { conv_igemm_body<1, 4, 1, 1, false>(p); }

}  // namespace lwm
