// attn_bwd.h -- blockwise attention backward for one (q block, kv block) ring
// step on gfx950.  Requires wave_ops.h + attn_common.h.
//
// Replaces the custom-VJP backward of `ringattention` (call site
// lwm/llama.py:539-569): recompute p from the saved
// LSE, dv += p^T do, dp = do v^T, ds = p*(dp - rowsum(do*o)), dq += ds k,
// dk += ds^T q, all scaled by 1/sqrt(D) where the reference does.
//
// Three kernels, no atomics, deterministic:
//   attn_bwd_delta_kernel : delta[b,h,q] = sum_d dO*O               (HBM bound)
//   attn_bwd_dq_kernel    : workgroup owns 256 queries, streams K/V (3 GEMMs)
//   attn_bwd_dkdv_kernel  : workgroup owns 256 keys, streams Q/dO   (4 GEMMs)
// The f32 *_acc carries let a ring driver accumulate dq locally and dk/dv in
// buffers that travel with the K/V block.
#pragma once

namespace lwm {

// ------------------------------------------------------------------ delta
constexpr int kDeltaThreads = 256;

LWM_KERNEL(kDeltaThreads) void attn_bwd_delta_kernel(AttnParams p, float* delta) {
    const int tid = thread_idx();
    const int64_t rows = (int64_t)p.B * p.H * p.Sq;
    const int part = tid & 15;
    int64_t row = (int64_t)block_idx_x() * (kDeltaThreads / 16) + (tid >> 4);
    const int64_t row_step = (int64_t)grid_dim_x() * (kDeltaThreads / 16);
    // every lane runs the same number of iterations (shuffles need full waves)
    const int64_t iters = (rows + row_step - 1) / row_step;
    for (int64_t it = 0; it < iters; ++it, row += row_step) {
        float s = 0.0f;
        const bool ok = row < rows;
        int64_t b = 0, h = 0, q = 0;
        if (ok) {
            // row = (b*H + h)*Sq + q   (matches the [B,H,Sq] layout of delta/lse)
            q = row % p.Sq;
            int64_t bh = row / p.Sq;
            h = bh % p.H;
            b = bh / p.H;
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
        if (ok && part == 0) delta[row] = s;
    }
}

// ------------------------------------------------------------------ dQ
constexpr int kDqBQ = 256;
constexpr int kDqBK = 32;
constexpr int kDqThreads = 512;
constexpr int kDqTileBytes = kDqBK * kRowBytes;                 // 8 KiB
constexpr int kDqLdsBytes = 4 * kDqTileBytes + 2 * kDqBK * 4;  // K,V x2 + kseg x2

struct DqStage {
    u32x4 k;
    u32x4 v;
    int32_t kseg;
};

LWM_DEVICE void dq_stage_load(const AttnParams& p, const bf16_t* kb, const bf16_t* vb, int b,
                              int kt, int tid, DqStage& st) {
    int row = tid >> 4, slot = tid & 15;
    int krow = kt * kDqBK + row;
    if (krow < p.Sk) {
        st.k = global_load_b128(kb + (int64_t)krow * p.k_ss + slot * 8);
        st.v = global_load_b128(vb + (int64_t)krow * p.v_ss + slot * 8);
    } else {
        u32x4 z = {0u, 0u, 0u, 0u};
        st.k = z;
        st.v = z;
    }
    if (tid < kDqBK) {
        int kr = kt * kDqBK + tid;
        int32_t s = kSegInvalid;
        if (kr < p.Sk) {
            bool valid = p.key_valid ? (p.key_valid[(int64_t)b * p.Sk + kr] != 0) : true;
            if (valid) s = p.seg_k ? p.seg_k[(int64_t)b * p.Sk + kr] : 0;
        }
        st.kseg = s;
    }
}

LWM_DEVICE void dq_stage_write(char* kbuf, char* vbuf, int32_t* ksegbuf, int tid,
                               const DqStage& st) {
    int row = tid >> 4, slot = tid & 15;
    lds_write_b128(kbuf + tile_off(row, slot), st.k);
    lds_write_b128(vbuf + tile_off(row, slot), st.v);
    if (tid < kDqBK) ksegbuf[tid] = st.kseg;
}

LWM_KERNEL(kDqThreads) void attn_bwd_dq_kernel(AttnParams p) {
    char* lds = dyn_lds();
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

    char* kbuf[2] = {lds, lds + kDqTileBytes};
    char* vbuf[2] = {lds + 2 * kDqTileBytes, lds + 3 * kDqTileBytes};
    int32_t* ksegbuf[2] = {(int32_t*)(lds + 4 * kDqTileBytes),
                           (int32_t*)(lds + 4 * kDqTileBytes) + kDqBK};

    const int q_row = qt * kDqBQ + wave * 32 + l31;
    const bool q_ok = q_row < p.Sq;
    const int64_t q_pos = p.q_start + q_row;
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
    const int64_t stat_idx = ((int64_t)b * p.H + h) * p.Sq + q_row;
    float lse2 = INFINITY, dlt = 0.0f;
    if (q_ok) {
        float l = p.lse[stat_idx];
        lse2 = (l == -INFINITY) ? INFINITY : l * kLog2e;
        dlt = p.delta[stat_idx];
    }
    const int32_t seg_q = (q_ok && p.seg_q) ? p.seg_q[(int64_t)b * p.Sq + q_row] : 0;
    const bool has_kmeta = (p.seg_k != nullptr) || (p.key_valid != nullptr) || (p.Sk % kDqBK != 0);

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
    const int64_t wq_min = p.q_start + qt * kDqBQ + wave * 32;
    const int64_t wq_max = wq_min + 31;

    const float c = p.scale * kLog2e;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = zero_f32x16();

    DqStage stg;
    if (nkt > 0) {
        dq_stage_load(p, kb, vb, b, 0, tid, stg);
        dq_stage_write(kbuf[0], vbuf[0], ksegbuf[0], tid, stg);
    }
    block_sync();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) dq_stage_load(p, kb, vb, b, kt + 1, tid, stg);

        const int64_t k_pos0 = p.k_start + (int64_t)kt * kDqBK;
        const bool wave_active = !p.causal || k_pos0 <= wq_max;
        if (wave_active) {
            f32x16 st = zero_f32x16(), dpt = zero_f32x16();
            for (int s = 0; s < 8; ++s) {
                bf16x8 a = frag_rows(kbuf[cur], 0, s, l31, hi);
                st = mfma_32x32x16(a, qf[s], st);
            }
            for (int s = 0; s < 8; ++s) {
                bf16x8 a = frag_rows(vbuf[cur], 0, s, l31, hi);
                dpt = mfma_32x32x16(a, dof[s], dpt);
            }
            const bool need_mask = has_kmeta || (p.causal && k_pos0 + kDqBK - 1 > wq_min);
            const int32_t* ks = ksegbuf[cur];
            for (int r = 0; r < 16; ++r) {
                float pv = fast_exp2(fmaf(st[r], c, -lse2));
                if (need_mask) {
                    int kl = cd_row(r, hi);
                    bool vis = (ks[kl] == seg_q);
                    if (p.causal) vis = vis && (k_pos0 + kl <= q_pos);
                    pv = vis ? pv : 0.0f;
                }
                st[r] = pv * (dpt[r] - dlt);  // dS^T (unscaled)
            }
            for (int t = 0; t < 2; ++t) {
                bf16x8 dsb = cvt_frag(st, 8 * t);
                for (int db = 0; db < 4; ++db) {
                    bf16x8 a = frag_cols_tr(kbuf[cur], 16 * t, 32 * db, lane);
                    acc[db] = mfma_32x32x16(a, dsb, acc[db]);
                }
            }
        }
        if (more) dq_stage_write(kbuf[cur ^ 1], vbuf[cur ^ 1], ksegbuf[cur ^ 1], tid, stg);
        block_sync();
    }

    if (q_ok) {
        const int64_t orow = (int64_t)b * p.dq_sb + (int64_t)q_row * p.dq_ss + (int64_t)h * p.dq_sh;
        const int64_t arow = (((int64_t)b * p.Sq + q_row) * p.H + h) * kHeadDim;
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

// ------------------------------------------------------------------ dK, dV
constexpr int kDkvBK = 256;   // keys per workgroup (32 per wave)
constexpr int kDkvBQ = 32;    // queries per LDS tile
constexpr int kDkvThreads = 512;
constexpr int kDkvQTileBytes = kDkvBQ * kRowBytes;  // 8 KiB
constexpr int kDkvVBytes = kDkvBK * kRowBytes;      // 64 KiB
// V (resident) | Q x2 | dO x2 | stats x2 (lse2, delta, seg_q : 32 each)
constexpr int kDkvStatBytes = 3 * kDkvBQ * 4;
constexpr int kDkvLdsBytes = kDkvVBytes + 4 * kDkvQTileBytes + 2 * kDkvStatBytes;

struct DkvStage {
    u32x4 q;
    u32x4 d;
    float lse2, delta;
    int32_t segq;
};

LWM_DEVICE void dkv_stage_load(const AttnParams& p, const bf16_t* qb, const bf16_t* dob, int b,
                               int h, int qt, int tid, DkvStage& st) {
    int row = tid >> 4, slot = tid & 15;
    int qrow = qt * kDkvBQ + row;
    if (qrow < p.Sq) {
        st.q = global_load_b128(qb + (int64_t)qrow * p.q_ss + slot * 8);
        st.d = global_load_b128(dob + (int64_t)qrow * p.do_ss + slot * 8);
    } else {
        u32x4 z = {0u, 0u, 0u, 0u};
        st.q = z;
        st.d = z;
    }
    if (tid < kDkvBQ) {
        int qr = qt * kDkvBQ + tid;
        st.lse2 = INFINITY;
        st.delta = 0.0f;
        st.segq = 0;
        if (qr < p.Sq) {
            int64_t idx = ((int64_t)b * p.H + h) * p.Sq + qr;
            float l = p.lse[idx];
            st.lse2 = (l == -INFINITY) ? INFINITY : l * kLog2e;
            st.delta = p.delta[idx];
            st.segq = p.seg_q ? p.seg_q[(int64_t)b * p.Sq + qr] : 0;
        }
    }
}

LWM_DEVICE void dkv_stage_write(char* qbuf, char* dbuf, char* statbuf, int tid,
                                const DkvStage& st) {
    int row = tid >> 4, slot = tid & 15;
    lds_write_b128(qbuf + tile_off(row, slot), st.q);
    lds_write_b128(dbuf + tile_off(row, slot), st.d);
    if (tid < kDkvBQ) {
        ((float*)statbuf)[tid] = st.lse2;
        ((float*)statbuf)[kDkvBQ + tid] = st.delta;
        ((int32_t*)statbuf)[2 * kDkvBQ + tid] = st.segq;
    }
}

LWM_KERNEL(kDkvThreads) void attn_bwd_dkdv_kernel(AttnParams p) {
    char* lds = dyn_lds();
    const int tid = thread_idx();
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;

    const int nkb = (p.Sk + kDkvBK - 1) / kDkvBK;
    const int HB = p.H * p.B;
    int lin = block_idx_x(), kbi, hb;
    if ((HB & 7) == 0) {
        int xcd = lin & 7, i = lin >> 3;
        hb = xcd + 8 * (i / nkb);
        kbi = i % nkb;  // causal: low key blocks have the longest q loops -> first
    } else {
        hb = lin / nkb;
        kbi = lin % nkb;
    }
    const int b = hb / p.H, h = hb % p.H;

    const bf16_t* qb = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* kb = p.k + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const bf16_t* vb = p.v + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const bf16_t* dob = p.dout + (int64_t)b * p.do_sb + (int64_t)h * p.do_sh;

    char* vtile = lds;
    char* qbuf[2] = {lds + kDkvVBytes, lds + kDkvVBytes + kDkvQTileBytes};
    char* dbuf[2] = {lds + kDkvVBytes + 2 * kDkvQTileBytes, lds + kDkvVBytes + 3 * kDkvQTileBytes};
    char* statbuf[2] = {lds + kDkvVBytes + 4 * kDkvQTileBytes,
                        lds + kDkvVBytes + 4 * kDkvQTileBytes + kDkvStatBytes};

    // ---- this lane's key
    const int k_row = kbi * kDkvBK + wave * 32 + l31;
    const bool k_ok = k_row < p.Sk;
    const int64_t k_pos = p.k_start + k_row;
    bf16x8 kf[8];
    for (int s = 0; s < 8; ++s) {
        if (k_ok)
            kf[s] = __builtin_bit_cast(
                bf16x8, global_load_b128(kb + (int64_t)k_row * p.k_ss + 16 * s + 8 * hi));
        else
            kf[s] = zero_bf16x8();
    }
    int32_t kseg = kSegInvalid;
    if (k_ok) {
        bool valid = p.key_valid ? (p.key_valid[(int64_t)b * p.Sk + k_row] != 0) : true;
        if (valid) kseg = p.seg_k ? p.seg_k[(int64_t)b * p.Sk + k_row] : 0;
    }
    const bool has_meta =
        (p.seg_k != nullptr) || (p.key_valid != nullptr) || (kbi * kDkvBK + kDkvBK > p.Sk);

    // ---- resident V tile (this workgroup's 256 keys)
    for (int i = 0; i < 8; ++i) {
        int cidx = tid + kDkvThreads * i;
        int row = cidx >> 4, slot = cidx & 15;
        int kr = kbi * kDkvBK + row;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (kr < p.Sk) val = global_load_b128(vb + (int64_t)kr * p.v_ss + slot * 8);
        lds_write_b128(vtile + tile_off(row, slot), val);
    }

    // ---- q tile range (causal: skip q tiles wholly before this key block)
    const int nqt = (p.Sq + kDkvBQ - 1) / kDkvBQ;
    int qt0 = 0;
    if (p.causal) {
        int64_t d = p.k_start + (int64_t)kbi * kDkvBK - p.q_start;  // first q row that can see key 0
        if (d > 0) qt0 = (int)(d / kDkvBQ < nqt ? d / kDkvBQ : nqt);
    }
    const int64_t wk_min = p.k_start + (int64_t)kbi * kDkvBK + wave * 32;
    const int64_t wk_max = wk_min + 31;

    const float c = p.scale * kLog2e;
    f32x16 dk[4], dv[4];
    for (int i = 0; i < 4; ++i) {
        dk[i] = zero_f32x16();
        dv[i] = zero_f32x16();
    }

    DkvStage stg;
    if (qt0 < nqt) {
        dkv_stage_load(p, qb, dob, b, h, qt0, tid, stg);
        dkv_stage_write(qbuf[0], dbuf[0], statbuf[0], tid, stg);
    }
    block_sync();

    for (int qt = qt0; qt < nqt; ++qt) {
        const int cur = (qt - qt0) & 1;
        const bool more = qt + 1 < nqt;
        if (more) dkv_stage_load(p, qb, dob, b, h, qt + 1, tid, stg);

        const int64_t q_pos0 = p.q_start + (int64_t)qt * kDkvBQ;
        const bool wave_active = !p.causal || q_pos0 + kDkvBQ - 1 >= wk_min;
        if (wave_active) {
            // S = Q K^T and dP = dO V^T  (rows = queries, cols = keys)
            f32x16 s = zero_f32x16(), dp = zero_f32x16();
            for (int st = 0; st < 8; ++st) {
                bf16x8 a = frag_rows(qbuf[cur], 0, st, l31, hi);
                s = mfma_32x32x16(a, kf[st], s);
            }
            for (int st = 0; st < 8; ++st) {
                bf16x8 a = frag_rows(dbuf[cur], 0, st, l31, hi);
                bf16x8 vfr = frag_rows(vtile, wave * 32, st, l31, hi);
                dp = mfma_32x32x16(a, vfr, dp);
            }
            const bool need_mask = has_meta || (p.causal && q_pos0 < wk_max);
            const float* lse2s = (const float*)statbuf[cur];
            const float* dlts = lse2s + kDkvBQ;
            const int32_t* segs = (const int32_t*)statbuf[cur] + 2 * kDkvBQ;
            f32x16 ds;
            for (int r = 0; r < 16; ++r) {
                int ql = cd_row(r, hi);
                float pv = fast_exp2(fmaf(s[r], c, -lse2s[ql]));
                if (need_mask) {
                    bool vis = (segs[ql] == kseg);
                    if (p.causal) vis = vis && (k_pos <= q_pos0 + ql);
                    pv = vis ? pv : 0.0f;
                }
                s[r] = pv;
                ds[r] = pv * (dp[r] - dlts[ql]);
            }
            for (int t = 0; t < 2; ++t) {
                bf16x8 pb = cvt_frag(s, 8 * t);
                bf16x8 dsb = cvt_frag(ds, 8 * t);
                for (int db = 0; db < 4; ++db) {
                    bf16x8 a = frag_cols_tr(dbuf[cur], 16 * t, 32 * db, lane);
                    dv[db] = mfma_32x32x16(a, pb, dv[db]);
                }
                for (int db = 0; db < 4; ++db) {
                    bf16x8 a = frag_cols_tr(qbuf[cur], 16 * t, 32 * db, lane);
                    dk[db] = mfma_32x32x16(a, dsb, dk[db]);
                }
            }
        }
        if (more) dkv_stage_write(qbuf[cur ^ 1], dbuf[cur ^ 1], statbuf[cur ^ 1], tid, stg);
        block_sync();
    }

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
                    u32x2 pk = {pack_bf16x2(k0, k1), pack_bf16x2(k2, k3)};
                    u32x2 pv = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
                    global_store_b64(p.dk + krow_o + d0, pk);
                    global_store_b64(p.dv + vrow_o + d0, pv);
                } else {
                    u32x4 pk = {__builtin_bit_cast(uint32_t, k0), __builtin_bit_cast(uint32_t, k1),
                                __builtin_bit_cast(uint32_t, k2), __builtin_bit_cast(uint32_t, k3)};
                    u32x4 pv = {__builtin_bit_cast(uint32_t, v0), __builtin_bit_cast(uint32_t, v1),
                                __builtin_bit_cast(uint32_t, v2), __builtin_bit_cast(uint32_t, v3)};
                    global_store_b128(p.dk_acc + arow + d0, pk);
                    global_store_b128(p.dv_acc + arow + d0, pv);
                }
            }
    }
}

}  // namespace lwm
