// attn_decode.h -- cached-decode attention (one query per batch row) as a streaming
// GEMV over the K/V cache.  Requires wave_ops.h + attn_common.h.
//
// Replaces ringattention_inference for q_len == 1 (lwm/llama.py:571-614: the decode
// step reads the whole (B, max_len/sp, H, D) cache shard once).  HBM-bound: the
// algorithmic traffic is the K and V shards read once, 2*Sk*H*D*2 bytes.
//
// The cache layout is the reference's (B, S, H, D): for one key ALL heads are
// contiguous (H*256 B), so a workgroup streams whole key rows -- every wave load is
// 1 KiB contiguous -- instead of striding through one head.  Workgroup = 8 waves; a
// 16-lane group owns one head (lane i holds d = 8i..8i+7 of q, of the running output
// and of each K/V row), a wave 4 heads, the workgroup 32 heads per pass.  The online
// softmax state (m, l) is replicated in the 16 lanes of a group; scores are reduced
// with 4 xor-shuffles.  Keys are processed 4 at a time (8 x 16-byte loads in flight
// per lane).  Each workgroup handles one contiguous piece of the key range and
// writes a normalised partial (out f32, lse) -- merged by attn_combine_kernel
// exactly like the split-K pieces of the MFMA kernel.
#pragma once

namespace lwm {

constexpr int kDecThreads = 512;
constexpr int kDecUnroll = 4;     // (2 and 8 measured: 368.7 / 358.2 us against 356.1 at 131072 keys, 512 pieces)

LWM_DEVICE void unpack_bf16x8(u32x4 raw, float (&f)[8]) {
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __builtin_bit_cast(float, raw[j] << 16);
        f[2 * j + 1] = __builtin_bit_cast(float, raw[j] & 0xffff0000u);
    }
}

LWM_KERNEL(kDecThreads) void attn_decode_kernel(AttnParams p) {
    const int tid = thread_idx();
    const int grp = tid >> 4, li = tid & 15;       // 32 head slots per pass, 16 lanes each
    const int nsplit = p.k_splits > 1 ? p.k_splits : 1;
    const int b = block_idx_x() / nsplit, split = block_idx_x() % nsplit;
    const int per = (p.Sk + nsplit - 1) / nsplit;
    const int k0 = split * per;
    const int k1 = k0 + per < p.Sk ? k0 + per : p.Sk;
    const float c = p.scale * kLog2e;
    const uint8_t* mrow = p.dense_mask ? p.dense_mask + (int64_t)b * p.msk_sb : nullptr;

    // Visible key range.  The reference attends over the WHOLE cache under the mask (kv_len = max_length,
    // lwm/llama.py:571-614); the keys beyond cache_index (and padding at either end) are masked.  Every
    // workgroup scans the mask row -- 1 B per key, L2-resident, against 8 KiB of K/V per key -- finds
    // [first, last] visible, and the k_splits pieces partition THAT range instead of [0, Sk): a generation
    // that has filled 2K of a 32K cache moves 1/16 of the bytes and still spreads them over every
    // workgroup (a piece's keys are a serial chain of HBM round trips, so few long pieces are slow).
    // Holes inside the range are handled per key below.  Nothing visible -> every piece writes (0, -inf).
    int ka = k0, kz = k1;
    if (mrow) {
        int first = 0x7fffffff, last = -1;
        // 16 mask bytes per load where the row allows it (a byte-wise scan of a 131072-key row by all 512
        // pieces cost ~100 us); lowest / highest nonzero byte of a word from its lowest / highest set bit
        const int nvec = (((uintptr_t)mrow & 15) == 0) ? (p.Sk >> 4) : 0;
        for (int i = tid; i < nvec; i += kDecThreads) {
            const u32x4 w = global_load_b128(mrow + 16 * i);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (w[c] != 0u) {
                    const int lo = 16 * i + 4 * c + (__builtin_ctz(w[c]) >> 3);
                    const int hi = 16 * i + 4 * c + ((31 - __builtin_clz(w[c])) >> 3);
                    first = lo < first ? lo : first;
                    last = hi > last ? hi : last;
                }
        }
        for (int j = 16 * nvec + tid; j < p.Sk; j += kDecThreads)
            if (mrow[j] != 0) {
                first = j < first ? j : first;
                last = j > last ? j : last;
            }
        for (int msk = 1; msk < 64; msk <<= 1) {
            const int of = shfl_xor_i(first, msk), ol = shfl_xor_i(last, msk);
            first = of < first ? of : first;
            last = ol > last ? ol : last;
        }
        const lds_t scan = dyn_lds();
        if ((tid & 63) == 0) {
            lds_write_i32(scan + (tid >> 6) * 8, first);
            lds_write_i32(scan + (tid >> 6) * 8 + 4, last);
        }
        block_sync();
        for (int w = 0; w < kDecThreads / 64; ++w) {
            const int of = lds_read_i32(scan + w * 8), ol = lds_read_i32(scan + w * 8 + 4);
            first = of < first ? of : first;
            last = ol > last ? ol : last;
        }
        if (last < 0) {
            ka = kz = 0;
        } else {
            const int nv = last - first + 1;
            const int pv = (nv + nsplit - 1) / nsplit;
            ka = first + split * pv;
            kz = ka + pv < last + 1 ? ka + pv : last + 1;
            if (ka > kz) ka = kz;
        }
    }

    for (int h0 = 0; h0 < p.H; h0 += 32) {
        const int h = h0 + grp;
        const bool h_ok = h < p.H;
        const int hc = h_ok ? h : p.H - 1;         // clamped: loads stay in bounds
        float qf[8], o[8];
        unpack_bf16x8(global_load_b128(p.q + (int64_t)b * p.q_sb + (int64_t)hc * p.q_sh + li * 8), qf);
        for (int j = 0; j < 8; ++j) {
            qf[j] *= c;                            // scores directly in log2 units
            o[j] = 0.0f;
        }
        float m = -INFINITY, l = 0.0f;
        const bf16_t* kb = p.k + (int64_t)b * p.k_sb + (int64_t)hc * p.k_sh + li * 8;
        const bf16_t* vb = p.v + (int64_t)b * p.v_sb + (int64_t)hc * p.v_sh + li * 8;
        // Every piece starts at a different phase of its key range (softmax accumulation is
        // order-free): pieces are a power-of-two number of bytes apart, and workgroups that
        // walk them in lock step would otherwise camp on the same HBM channels.
        const int nq = (kz - ka + kDecUnroll - 1) / kDecUnroll;   // groups of 4 keys
        const int rot = nq > 0 ? (int)(((uint32_t)block_idx_x() * 2654435761u) >> 8) % nq : 0;
        for (int g = 0; g < nq; ++g) {
            const int gq = g + rot < nq ? g + rot : g + rot - nq;
            const int j0 = ka + gq * kDecUnroll;
            u32x4 kr[kDecUnroll], vr[kDecUnroll];
            bool vis[kDecUnroll];
            for (int u = 0; u < kDecUnroll; ++u) {
                const int j = j0 + u < kz ? j0 + u : kz - 1;
                kr[u] = global_load_b128(kb + (int64_t)j * p.k_ss);
                vr[u] = global_load_b128(vb + (int64_t)j * p.v_ss);
                vis[u] = (j0 + u < kz) && (!mrow || mrow[j] != 0);
            }
            float s[kDecUnroll];
            float mx = -INFINITY;
            for (int u = 0; u < kDecUnroll; ++u) {
                float kf[8];
                unpack_bf16x8(kr[u], kf);
                float a = 0.0f;
                for (int j = 0; j < 8; ++j) a = fmaf(qf[j], kf[j], a);
                a += shfl_xor_f(a, 1);
                a += shfl_xor_f(a, 2);
                a += shfl_xor_f(a, 4);
                a += shfl_xor_f(a, 8);
                s[u] = vis[u] ? a : -INFINITY;
                mx = fmaxf(mx, s[u]);
            }
            const float m_new = fmaxf(m, mx);
            const float m_safe = m_new == -INFINITY ? 0.0f : m_new;
            const float alpha = fast_exp2(m - m_safe);
            l *= alpha;
            for (int j = 0; j < 8; ++j) o[j] *= alpha;
            for (int u = 0; u < kDecUnroll; ++u) {
                const float pu = fast_exp2(s[u] - m_safe);
                l += pu;
                float vf[8];
                unpack_bf16x8(vr[u], vf);
                for (int j = 0; j < 8; ++j) o[j] = fmaf(pu, vf[j], o[j]);
            }
            m = m_new;
        }
        if (h_ok) {
            const float inv = l > 0.0f ? 1.0f / l : 0.0f;
            float* op = p.out_acc + (((int64_t)split * p.B + b) * p.H + h) * kHeadDim + li * 8;
            f32x4 w0 = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
            f32x4 w1 = {o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv};
            global_store_f32x4(op, w0);
            global_store_f32x4(op + 4, w1);
            if (li == 0)   // m, l are in log2 units: lse = (m + log2 l) * ln 2
                p.lse_acc[((int64_t)split * p.B + b) * p.H + h] =
                    l > 0.0f ? (m + fast_log2(l)) * kLn2 : -INFINITY;
        }
    }
}

}  // namespace lwm
