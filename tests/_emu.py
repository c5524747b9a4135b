"""Test helper: the HIP kernel headers compiled for the HOST (tests/emu/) and
driven through the same C ABI with numpy buffers.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from lwm_amd import _capi
from oracle.attention_ref import from_bf16_bits, to_bf16_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "emu", "liblwm_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
_lib = None


def _sources():
    out = []
    for d in ("tests/emu", "lwm_amd/csrc", "include"):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".h", ".inc", ".cpp")):
                out.append(os.path.join(ROOT, d, f))
    return out


def build():
    if os.path.exists(EMU_SO):
        t = os.path.getmtime(EMU_SO)
        if all(os.path.getmtime(s) <= t for s in _sources()):
            return EMU_SO
    # LWM_EMU_CFLAGS: extra -D switches, to run the emulated tests against a kernel variant
    cmd = [CLANG, "-O2", "-std=c++17", "-mavx2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC",
           *os.environ.get("LWM_EMU_CFLAGS", "").split(), "-I", "tests", "-I", "lwm_amd/csrc",
           "-I", "include", "tests/emu/lwm_emu.cpp", "-o", EMU_SO, "-lpthread"]
    subprocess.run(cmd, cwd=ROOT, check=True)
    return EMU_SO


def lib():
    global _lib
    if _lib is None:
        _lib = _capi.bind(C.CDLL(build()))
    return _lib


def _t4(arr):
    """numpy uint16 (bf16 bits) array (B,S,H,D) -> LwmTensor4."""
    assert arr.dtype == np.uint16 and arr.ndim == 4 and arr.strides[-1] == 2
    sb, ss, sh, _ = (s // 2 for s in arr.strides)
    return _capi.LwmTensor4(arr.ctypes.data, sb, ss, sh)


def _ptr(a):
    return None if a is None else a.ctypes.data


def aligned(shape, dtype, align=16):
    """`align`-byte aligned zero array."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros(n + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


def bf16_array(x):
    a = aligned(x.shape, np.uint16)
    a[...] = to_bf16_bits(x)
    return a


def base_args(q, k, v, *, causal, q_start, k_start, seg_q, seg_k, key_valid, scale, q_piece2=None, k_piece2=None):
    """q_piece2 / k_piece2 = (split row, position of that row) -- or a list of such cuts: the extra pieces of
    LwmAttnArgs' piecewise position maps"""
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    a = _capi.LwmAttnArgs()
    a.q, a.k, a.v = _t4(q), _t4(k), _t4(v)
    a.B, a.H, a.Sq, a.Sk, a.D = B, H, Sq, Sk, D
    a.q_start, a.k_start = q_start, k_start
    cuts = lambda c: [c] if isinstance(c, tuple) else list(c)
    if q_piece2 is not None:
        _capi.set_pieces(a, "q", [(0, q_start)] + cuts(q_piece2))
    if k_piece2 is not None:
        _capi.set_pieces(a, "k", [(0, k_start)] + cuts(k_piece2))
    a.scale = scale if scale is not None else 1.0 / np.sqrt(D)
    a.causal = int(causal)
    keep = []
    if seg_q is not None:
        sq = np.ascontiguousarray(seg_q, dtype=np.int32)
        sk = np.ascontiguousarray(seg_k, dtype=np.int32)
        a.segment_ids_q, a.segment_ids_k = sq.ctypes.data, sk.ctypes.data
        keep += [sq, sk]
    kv = None
    if key_valid is not None:
        kv = np.ascontiguousarray(key_valid, dtype=np.uint8)
        a.key_valid = kv.ctypes.data
        keep.append(kv)
    if seg_q is not None and SEGMENT_SKIP:
        L = lib()
        bq = aligned((B, (Sq + 31) // 32, 2), np.int32)
        bk = aligned((B, (Sk + 31) // 32, 2), np.int32)
        _capi.check(L, L.lwm_attn_segment_blocks(sq.ctypes.data, None, bq.ctypes.data, B, Sq, None), "seg_blocks")
        _capi.check(L, L.lwm_attn_segment_blocks(sk.ctypes.data, _ptr(kv), bk.ctypes.data, B, Sk, None), "seg_blocks")
        a.seg_blocks_q, a.seg_blocks_k = bq.ctypes.data, bk.ctypes.data
        keep += [bq, bk]
    return a, keep


SEGMENT_SKIP = True


def attn_fwd(q, k, v, *, causal=True, q_start=0, k_start=0, seg_q=None, seg_k=None,
             key_valid=None, scale=None, carry=None, final=True, q_piece2=None, k_piece2=None):
    """q,k,v: float arrays (rounded to bf16 here).  Returns (out f32, lse f32) or the
    updated carry (out_acc, lse_acc) when final=False."""
    L = lib()
    qb, kb, vb = bf16_array(q), bf16_array(k), bf16_array(v)
    B, Sq, H, D = q.shape
    a, keep = base_args(qb, kb, vb, causal=causal, q_start=q_start, k_start=k_start, seg_q=seg_q,
                        seg_k=seg_k, key_valid=key_valid, scale=scale, q_piece2=q_piece2, k_piece2=k_piece2)
    out = aligned((B, Sq, H, D), np.uint16)
    lse = aligned((B, H, Sq), np.float32)
    if carry is not None:
        out_acc, lse_acc = carry
        a.carry_in = 1
    else:
        out_acc, lse_acc = aligned((B, Sq, H, D), np.float32), aligned((B, H, Sq), np.float32)
    a.out = _t4(out)
    a.lse = lse.ctypes.data
    a.out_acc, a.lse_acc = out_acc.ctypes.data, lse_acc.ctypes.data
    a.final_out = int(final)
    _capi.check(L, L.lwm_attn_fwd(C.byref(a), None), "lwm_attn_fwd")
    if final:
        return from_bf16_bits(out), lse
    return out_acc, lse_acc


def attn_bwd(q, k, v, out, lse, dout, *, causal=True, q_start=0, k_start=0, seg_q=None,
             seg_k=None, key_valid=None, scale=None, carry=None, final=True, q_piece2=None, k_piece2=None):
    """Returns (dq, dk, dv) as f32 (bf16-rounded when final): lwm_attn_bwd_delta + lwm_attn_bwd_dkdv + lwm_attn_bwd_dq."""
    L = lib()
    qb, kb, vb = bf16_array(q), bf16_array(k), bf16_array(v)
    ob, dob = bf16_array(out), bf16_array(dout)
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    a, keep = base_args(qb, kb, vb, causal=causal, q_start=q_start, k_start=k_start, seg_q=seg_q,
                        seg_k=seg_k, key_valid=key_valid, scale=scale, q_piece2=q_piece2, k_piece2=k_piece2)
    lse_a = aligned((B, H, Sq), np.float32)
    lse_a[...] = lse
    delta = aligned((L.lwm_attn_bwd_delta_bytes(B, H, Sq) // 4,), np.float32)
    delta[...] = np.nan      # the call must write everything the kernels read
    dq, dk, dv = (aligned((B, Sq, H, D), np.uint16), aligned((B, Sk, H, D), np.uint16),
                  aligned((B, Sk, H, D), np.uint16))
    if carry is not None:
        dq_acc, dk_acc, dv_acc = carry
        a.carry_in = 1
    else:
        dq_acc = aligned((B, Sq, H, D), np.float32)
        dk_acc = aligned((B, Sk, H, D), np.float32)
        dv_acc = aligned((B, Sk, H, D), np.float32)
    a.out, a.dout = _t4(ob), _t4(dob)
    a.dq, a.dk, a.dv = _t4(dq), _t4(dk), _t4(dv)
    a.lse, a.delta = lse_a.ctypes.data, delta.ctypes.data
    a.delta_bytes = delta.nbytes
    a.dq_acc, a.dk_acc, a.dv_acc = dq_acc.ctypes.data, dk_acc.ctypes.data, dv_acc.ctypes.data
    a.final_out = int(final)
    _capi.check(L, L.lwm_attn_bwd_delta(C.byref(a), None), "lwm_attn_bwd_delta")
    _capi.check(L, L.lwm_attn_bwd_dkdv(C.byref(a), None), "lwm_attn_bwd_dkdv")
    _capi.check(L, L.lwm_attn_bwd_dq(C.byref(a), None), "lwm_attn_bwd_dq")
    if final:
        return from_bf16_bits(dq), from_bf16_bits(dk), from_bf16_bits(dv)
    return dq_acc, dk_acc, dv_acc


# ---------------------------------------------------------------- the float32 flavour (lwm_attn_*_f32)
def _t4f(arr):
    """numpy float32 array (B,S,H,D) -> LwmTensor4."""
    assert arr.dtype == np.float32 and arr.ndim == 4 and arr.strides[-1] == 4
    sb, ss, sh, _ = (s // 4 for s in arr.strides)
    return _capi.LwmTensor4(arr.ctypes.data, sb, ss, sh)


def f32_array(x):
    a = aligned(np.shape(x), np.float32)
    a[...] = x
    return a


def _base_args_f32(qa, ka, va, *, causal, q_start, k_start, seg_q, seg_k, key_valid, scale):
    B, Sq, H, D = qa.shape
    Sk = ka.shape[1]
    a = _capi.LwmAttnArgs()
    a.q, a.k, a.v = _t4f(qa), _t4f(ka), _t4f(va)
    a.B, a.H, a.Sq, a.Sk, a.D = B, H, Sq, Sk, D
    a.q_start, a.k_start = q_start, k_start
    a.scale = scale if scale is not None else 1.0 / np.sqrt(D)
    a.causal = int(causal)
    keep = []
    if seg_q is not None:
        sq = np.ascontiguousarray(seg_q, dtype=np.int32)
        sk = np.ascontiguousarray(seg_k, dtype=np.int32)
        a.segment_ids_q, a.segment_ids_k = sq.ctypes.data, sk.ctypes.data
        keep += [sq, sk]
    if key_valid is not None:
        kv = np.ascontiguousarray(key_valid, dtype=np.uint8)
        a.key_valid = kv.ctypes.data
        keep.append(kv)
    return a, keep


def attn_fwd_f32(q, k, v, *, causal=True, q_start=0, k_start=0, seg_q=None, seg_k=None, key_valid=None, scale=None,
                 carry=None, final=True):
    """float32 operands, nothing rounded.  Returns (out, lse) or the updated carry (out_acc, lse_acc) when final=False."""
    L = lib()
    qa, ka, va = f32_array(q), f32_array(k), f32_array(v)
    B, Sq, H, D = qa.shape
    a, keep = _base_args_f32(qa, ka, va, causal=causal, q_start=q_start, k_start=k_start, seg_q=seg_q, seg_k=seg_k,
                             key_valid=key_valid, scale=scale)
    out = aligned((B, Sq, H, D), np.float32)
    out[...] = np.nan
    lse = aligned((B, H, Sq), np.float32)
    lse[...] = np.nan
    if carry is not None:
        out_acc, lse_acc = carry
        a.carry_in = 1
    else:
        out_acc, lse_acc = aligned((B, Sq, H, D), np.float32), aligned((B, H, Sq), np.float32)
    a.out = _t4f(out)
    a.lse = lse.ctypes.data
    a.out_acc, a.lse_acc = out_acc.ctypes.data, lse_acc.ctypes.data
    a.final_out = int(final)
    _capi.check(L, L.lwm_attn_fwd_f32(C.byref(a), None), "lwm_attn_fwd_f32")
    return (out, lse) if final else (out_acc, lse_acc)


def attn_bwd_f32(q, k, v, out, lse, dout, *, causal=True, q_start=0, k_start=0, seg_q=None, seg_k=None, key_valid=None,
                 scale=None, carry=None, final=True):
    """(dq, dk, dv) f32: lwm_attn_bwd_delta_f32 + lwm_attn_bwd_dkdv_f32 + lwm_attn_bwd_dq_f32."""
    L = lib()
    qa, ka, va, oa, doa = (f32_array(t) for t in (q, k, v, out, dout))
    B, Sq, H, D = qa.shape
    Sk = ka.shape[1]
    a, keep = _base_args_f32(qa, ka, va, causal=causal, q_start=q_start, k_start=k_start, seg_q=seg_q, seg_k=seg_k,
                             key_valid=key_valid, scale=scale)
    lse_a = f32_array(lse)
    delta = aligned((L.lwm_attn_bwd_delta_bytes(B, H, Sq) // 4,), np.float32)
    delta[...] = np.nan
    dq, dk, dv = (aligned((B, Sq, H, D), np.float32), aligned((B, Sk, H, D), np.float32), aligned((B, Sk, H, D), np.float32))
    for t in (dq, dk, dv):
        t[...] = np.nan
    if carry is not None:
        dq_acc, dk_acc, dv_acc = carry
        a.carry_in = 1
    else:
        dq_acc, dk_acc, dv_acc = (aligned((B, Sq, H, D), np.float32), aligned((B, Sk, H, D), np.float32),
                                  aligned((B, Sk, H, D), np.float32))
    a.out, a.dout = _t4f(oa), _t4f(doa)
    a.dq, a.dk, a.dv = _t4f(dq), _t4f(dk), _t4f(dv)
    a.lse, a.delta = lse_a.ctypes.data, delta.ctypes.data
    a.delta_bytes = delta.nbytes
    a.dq_acc, a.dk_acc, a.dv_acc = dq_acc.ctypes.data, dk_acc.ctypes.data, dv_acc.ctypes.data
    a.final_out = int(final)
    _capi.check(L, L.lwm_attn_bwd_delta_f32(C.byref(a), None), "lwm_attn_bwd_delta_f32")
    _capi.check(L, L.lwm_attn_bwd_dkdv_f32(C.byref(a), None), "lwm_attn_bwd_dkdv_f32")
    _capi.check(L, L.lwm_attn_bwd_dq_f32(C.byref(a), None), "lwm_attn_bwd_dq_f32")
    return (dq, dk, dv) if final else (dq_acc, dk_acc, dv_acc)


# ---------------------------------------------------------------- VQGAN primitives
def _af32(x):
    a = aligned(np.shape(x), np.float32)
    a[...] = x
    return a


def conv2d(x, w, bias=None, residual=None, *, stride=1, pad=None, up_shift=0, out_hw=None,
           clip=False):
    L = lib()
    x, w = _af32(x), _af32(w)
    B, Hin, Win, Cin = x.shape
    KH, KW, _, Cout = w.shape
    if pad is None:
        pad = (KH - 1) // 2
    Hv, Wv = Hin << up_shift, Win << up_shift
    if out_hw is None:
        out_hw = ((Hv + 2 * pad - KH) // stride + 1, (Wv + 2 * pad - KW) // stride + 1)
    Ho, Wo = out_hw
    y = aligned((B, Ho, Wo, Cout), np.float32)
    b = None if bias is None else _af32(bias)
    r = None if residual is None else _af32(residual)
    a = _capi.LwmConvArgs(x.ctypes.data, w.ctypes.data, _ptr(b), _ptr(r), y.ctypes.data, B, Hin, Win,
                          Cin, Cout, KH, KW, stride, pad, up_shift, Ho, Wo, int(clip))
    _capi.check(L, L.lwm_conv2d_nhwc_f32(C.byref(a), None), "lwm_conv2d_nhwc_f32")
    return y


def groupnorm(x, gamma, beta, *, groups=32, eps=1e-6, silu=False):
    L = lib()
    x = _af32(x)
    B, Cc = x.shape[0], x.shape[-1]
    HW = int(np.prod(x.shape[1:-1]))
    y = aligned(x.shape, np.float32)
    g, bt = _af32(gamma), _af32(beta)
    ws = aligned((max(L.lwm_groupnorm_workspace_bytes(B, HW, Cc, groups), 16) // 8,), np.float64)
    _capi.check(L, L.lwm_groupnorm_silu_f32(x.ctypes.data, g.ctypes.data, bt.ctypes.data, y.ctypes.data,
                                            ws.ctypes.data, B, HW, Cc, groups, eps, int(silu), None),
                "lwm_groupnorm_silu_f32")
    return y


def vq_argmin(z, codebook):
    L = lib()
    z, cb = _af32(z), _af32(codebook)
    E, D = cb.shape
    N = z.size // D
    se = aligned((E,), np.float32)
    _capi.check(L, L.lwm_vq_sqnorm_f32(cb.ctypes.data, se.ctypes.data, E, D, None), "lwm_vq_sqnorm_f32")
    idx = aligned(z.shape[:-1], np.int32)
    _capi.check(L, L.lwm_vq_argmin_f32(z.ctypes.data, cb.ctypes.data, se.ctypes.data, idx.ctypes.data,
                                       N, E, D, None), "lwm_vq_argmin_f32")
    return idx


def vq_gather(codebook, idx, z=None):
    L = lib()
    cb = _af32(codebook)
    E, D = cb.shape
    ia = aligned(np.shape(idx), np.int32)
    ia[...] = idx
    out = aligned(ia.shape + (D,), np.float32)
    zz = None if z is None else _af32(z)
    _capi.check(L, L.lwm_vq_gather_f32(cb.ctypes.data, ia.ctypes.data, _ptr(zz), out.ctypes.data,
                                       ia.size, E, D, None), "lwm_vq_gather_f32")
    return out


# ---------------------------------------------------------------- inference (dense mask, split-K)
def attn_infer(q, k, v, mask, *, k_splits=1, scale=None):
    """Split-K forward with a dense u8 mask (B,Sq,Sk) + combine; returns (out f32 from bf16, lse)."""
    L = lib()
    qb, kb, vb = bf16_array(q), bf16_array(k), bf16_array(v)
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    a, keep = base_args(qb, kb, vb, causal=False, q_start=0, k_start=0, seg_q=None, seg_k=None,
                        key_valid=None, scale=scale)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    a.dense_mask, a.mask_stride_b, a.mask_stride_q = m.ctypes.data, m.strides[0], m.strides[1]
    o_parts = aligned((k_splits, B, Sq, H, D), np.float32)
    l_parts = aligned((k_splits, B, H, Sq), np.float32)
    a.out_acc, a.lse_acc = o_parts.ctypes.data, l_parts.ctypes.data
    a.k_splits = k_splits
    _capi.check(L, L.lwm_attn_fwd(C.byref(a), None), "lwm_attn_fwd")
    out = aligned((B, Sq, H, D), np.uint16)
    lse = aligned((B, H, Sq), np.float32)
    _capi.check(L, L.lwm_attn_combine(o_parts.ctypes.data, l_parts.ctypes.data, k_splits, _t4(out), None,
                                      lse.ctypes.data, B, Sq, H, D, None), "lwm_attn_combine")
    return from_bf16_bits(out), lse


def kv_cache_write(cache_bits, src_bits, dst_row0, src_row0, nrows):
    L = lib()
    B, S, H, D = cache_bits.shape
    _capi.check(L, L.lwm_kv_cache_write(cache_bits.ctypes.data, src_bits.ctypes.data, B,
                                        cache_bits.strides[0] // 2, src_bits.strides[0] // 2, dst_row0,
                                        src_row0, nrows, H * D, None), "lwm_kv_cache_write")
    return cache_bits


# ---------------------------------------------------------------- RoPE / RMSNorm
def rope(x, table, pos, conj=False):
    L = lib()
    xb = bf16_array(x)
    B, S, H, D = x.shape
    y = aligned((B, S, H, D), np.uint16)
    tab = aligned(table.shape, np.float32)
    tab[...] = table
    ps = np.ascontiguousarray(pos, dtype=np.int32)
    _capi.check(L, L.lwm_rope_bf16(_t4(xb), _t4(y), tab.ctypes.data, ps.ctypes.data, B, S, H, D, table.shape[0],
                                   int(conj), None), "lwm_rope_bf16")
    return from_bf16_bits(y)


def rmsnorm_fwd(x, w, eps=1e-6):
    L = lib()
    xb, wb = bf16_array(x), bf16_array(w)
    Cc = x.shape[-1]
    rows = x.size // Cc
    y = aligned(x.shape, np.uint16)
    rstd = aligned((rows,), np.float32)
    _capi.check(L, L.lwm_rmsnorm_fwd_bf16(xb.ctypes.data, wb.ctypes.data, y.ctypes.data, rstd.ctypes.data, rows, Cc,
                                          eps, None), "lwm_rmsnorm_fwd_bf16")
    return from_bf16_bits(y), rstd


def rmsnorm_bwd(x, w, g, rstd):
    L = lib()
    xb, wb, gb = bf16_array(x), bf16_array(w), bf16_array(g)
    Cc = x.shape[-1]
    rows = x.size // Cc
    dx = aligned(x.shape, np.uint16)
    dw = aligned((Cc,), np.uint16)
    ws = aligned((max(L.lwm_rmsnorm_bwd_workspace_bytes(rows, Cc), 16) // 4,), np.float32)
    r = aligned((rows,), np.float32)
    r[...] = rstd
    _capi.check(L, L.lwm_rmsnorm_bwd_bf16(xb.ctypes.data, wb.ctypes.data, gb.ctypes.data, r.ctypes.data,
                                          dx.ctypes.data, dw.ctypes.data, ws.ctypes.data, rows, Cc, None),
                "lwm_rmsnorm_bwd_bf16")
    return from_bf16_bits(dx), from_bf16_bits(dw)


def softmax_ce(logits, target, weight=None, want_grad=True):
    L = lib()
    lb = bf16_array(logits)
    rows, V = logits.shape
    tg = np.ascontiguousarray(target, dtype=np.int32)
    w = None if weight is None else _af32(weight)
    nll = aligned((rows,), np.float32)
    cor = aligned((rows,), np.int32)
    dl = aligned((rows, V), np.uint16) if want_grad else None
    _capi.check(L, L.lwm_softmax_ce_bf16(lb.ctypes.data, tg.ctypes.data, _ptr(w), nll.ctypes.data, cor.ctypes.data,
                                         _ptr(dl), rows, V, None), "lwm_softmax_ce_bf16")
    return nll, cor, (None if dl is None else from_bf16_bits(dl))


def kv_cache_write_at(cache_bits, src_bits, index, row_offset, src_row0, nrows):
    L = lib()
    B, S, H, D = cache_bits.shape
    idx = aligned((4,), np.int32)
    idx[0] = index
    _capi.check(L, L.lwm_kv_cache_write_at(cache_bits.ctypes.data, src_bits.ctypes.data, B,
                                           cache_bits.strides[0] // 2, src_bits.strides[0] // 2,
                                           idx.ctypes.data, row_offset, S, src_row0, nrows, H * D, None),
                "lwm_kv_cache_write_at")
    return cache_bits


def gemv(x, w, want_f32=False):
    """x (rows, K), w (K, N) float32 arrays holding bf16 values -> (rows, N) through lwm_gemv_bf16."""
    L = lib()
    xb, wb = bf16_array(x), bf16_array(w)
    rows, K = x.shape
    N = w.shape[1]
    ws = aligned((max(L.lwm_gemv_workspace_bytes(rows, K, N), 16) // 4,), np.float32)
    y = aligned((rows, N), np.uint16)
    yf = aligned((rows, N), np.float32)
    _capi.check(L, L.lwm_gemv_bf16(xb.ctypes.data, K, wb.ctypes.data, y.ctypes.data, N, yf.ctypes.data if want_f32 else None,
                                   ws.ctypes.data, rows, K, N, None), "lwm_gemv_bf16")
    return (from_bf16_bits(y), yf.copy()) if want_f32 else from_bf16_bits(y)


def gemv_multi(x, ws, want_f32=False):
    """1..3 kernels that share x through ONE lwm_gemv_multi_bf16 call."""
    import ctypes as C
    L = lib()
    xb = bf16_array(x)
    wbs = [bf16_array(w) for w in ws]
    rows, K = x.shape
    Ns = [w.shape[1] for w in ws]
    n = len(ws)
    wsz = sum(max(L.lwm_gemv_workspace_bytes(rows, K, N), 16) for N in Ns)
    work = aligned((wsz // 4,), np.float32)
    ys = [aligned((rows, N), np.float32 if want_f32 else np.uint16) for N in Ns]
    vp = C.c_void_p * n
    y_arr = vp(*[y.ctypes.data for y in ys])
    _capi.check(L, L.lwm_gemv_multi_bf16(xb.ctypes.data, K, n, vp(*[w.ctypes.data for w in wbs]), None if want_f32 else y_arr,
                                         (C.c_int64 * n)(*Ns), y_arr if want_f32 else None, (C.c_int32 * n)(*Ns),
                                         work.ctypes.data, rows, K, None), "lwm_gemv_multi_bf16")
    return [y.copy() if want_f32 else from_bf16_bits(y) for y in ys]


def gemv_fused(x, ws, *, norm=None, residual=None, want_ss=False, want_f32=False):
    """lwm_gemv_fused_bf16: 1..3 kernels that share x; norm = (ss_in (rows, n) f32, weight (K,), eps) normalises x on
    load; residual (rows, N) is added in the reduction (one kernel); want_ss returns the (rows, N/128) partial sums of
    squares of the output."""
    import ctypes as C
    L = lib()
    xb = bf16_array(x)
    wbs = [bf16_array(w) for w in ws]
    rows, K = x.shape
    Ns = [w.shape[1] for w in ws]
    n = len(ws)
    wsz = sum(max(L.lwm_gemv_workspace_bytes(rows, K, N), 16) for N in Ns)
    work = aligned((wsz // 4,), np.float32)
    ys = [aligned((rows, N), np.float32 if want_f32 else np.uint16) for N in Ns]
    a = _capi.LwmGemvArgs()
    a.x, a.ldx, a.nmat, a.rows, a.K = xb.ctypes.data, K, n, rows, K
    a.workspace = work.ctypes.data
    for i in range(n):
        a.w[i], a.N[i] = wbs[i].ctypes.data, Ns[i]
        if want_f32:
            a.y_f32[i] = ys[i].ctypes.data
        else:
            a.y[i], a.ldy[i] = ys[i].ctypes.data, Ns[i]
    keep = []
    if norm is not None:
        ss, w, eps = norm
        ssa = aligned(ss.shape, np.float32)
        ssa[...] = ss
        wb = bf16_array(w)
        keep += [ssa, wb]
        a.norm_weight, a.ss_in, a.ss_n, a.eps = wb.ctypes.data, ssa.ctypes.data, ss.shape[1], eps
    if residual is not None:
        rb = bf16_array(residual)
        keep.append(rb)
        a.residual[0], a.ldres[0] = rb.ctypes.data, Ns[0]
    sso = None
    if want_ss:
        sso = aligned((rows, Ns[0] // 128), np.float32)
        a.ss_out = sso.ctypes.data
    _capi.check(L, L.lwm_gemv_fused_bf16(C.byref(a), None), "lwm_gemv_fused_bf16")
    out = [y.copy() if want_f32 else from_bf16_bits(y) for y in ys]
    return (out, sso.copy()) if want_ss else out
