"""Thin torch-tensor front end of the C ABI (include/lwm_hip.h).

Every function here enqueues hand-written HIP kernels from liblwm_hip.so on the
current torch stream.  There is no PyTorch / CPU fallback: tensors must be
bf16/f32 on a ROCm device and the shared library must be present.
"""
import ctypes as C
import os
import math

import torch

from . import _capi
from ._lib import lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(stream=None):
    """The HIP stream the launches go to: torch's current stream of the current device (the raw-handle query is
    ~10x cheaper than building a torch.cuda.Stream object -- a decode step asks some 50 times per token)."""
    if stream is not None:
        return C.c_void_p(stream.cuda_stream)
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _t4(t, name, dtype=None):
    """(B,S,H,D) bf16 -- or f32, the fp32 flavour of the op -- device tensor -> LwmTensor4; `dtype`: the one it must have"""
    if t is None:
        return _capi.LwmTensor4(None, 0, 0, 0)
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a ROCm device tensor (lwm_amd has no CPU path)")
    if t.dtype not in ((torch.bfloat16, torch.float32) if dtype is None else (dtype,)) or t.dim() != 4 or t.stride(3) != 1:
        raise ValueError(f"{name}: expected {'bf16 / f32' if dtype is None else dtype} (B,S,H,D) with contiguous D, got "
                         f"{t.dtype} {tuple(t.shape)} strides {t.stride()}")
    return _capi.LwmTensor4(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def _entry(L, name, dtype):
    """The C entry point of an attention launch for operands of `dtype`: bf16 = the headline kernels, f32 = the
    `--dtype=fp32` flavour on the exact-f32 matrix instruction (lwm_attn_*_f32, csrc/attn_f32.h)."""
    return getattr(L, name + "_f32") if dtype == torch.float32 else getattr(L, name)


def _f32(t, name, shape=None):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous f32 device tensor")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.data_ptr()


def _base(q, k, v, *, q_start, k_start, causal, seg_q, seg_k, key_valid, scale, q_piece2=None, k_piece2=None):
    """q_piece2 / k_piece2 = (cut row, position of that row) -- or a list of such cuts, in ascending order: the pieces of
    LwmAttnArgs' piecewise position maps beyond the first (rows before the first cut sit at *_start + row, rows from a cut
    on at its position + (row - cut); cuts are multiples of 256 rows)"""
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    a = _capi.LwmAttnArgs()
    a.q, a.k, a.v = _t4(q, "q"), _t4(k, "k", q.dtype), _t4(v, "v", q.dtype)
    a.B, a.H, a.Sq, a.Sk, a.D = B, H, Sq, Sk, D
    a.q_start, a.k_start = int(q_start), int(k_start)
    cuts = lambda c: [c] if isinstance(c, tuple) else list(c)
    if q_piece2 is not None:
        _capi.set_pieces(a, "q", [(0, q_start)] + cuts(q_piece2))
    if k_piece2 is not None:
        _capi.set_pieces(a, "k", [(0, k_start)] + cuts(k_piece2))
    a.scale = float(scale) if scale is not None else 1.0 / math.sqrt(D)
    a.causal = int(bool(causal))
    if (seg_q is None) != (seg_k is None):
        raise ValueError("seg_q and seg_k must be given together")
    if seg_q is not None:
        for n, s, L in (("seg_q", seg_q, Sq), ("seg_k", seg_k, Sk)):
            if s.dtype != torch.int32 or not s.is_contiguous() or tuple(s.shape) != (B, L) or not s.is_cuda:
                raise ValueError(f"{n}: expected contiguous int32 device tensor of shape {(B, L)}")
        a.segment_ids_q, a.segment_ids_k = seg_q.data_ptr(), seg_k.data_ptr()
    if key_valid is not None:
        if key_valid.dtype != torch.uint8 or not key_valid.is_contiguous() or \
                tuple(key_valid.shape) != (B, Sk) or not key_valid.is_cuda:
            raise ValueError(f"key_valid: expected contiguous uint8 device tensor of shape {(B, Sk)}")
        a.key_valid = key_valid.data_ptr()
    if seg_q is not None and SEGMENT_SKIP and q.dtype == torch.bfloat16:
        # block-sparsity hints: whole documents of a packed batch are skipped in-kernel (the f32 kernels do not read them)
        bq, bk = _cached_segment_blocks(seg_q, None), _cached_segment_blocks(seg_k, key_valid)
        a.seg_blocks_q, a.seg_blocks_k = bq.data_ptr(), bk.data_ptr()
        a._keep = (bq, bk)
    return a


def _cached_segment_blocks(seg, valid):
    """The hint table of a segment-id tensor is computed once and kept ON that tensor object (the ring
    driver hands the same slice objects to the forward, dQ and dK/dV launches of a block pair:
    lwm_amd/ring.py::_MaskSlices); an in-place edit (tensor._version) or another key_valid recomputes."""
    key = (seg._version, None if valid is None else (id(valid), valid._version))
    hit = getattr(seg, "_lwm_seg_blocks", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    blocks = segment_blocks(seg, valid)
    seg._lwm_seg_blocks = (key, blocks, valid)      # `valid` kept alive: its id is part of the key
    return blocks


SEGMENT_SKIP = True   # set False to A/B the in-kernel document skipping


def segment_blocks(seg, valid=None):
    """(min, max) segment id per block of 32 rows -> int32 (B, ceil(S/32), 2) (lwm_attn_segment_blocks)."""
    B, S = seg.shape
    out = torch.empty((B, (S + 31) // 32, 2), dtype=torch.int32, device=seg.device)
    L = lib()
    _capi.check(L, L.lwm_attn_segment_blocks(seg.data_ptr(), None if valid is None else valid.data_ptr(),
                                             out.data_ptr(), B, S, _stream_ptr()), "lwm_attn_segment_blocks")
    return out


def attn_fwd_block(q, k, v, *, q_start=0, k_start=0, causal=True, seg_q=None, seg_k=None,
                   key_valid=None, scale=None, out=None, lse=None, out_acc=None, lse_acc=None,
                   carry_in=False, final=True, dense_mask=None, q_piece2=None, k_piece2=None):
    """One ring step of the forward (lwm_attn_fwd).  Returns (out, lse) when
    `final`, else the updated (out_acc, lse_acc).  dense_mask: optional u8
    (B,Sq,Sk) view (last dim contiguous) ANDed with the other masks."""
    B, Sq, H, D = q.shape
    a = _base(q, k, v, q_start=q_start, k_start=k_start, causal=causal, seg_q=seg_q, seg_k=seg_k,
              key_valid=key_valid, scale=scale, q_piece2=q_piece2, k_piece2=k_piece2)
    _set_dense_mask(a, dense_mask, B, Sq, k.shape[1])
    if final:
        if out is None:
            out = torch.empty((B, Sq, H, D), dtype=q.dtype, device=q.device)
        if lse is None:
            lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
        a.out = _t4(out, "out", q.dtype)
        a.lse = _f32(lse, "lse", (B, H, Sq))
    else:
        if out_acc is None:
            out_acc = torch.empty((B, Sq, H, D), dtype=torch.float32, device=q.device)
        if lse_acc is None:
            lse_acc = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    a.out_acc = _f32(out_acc, "out_acc", (B, Sq, H, D))
    a.lse_acc = _f32(lse_acc, "lse_acc", (B, H, Sq))
    a.carry_in = int(bool(carry_in))
    a.final_out = int(bool(final))
    L = lib()
    _capi.check(L, _entry(L, "lwm_attn_fwd", q.dtype)(C.byref(a), _stream_ptr()), "lwm_attn_fwd")
    return (out, lse) if final else (out_acc, lse_acc)


def _set_dense_mask(a, dense_mask, B, Sq, Sk):
    if dense_mask is None:
        return
    m = dense_mask
    if not m.is_cuda or m.dtype != torch.uint8 or tuple(m.shape) != (B, Sq, Sk) or m.stride(2) != 1:
        raise ValueError(f"dense_mask: expected a u8 device tensor of shape {(B, Sq, Sk)} with contiguous keys")
    a.dense_mask, a.mask_stride_b, a.mask_stride_q = m.data_ptr(), m.stride(0), m.stride(1)


def attn_fwd_splitk(q, k, v, *, k_splits, q_start=0, k_start=0, causal=False, seg_q=None, seg_k=None,
                    key_valid=None, dense_mask=None, scale=None):
    """Split-K forward for short query blocks (decode): returns normalised partials
    (o_parts f32 [k_splits,B,Sq,H,D], lse_parts f32 [k_splits,B,H,Sq]) -- merge with
    attn_combine."""
    B, Sq, H, D = q.shape
    if q.dtype != torch.bfloat16:
        raise ValueError("attn_fwd_splitk: the inference kernels take bf16 operands (the f32 flavour serves the training op)")
    a = _base(q, k, v, q_start=q_start, k_start=k_start, causal=causal, seg_q=seg_q, seg_k=seg_k,
              key_valid=key_valid, scale=scale)
    _set_dense_mask(a, dense_mask, B, Sq, k.shape[1])
    k_splits = max(1, int(k_splits))
    o_parts = torch.empty((k_splits, B, Sq, H, D), dtype=torch.float32, device=q.device)
    lse_parts = torch.empty((k_splits, B, H, Sq), dtype=torch.float32, device=q.device)
    a.out_acc, a.lse_acc = o_parts.data_ptr(), lse_parts.data_ptr()
    a.carry_in, a.final_out, a.k_splits = 0, 0, k_splits
    L = lib()
    _capi.check(L, L.lwm_attn_fwd(C.byref(a), _stream_ptr()), "lwm_attn_fwd")
    return o_parts, lse_parts


def attn_combine(o_parts, lse_parts, *, out=None, out_f32=None, lse=None, want_bf16=True):
    """Merge normalised partials (lwm_attn_combine).  Returns (out bf16 or f32, lse)."""
    P, B, Sq, H, D = o_parts.shape
    if lse is None:
        lse = torch.empty((B, H, Sq), dtype=torch.float32, device=o_parts.device)
    if want_bf16 and out is None:
        out = torch.empty((B, Sq, H, D), dtype=torch.bfloat16, device=o_parts.device)
    if not want_bf16 and out_f32 is None:
        out_f32 = torch.empty((B, Sq, H, D), dtype=torch.float32, device=o_parts.device)
    L = lib()
    _capi.check(L, L.lwm_attn_combine(_f32(o_parts, "o_parts"), _f32(lse_parts, "lse_parts", (P, B, H, Sq)), P,
                                      _t4(out, "out", torch.bfloat16) if want_bf16 else _capi.LwmTensor4(None, 0, 0, 0),
                                      None if want_bf16 else _f32(out_f32, "out_f32"),
                                      _f32(lse, "lse", (B, H, Sq)), B, Sq, H, D, _stream_ptr()),
                "lwm_attn_combine")
    return (out if want_bf16 else out_f32), lse


def kv_cache_write(cache, src, *, dst_row0, src_row0=0, nrows=None):
    """cache[:, dst_row0:dst_row0+nrows] = src[:, src_row0:src_row0+nrows] for (B,S,H,D) bf16 -- or f32: the copy moves
    bytes, a float row is two bf16-sized elements per value -- tensors whose (S,H,D) block is contiguous
    (lwm_kv_cache_write)."""
    for n, t in (("cache", cache), ("src", src)):
        if not t.is_cuda or t.dtype not in (torch.bfloat16, torch.float32) or t.dtype != cache.dtype or t.dim() != 4 or \
                not t[0].is_contiguous():
            raise ValueError(f"{n}: expected bf16 / f32 (B,S,H,D) device tensors of one dtype with contiguous (S,H,D)")
    B, _, H, D = cache.shape
    if nrows is None:
        nrows = src.shape[1] - src_row0
    if dst_row0 < 0 or dst_row0 + nrows > cache.shape[1] or src_row0 + nrows > src.shape[1]:
        raise ValueError("kv_cache_write: row range out of bounds")
    w = cache.element_size() // 2            # bf16-sized units per element
    L = lib()
    _capi.check(L, L.lwm_kv_cache_write(cache.data_ptr(), src.data_ptr(), B, cache.stride(0) * w, src.stride(0) * w,
                                        dst_row0, src_row0, nrows, H * D * w, _stream_ptr()),
                "lwm_kv_cache_write")
    return cache


def kv_cache_write_at(cache, src, index_dev, *, row_offset=0, src_row0=0, nrows=None):
    """cache[:, index + row_offset + i] = src[:, src_row0 + i] with `index` an int32 DEVICE tensor
    (lwm_kv_cache_write_at); rows that fall outside the cache are skipped."""
    for n, t in (("cache", cache), ("src", src)):
        if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 4 or not t[0].is_contiguous():
            raise ValueError(f"{n}: expected bf16 (B,S,H,D) device tensor with contiguous (S,H,D)")
    if not index_dev.is_cuda or index_dev.dtype != torch.int32 or index_dev.numel() != 1:
        raise ValueError("index_dev: expected a one-element int32 device tensor")
    B, rows, H, D = cache.shape
    if nrows is None:
        nrows = src.shape[1] - src_row0
    if src_row0 < 0 or src_row0 + nrows > src.shape[1]:
        raise ValueError("kv_cache_write_at: source row range out of bounds")
    L = lib()
    _capi.check(L, L.lwm_kv_cache_write_at(cache.data_ptr(), src.data_ptr(), B, cache.stride(0), src.stride(0),
                                           index_dev.data_ptr(), row_offset, rows, src_row0, nrows, H * D,
                                           _stream_ptr()), "lwm_kv_cache_write_at")
    return cache


def bwd_stats_shape(B, H, Sq):
    """Shape of the backward's row statistics for a (B, Sq, H, D) query block: per (b, h) the rows
    [-lse * log2 e | -rowsum(dout * out)], padded to a multiple of 64 queries (lwm_attn_bwd_delta_bytes)."""
    return (B, H, 2, (Sq + 63) // 64 * 64)


def attn_bwd_delta(out, dout, lse, delta=None):
    """The row statistics the backward kernels consume (lwm_attn_bwd_delta), from out, dout and the forward's lse."""
    B, Sq, H, D = out.shape
    if delta is None:
        delta = torch.empty(bwd_stats_shape(B, H, Sq), dtype=torch.float32, device=out.device)
    a = _capi.LwmAttnArgs()
    a.out, a.dout = _t4(out, "out"), _t4(dout, "dout", out.dtype)
    a.B, a.H, a.Sq, a.Sk, a.D = B, H, Sq, 0, D
    a.lse = _f32(lse, "lse", (B, H, Sq))
    a.delta = _f32(delta, "delta", bwd_stats_shape(B, H, Sq))
    a.delta_bytes = delta.numel() * 4
    L = lib()
    _capi.check(L, _entry(L, "lwm_attn_bwd_delta", out.dtype)(C.byref(a), _stream_ptr()), "lwm_attn_bwd_delta")
    return delta


def _bwd_base(q, k, v, dout, lse, delta, kw):
    B, Sq, H, D = q.shape
    a = _base(q, k, v, **kw)
    a.dout = _t4(dout, "dout", q.dtype)
    a.lse = _f32(lse, "lse", (B, H, Sq))
    a.delta = _f32(delta, "delta", bwd_stats_shape(B, H, Sq))
    a.delta_bytes = delta.numel() * 4
    return a


def _acc_shape(B, Sq, H, D, head_major):
    """dq accumulator / carry: (B,Sq,H,D), or head-major (B,H,Sq,D)."""
    return (B, H, Sq, D) if head_major else (B, Sq, H, D)


def attn_bwd_dq_block(q, k, v, dout, lse, delta, *, q_start=0, k_start=0, causal=True,
                      seg_q=None, seg_k=None, key_valid=None, scale=None, dq=None, dq_acc=None,
                      carry_in=False, final=True, acc_head_major=False, q_piece2=None, k_piece2=None):
    B, Sq, H, D = q.shape
    a = _bwd_base(q, k, v, dout, lse, delta,
                  dict(q_start=q_start, k_start=k_start, causal=causal, seg_q=seg_q, seg_k=seg_k,
                       key_valid=key_valid, scale=scale, q_piece2=q_piece2, k_piece2=k_piece2))
    if final:
        if dq is None:
            dq = torch.empty((B, Sq, H, D), dtype=q.dtype, device=q.device)
        a.dq = _t4(dq, "dq", q.dtype)
    elif dq_acc is None:
        dq_acc = torch.empty(_acc_shape(B, Sq, H, D, acc_head_major), dtype=torch.float32, device=q.device)
    a.dq_acc = _f32(dq_acc, "dq_acc", _acc_shape(B, Sq, H, D, acc_head_major))
    a.dq_acc_head_major = int(bool(acc_head_major))
    a.carry_in = int(bool(carry_in))
    a.final_out = int(bool(final))
    L = lib()
    _capi.check(L, _entry(L, "lwm_attn_bwd_dq", q.dtype)(C.byref(a), _stream_ptr()), "lwm_attn_bwd_dq")
    return dq if final else dq_acc


def attn_bwd_dkdv_block(q, k, v, dout, lse, delta, *, q_start=0, k_start=0, causal=True,
                        seg_q=None, seg_k=None, key_valid=None, scale=None, dk=None, dv=None,
                        dk_acc=None, dv_acc=None, carry_in=False, final=True, q_piece2=None, k_piece2=None):
    B, Sk, H, D = k.shape
    a = _bwd_base(q, k, v, dout, lse, delta,
                  dict(q_start=q_start, k_start=k_start, causal=causal, seg_q=seg_q, seg_k=seg_k,
                       key_valid=key_valid, scale=scale, q_piece2=q_piece2, k_piece2=k_piece2))
    if final:
        if dk is None:
            dk = torch.empty((B, Sk, H, D), dtype=q.dtype, device=q.device)
        if dv is None:
            dv = torch.empty((B, Sk, H, D), dtype=q.dtype, device=q.device)
        a.dk, a.dv = _t4(dk, "dk", q.dtype), _t4(dv, "dv", q.dtype)
    else:
        if dk_acc is None:
            dk_acc = torch.empty((B, Sk, H, D), dtype=torch.float32, device=q.device)
        if dv_acc is None:
            dv_acc = torch.empty((B, Sk, H, D), dtype=torch.float32, device=q.device)
    a.dk_acc = _f32(dk_acc, "dk_acc", (B, Sk, H, D))
    a.dv_acc = _f32(dv_acc, "dv_acc", (B, Sk, H, D))
    a.carry_in = int(bool(carry_in))
    a.final_out = int(bool(final))
    L = lib()
    _capi.check(L, _entry(L, "lwm_attn_bwd_dkdv", q.dtype)(C.byref(a), _stream_ptr()), "lwm_attn_bwd_dkdv")
    return (dk, dv) if final else (dk_acc, dv_acc)


def cast_f32_to_bf16(src, dst=None):
    if not src.is_cuda or src.dtype != torch.float32 or not src.is_contiguous():
        raise ValueError("cast_f32_to_bf16: expected contiguous f32 device tensor")
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    L = lib()
    _capi.check(L, L.lwm_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(),
                                          _stream_ptr()), "lwm_cast_f32_to_bf16")
    return dst


def sum_f32_to_bf16(srcs, dst=None):
    """bf16(((srcs[0] + srcs[1]) + ...)) -- the owner-side reduction of returned dK/dV partials."""
    srcs = list(srcs)
    for s in srcs:
        if not s.is_cuda or s.dtype != torch.float32 or not s.is_contiguous() or s.shape != srcs[0].shape:
            raise ValueError("sum_f32_to_bf16: expected same-shaped contiguous f32 device tensors")
    if dst is None:
        dst = torch.empty(srcs[0].shape, dtype=torch.bfloat16, device=srcs[0].device)
    elif not dst.is_contiguous() or dst.dtype not in (torch.bfloat16, torch.float32) or dst.numel() != srcs[0].numel():
        raise ValueError("sum_f32_to_bf16: dst must be a contiguous bf16 (or, the fp32 flavour, f32) tensor of the same size")
    ptrs = (C.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
    L = lib()
    if dst.dtype == torch.float32:      # f32 operands: the same ordered sum, nothing rounded (lwm_sum_f32)
        _capi.check(L, L.lwm_sum_f32(ptrs, len(srcs), dst.data_ptr(), srcs[0].numel(), _stream_ptr()), "lwm_sum_f32")
        return dst
    _capi.check(L, L.lwm_sum_f32_to_bf16(ptrs, len(srcs), dst.data_ptr(), srcs[0].numel(), _stream_ptr()),
                "lwm_sum_f32_to_bf16")
    return dst


# ---------------------------------------------------------------- VQGAN primitives
def _f32c(t, name):
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous f32 ROCm device tensor (lwm_amd has no CPU path)")
    return t.data_ptr()


def conv2d_nhwc(x, w, bias=None, residual=None, *, stride=1, pad=None, up_shift=0, out_hw=None,
                clip=False, out=None):
    """flax nn.Conv on NHWC f32 (lwm_conv2d_nhwc_f32).  x (B,H,W,Cin), w (KH,KW,Cin,Cout) HWIO.
    pad=None = 'SAME' for stride 1.  Downsample: stride=2, pad=0, out_hw=(H//2, W//2);
    Upsample+conv: up_shift=1."""
    B, Hin, Win, Cin = x.shape
    KH, KW, Cin2, Cout = w.shape
    if Cin2 != Cin:
        raise ValueError(f"conv2d_nhwc: kernel expects {Cin2} input channels, x has {Cin}")
    if pad is None:
        pad = (KH - 1) // 2
    Hv, Wv = Hin << up_shift, Win << up_shift
    if out_hw is None:
        out_hw = ((Hv + 2 * pad - KH) // stride + 1, (Wv + 2 * pad - KW) // stride + 1)
    Ho, Wo = out_hw
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    if residual is not None and tuple(residual.shape) != tuple(out.shape):
        raise ValueError("conv2d_nhwc: residual shape mismatch")
    a = _capi.LwmConvArgs(_f32c(x, "x"), _f32c(w, "w"),
                          None if bias is None else _f32c(bias, "bias"),
                          None if residual is None else _f32c(residual, "residual"),
                          _f32c(out, "out"), B, Hin, Win, Cin, Cout, KH, KW, stride, pad, up_shift,
                          Ho, Wo, int(bool(clip)))
    L = lib()
    _capi.check(L, L.lwm_conv2d_nhwc_f32(C.byref(a), _stream_ptr()), "lwm_conv2d_nhwc_f32")
    return out


def groupnorm_silu(x, gamma, beta, *, groups=32, eps=1e-6, silu=True, out=None, workspace=None):
    """flax nn.GroupNorm (+ nn.silu) over NHWC / (B, ..., C) f32 (lwm_groupnorm_silu_f32)."""
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc) if B else 0
    if gamma.numel() != Cc or beta.numel() != Cc:
        raise ValueError(f"groupnorm_silu: scale/bias have {gamma.numel()}/{beta.numel()} elements, x has {Cc} channels")
    L = lib()
    need = L.lwm_groupnorm_workspace_bytes(B, HW, Cc, groups)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((max(need, 16) + 7) // 8, dtype=torch.float64, device=x.device)
    if out is None:
        out = torch.empty_like(x)
    _capi.check(L, L.lwm_groupnorm_silu_f32(_f32c(x, "x"), _f32c(gamma, "gamma"), _f32c(beta, "beta"),
                                            _f32c(out, "out"), workspace.data_ptr(), B, HW, Cc, groups,
                                            float(eps), int(bool(silu)), _stream_ptr()),
                "lwm_groupnorm_silu_f32")
    return out


def vq_sqnorm(codebook):
    E, D = codebook.shape
    se = torch.empty(E, dtype=torch.float32, device=codebook.device)
    L = lib()
    _capi.check(L, L.lwm_vq_sqnorm_f32(_f32c(codebook, "codebook"), se.data_ptr(), E, D, _stream_ptr()),
                "lwm_vq_sqnorm_f32")
    return se


def vq_argmin(z, codebook, se=None):
    """First argmin over the codebook of the f32 squared distances (lwm_vq_argmin_f32)."""
    E, D = codebook.shape
    if z.shape[-1] != D:
        raise ValueError("vq_argmin: embed dim mismatch")
    if se is None:
        se = vq_sqnorm(codebook)
    N = z.numel() // D
    idx = torch.empty(z.shape[:-1], dtype=torch.int32, device=z.device)
    L = lib()
    _capi.check(L, L.lwm_vq_argmin_f32(_f32c(z, "z"), _f32c(codebook, "codebook"), _f32c(se, "se"),
                                       idx.data_ptr(), N, E, D, _stream_ptr()), "lwm_vq_argmin_f32")
    return idx


def vq_gather(codebook, idx, z=None):
    E, D = codebook.shape
    if idx.dtype != torch.int32 or not idx.is_contiguous() or not idx.is_cuda:
        raise ValueError("vq_gather: idx must be a contiguous int32 device tensor")
    out = torch.empty(tuple(idx.shape) + (D,), dtype=torch.float32, device=codebook.device)
    L = lib()
    _capi.check(L, L.lwm_vq_gather_f32(_f32c(codebook, "codebook"), idx.data_ptr(),
                                       None if z is None else _f32c(z, "z"), out.data_ptr(),
                                       idx.numel(), E, D, _stream_ptr()), "lwm_vq_gather_f32")
    return out
