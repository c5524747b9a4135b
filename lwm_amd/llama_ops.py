"""RoPE and RMSNorm of FlaxLLaMAAttention / FlaxLLaMABlock on MI355X (SURVEY.md
section 8f rank 2): the reference's function / module names over the HIP kernels of
lwm_amd/csrc/llama_elem.h, differentiable (torch.autograd).

    freqs_cis = precompute_freqs_cis(head_dim, max_len, theta)      # lwm/llama.py:344-350
    xq, xk = apply_rotary_emb(xq, xk, freqs_cis, position_ids)       # lwm/llama.py:353-375
    y = RMSNorm(dim, eps)(x)                                         # lwm/llama.py:320-341

No CPU path: tensors must be bf16 on the ROCm device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from ._lib import lib
from .ops import _stream_ptr, _t4


def precompute_freqs_cis(dim: int, max_position_embedding: int, theta: float = 10000.0,
                         dtype=np.float32, device=None) -> torch.Tensor:
    """Host-side table, operation for operation as lwm/llama.py:344-350:
    freqs = 1/theta**(arange(0,dim,2)/dim) in `dtype`; angles = outer(t, freqs).astype(dtype);
    returned as f32 (max_pos, dim/2, 2) = (cos, sin) instead of complex64."""
    freqs = 1.0 / (theta ** (np.arange(0, dim, 2)[: (dim // 2)].astype(dtype) / dim))
    t = np.arange(max_position_embedding)
    ang = np.outer(t, freqs).astype(dtype)
    tab = np.stack((np.cos(ang), np.sin(ang)), axis=-1).astype(np.float32)
    out = torch.from_numpy(np.ascontiguousarray(tab))
    return out.to(device) if device is not None else out


def _rope(x, table, pos, conj):
    B, S, H, D = x.shape
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.stride(3) != 1:
        raise ValueError("rope: expected a bf16 (B,S,H,D) ROCm tensor with contiguous D")
    if not table.is_cuda or table.dtype != torch.float32 or not table.is_contiguous() or \
            tuple(table.shape[1:]) != (D // 2, 2):
        raise ValueError(f"rope: table must be a contiguous f32 (max_pos, {D // 2}, 2) device tensor")
    if pos.dtype != torch.int32 or not pos.is_contiguous() or tuple(pos.shape) != (B, S) or not pos.is_cuda:
        raise ValueError(f"rope: position_ids must be contiguous int32 {(B, S)} on the device")
    y = torch.empty((B, S, H, D), dtype=torch.bfloat16, device=x.device)
    L = lib()
    _capi.check(L, L.lwm_rope_bf16(_t4(x, "x"), _t4(y, "y"), table.data_ptr(), pos.data_ptr(), B, S, H, D,
                                   table.shape[0], int(conj), _stream_ptr()), "lwm_rope_bf16")
    return y


class _Rope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, pos):
        ctx.save_for_backward(table, pos)
        return _rope(x, table, pos, 0)

    @staticmethod
    def backward(ctx, g):
        table, pos = ctx.saved_tensors
        return _rope(g if g.stride(3) == 1 else g.contiguous(), table, pos, 1), None, None


def apply_rotary_emb(xq, xk, freqs_cis, position_ids=None, dtype=None):
    """lwm/llama.py:353-375 + the table gather of :515 (jnp.take(freqs_cis, position_ids)).
    xq, xk: (B,S,H,D) bf16; freqs_cis from precompute_freqs_cis; position_ids (B,S) int
    (default arange, as lwm/llama.py:1081-1082)."""
    B, S = xq.shape[:2]
    if position_ids is None:
        position_ids = torch.arange(S, device=xq.device, dtype=torch.int32)[None].expand(B, S)
    pos = position_ids.to(torch.int32).contiguous()
    return _Rope.apply(xq, freqs_cis, pos), _Rope.apply(xk, freqs_cis, pos)


class _RmsNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps):
        if not x.is_cuda or x.dtype != torch.bfloat16 or not x.is_contiguous():
            raise ValueError("RMSNorm: expected a contiguous bf16 ROCm tensor")
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        w = weight.to(torch.bfloat16).contiguous()
        y = torch.empty_like(x)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        L = lib()
        _capi.check(L, L.lwm_rmsnorm_fwd_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), rstd.data_ptr(), rows,
                                              Cc, float(eps), _stream_ptr()), "lwm_rmsnorm_fwd_bf16")
        ctx.save_for_backward(x, w, rstd)
        ctx.wdtype = weight.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, rstd = ctx.saved_tensors
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        g = g.contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty(Cc, dtype=torch.bfloat16, device=x.device)
        L = lib()
        ws = torch.empty(max(L.lwm_rmsnorm_bwd_workspace_bytes(rows, Cc), 16) // 4, dtype=torch.float32,
                         device=x.device)
        _capi.check(L, L.lwm_rmsnorm_bwd_bf16(x.data_ptr(), w.data_ptr(), g.data_ptr(), rstd.data_ptr(),
                                              dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), rows, Cc,
                                              _stream_ptr()), "lwm_rmsnorm_bwd_bf16")
        return dx, dw.to(ctx.wdtype), None


class RMSNorm(torch.nn.Module):
    """lwm/llama.py:320-341 (parameter name `kernel`, ones-initialised)."""

    def __init__(self, dim: int, eps: float = 1e-6, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.dim, self.eps, self.dtype = dim, eps, dtype
        self.kernel = torch.nn.Parameter(torch.ones(dim, dtype=param_dtype))

    def forward(self, x):
        return _RmsNorm.apply(x.to(self.dtype), self.kernel, self.eps)
