"""RoPE and RMSNorm of FlaxLLaMAAttention / FlaxLLaMABlock on MI355X (SURVEY.md
section 8f rank 2): the reference's function / module names over the HIP kernels of
lwm_amd/csrc/llama_elem.h, differentiable (torch.autograd).

    freqs_cis = precompute_freqs_cis(head_dim, max_len, theta)      # lwm/llama.py:344-350
    xq, xk = apply_rotary_emb(xq, xk, freqs_cis, position_ids)       # lwm/llama.py:353-375
    y = RMSNorm(dim, eps)(x)                                         # lwm/llama.py:320-341

No CPU path: tensors must be on the ROCm device -- bf16 (the headline dtype), or float32: the reference's `--dtype=fp32`
(lwm/train.py:36 default; BASELINE configs[0]), served by the f32 flavour of every kernel here (csrc/elem_f32.h: the same
arithmetic minus the roundings to bf16) and of the attention op (csrc/attn_f32.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from ._lib import lib
from .ops import _stream_ptr, _t4

_DTYPES = (torch.bfloat16, torch.float32)


def _fn(L, name, dtype):
    """lwm_<name>_bf16 or lwm_<name>_f32"""
    return getattr(L, f"lwm_{name}_{'f32' if dtype == torch.float32 else 'bf16'}")


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
    if not x.is_cuda or x.dtype not in _DTYPES or x.stride(3) != 1:
        raise ValueError("rope: expected a bf16 / f32 (B,S,H,D) ROCm tensor with contiguous D")
    if not table.is_cuda or table.dtype != torch.float32 or not table.is_contiguous() or \
            tuple(table.shape[1:]) != (D // 2, 2):
        raise ValueError(f"rope: table must be a contiguous f32 (max_pos, {D // 2}, 2) device tensor")
    if pos.dtype != torch.int32 or not pos.is_contiguous() or tuple(pos.shape) != (B, S) or not pos.is_cuda:
        raise ValueError(f"rope: position_ids must be contiguous int32 {(B, S)} on the device")
    y = torch.empty((B, S, H, D), dtype=x.dtype, device=x.device)
    L = lib()
    _capi.check(L, _fn(L, "rope", x.dtype)(_t4(x, "x"), _t4(y, "y"), table.data_ptr(), pos.data_ptr(), B, S, H, D,
                                           table.shape[0], int(conj), _stream_ptr()), "lwm_rope")
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
        if not x.is_cuda or x.dtype not in _DTYPES or not x.is_contiguous():
            raise ValueError("RMSNorm: expected a contiguous bf16 / f32 ROCm tensor")
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        w = weight.to(x.dtype).contiguous()
        y = torch.empty_like(x)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        L = lib()
        _capi.check(L, _fn(L, "rmsnorm_fwd", x.dtype)(x.data_ptr(), w.data_ptr(), y.data_ptr(), rstd.data_ptr(), rows,
                                                      Cc, float(eps), _stream_ptr()), "lwm_rmsnorm_fwd")
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
        dw = torch.empty(Cc, dtype=x.dtype, device=x.device)
        L = lib()
        ws = torch.empty(max(L.lwm_rmsnorm_bwd_workspace_bytes(rows, Cc), 16) // 4, dtype=torch.float32,
                         device=x.device)
        _capi.check(L, _fn(L, "rmsnorm_bwd", x.dtype)(x.data_ptr(), w.data_ptr(), g.data_ptr(), rstd.data_ptr(),
                                                      dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), rows, Cc,
                                                      _stream_ptr()), "lwm_rmsnorm_bwd")
        return dx, dw.to(ctx.wdtype), None


class RMSNorm(torch.nn.Module):
    """lwm/llama.py:320-341 (parameter name `kernel`, ones-initialised)."""

    def __init__(self, dim: int, eps: float = 1e-6, dtype=torch.bfloat16, param_dtype=torch.float32):
        super().__init__()
        self.dim, self.eps, self.dtype = dim, eps, dtype
        self.kernel = torch.nn.Parameter(torch.ones(dim, dtype=param_dtype))

    def forward(self, x):
        return _RmsNorm.apply(x.to(self.dtype), self.kernel, self.eps)


# ---------------------------------------------------------------- loss (lwm/train.py:177-202)
def _softmax_ce(logits2d, target, weight, want_grad):
    rows, V = logits2d.shape
    if not logits2d.is_cuda or logits2d.dtype not in _DTYPES or not logits2d.is_contiguous():
        raise ValueError("cross entropy: expected contiguous bf16 / f32 ROCm logits")
    nll = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
    correct = torch.empty(rows, dtype=torch.int32, device=logits2d.device)
    dl = torch.empty_like(logits2d) if want_grad else None
    L = lib()
    _capi.check(L, _fn(L, "softmax_ce", logits2d.dtype)(logits2d.data_ptr(), target.data_ptr(),
                                                        None if weight is None else weight.data_ptr(), nll.data_ptr(),
                                                        correct.data_ptr(), None if dl is None else dl.data_ptr(), rows, V,
                                                        _stream_ptr()), "lwm_softmax_ce")
    return nll, correct, dl


def _row_weights(valid, B, S, device, sp_sharded=True):
    """valid (B,S) or None -> (valid f32, per-row gradient weight valid / (max(sum_s valid, 1e-10) * B)).
    Over a sequence ring S is this rank's SHARD of each sequence: the count of valid targets is summed over the "sp"
    axis, so that each rank's loss is its share of the reference's per-sequence mean (tux.cross_entropy_loss_and_accuracy,
    lwm/train.py:177-181) and the shares ADD UP to it -- whatever the masks do to the counts per rank.
    CONTRACT of that sum (sp_sharded=True, the default): every rank of the "sp" group enters the call (it is a
    collective), and the rows are this rank's shard.  A loss on tokens that are REPLICATED on the sp ranks (evaluation,
    scoring) must say sp_sharded=False: no collective, the count is the local one -- with the default the count would
    come out n times too large and the loss n times too small, without any error."""
    v = torch.ones(B, S, dtype=torch.float32, device=device) if valid is None else valid.to(torch.float32)
    count = v.sum(dim=-1, keepdim=True)
    if sp_sharded:
        from .ringattention import sp_all_reduce_sum
        sp_all_reduce_sum(count, "sp")
    denom = count.clamp_min(1e-10) * B
    return v, (v / denom).contiguous()


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, tokens, valid, sp_sharded=True):
        B, S, V = logits.shape
        v, w = _row_weights(valid, B, S, logits.device, sp_sharded)
        nll, correct, dl = _softmax_ce(logits.reshape(B * S, V), tokens.reshape(-1).to(torch.int32).contiguous(),
                                       w.reshape(-1), logits.requires_grad)
        loss = (nll.reshape(B, S) * w).sum()
        acc = (correct.reshape(B, S).to(torch.float32) * w).sum()
        ctx.save_for_backward(dl)
        ctx.shape = (B, S, V)
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, g_loss, _g_acc):
        (dl,) = ctx.saved_tensors
        return (dl.reshape(ctx.shape) * g_loss.to(dl.dtype)), None, None, None


def cross_entropy_loss_and_accuracy(logits, tokens, valid=None, sp_sharded=True):
    """tux.cross_entropy_loss_and_accuracy (call sites lwm/train.py:177-181, :192-201):
    loss = -mean_b( sum_s valid*log p(token) / max(sum_s valid, 1e-10) ), accuracy likewise
    with argmax == token.  logits (B,S,V) bf16 (upcast to f32 in the kernel), tokens (B,S) int.
    Over a sequence ring: this rank's share of that mean, a collective over "sp" -- see _row_weights for the contract
    and for sp_sharded=False (replicated tokens)."""
    return _CrossEntropy.apply(logits, tokens, valid, sp_sharded)


class _ChunkedHeadLoss(torch.autograd.Function):
    """lm_head + cross entropy over sequence chunks: the (B,S,V) logits of a 1M-token batch
    (128 GB in f32) are never materialised -- each chunk's logits live only between its GEMM
    (hipBLASLt through torch.matmul: a plain library GEMM) and the fused loss/gradient kernel."""

    @staticmethod
    def forward(ctx, hidden, kernel, tokens, valid, chunk, sp_sharded=True):
        B, S, Dm = hidden.shape
        V = kernel.shape[1]
        v, w = _row_weights(valid, B, S, hidden.device, sp_sharded)
        tok = tokens.to(torch.int32)
        need = hidden.requires_grad or kernel.requires_grad
        loss = torch.zeros((), dtype=torch.float32, device=hidden.device)
        acc = torch.zeros((), dtype=torch.float32, device=hidden.device)
        dh = torch.empty_like(hidden) if need else None
        dk = torch.zeros(kernel.shape, dtype=torch.float32, device=hidden.device) if need else None
        kb = kernel.to(hidden.dtype if hidden.dtype == torch.float32 else torch.bfloat16)
        for s0 in range(0, S, chunk):
            s1 = min(S, s0 + chunk)
            h = hidden[:, s0:s1].reshape(-1, Dm)
            logits = (h @ kb).contiguous()
            nll, correct, dl = _softmax_ce(logits, tok[:, s0:s1].reshape(-1).contiguous(),
                                           w[:, s0:s1].reshape(-1).contiguous(), need)
            wc = w[:, s0:s1].reshape(-1)
            loss += (nll * wc).sum()
            acc += (correct.to(torch.float32) * wc).sum()
            if need:
                dh[:, s0:s1] = (dl @ kb.t()).reshape(B, s1 - s0, Dm)
                dk += wgrad(h, dl).to(torch.float32)
        ctx.save_for_backward(dh, dk)
        ctx.kdtype = kernel.dtype
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, g_loss, _g_acc):
        dh, dk = ctx.saved_tensors
        return dh * g_loss.to(dh.dtype), (dk * g_loss).to(ctx.kdtype), None, None, None, None


def chunked_lm_head_loss(hidden, lm_head_kernel, tokens, valid=None, chunk=8192, sp_sharded=True):
    """logits = hidden @ kernel (lwm/llama.py:1101, flax Dense kernel (d_model, vocab)) followed by
    cross_entropy_loss_and_accuracy, chunked over the sequence.  Returns (loss, accuracy).  Over a sequence ring the
    call is a collective over "sp" and returns this rank's share (sp_sharded, see _row_weights)."""
    return _ChunkedHeadLoss.apply(hidden, lm_head_kernel, tokens, valid, int(chunk), bool(sp_sharded))


def vision_text_loss(vision_logits, text_logits, target_tokens, loss_masks, target_vision_masks):
    """lwm/train.py:192-202: 0.5 * (vision CE + text CE) with the two masked target sets."""
    tvm = target_vision_masks.to(torch.bool)
    lm = loss_masks.to(torch.float32)
    zeros = torch.zeros_like(target_tokens)
    v_loss, v_acc = cross_entropy_loss_and_accuracy(vision_logits, torch.where(tvm, target_tokens, zeros),
                                                    lm * tvm.to(torch.float32))
    t_loss, t_acc = cross_entropy_loss_and_accuracy(text_logits, torch.where(tvm, zeros, target_tokens),
                                                    lm * (1.0 - tvm.to(torch.float32)))
    return 0.5 * (v_loss + t_loss), dict(vision_loss=v_loss, vision_acc=v_acc, text_loss=t_loss, text_acc=t_acc)


# ---------------------------------------------------------------- SwiGLU MLP (lwm/llama.py:623-661)
class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        for t in (a, b):
            if not t.is_cuda or t.dtype not in _DTYPES or t.dtype != a.dtype or not t.is_contiguous():
                raise ValueError("swiglu: expected contiguous bf16 / f32 ROCm tensors of one dtype")
        y = torch.empty_like(a)
        L = lib()
        _capi.check(L, _fn(L, "swiglu_fwd", a.dtype)(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream_ptr()),
                    "lwm_swiglu_fwd")
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        L = lib()
        _capi.check(L, _fn(L, "swiglu_bwd", a.dtype)(a.data_ptr(), b.data_ptr(), g.data_ptr(), da.data_ptr(), db.data_ptr(),
                                                     a.numel(), _stream_ptr()), "lwm_swiglu_bwd")
        return da, db


def swiglu(a, b):
    """silu(a) * b (the gate of FlaxLLaMAMLP, lwm/llama.py:659)."""
    return _SwiGLU.apply(a, b)


_GEMV_WS = {}


def gemv_multi(x, kernels, out_dtype=torch.bfloat16):
    """x (rows <= 4, K) bf16 against 1..3 kernels (K, N_i) bf16 that share it -> [(rows, N_i)] in `out_dtype`
    (bf16 or f32) through ONE lwm_gemv_multi_bf16 call: every kernel streamed once from HBM, f32 accumulation
    along a fixed tree."""
    rows, K = x.shape
    if x.dtype != torch.bfloat16 or x.stride(1) != 1 or any(
            k.dtype != torch.bfloat16 or not k.is_contiguous() or k.shape[0] != K for k in kernels):
        raise ValueError("gemv: expected bf16 x (contiguous rows) and contiguous bf16 (K, N) kernels")
    L = lib()
    n = len(kernels)
    Ns = [int(k.shape[1]) for k in kernels]
    key = (x.device, rows, K, tuple(Ns))
    ws = _GEMV_WS.get(key)
    if ws is None:                     # (one workspace per shape: a hipGraph replays with the pointers it captured)
        need = sum(L.lwm_gemv_workspace_bytes(rows, K, N) for N in Ns)
        ws = _GEMV_WS[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
    ys = [torch.empty(rows, N, dtype=out_dtype, device=x.device) for N in Ns]
    f32 = out_dtype == torch.float32
    vp = C.c_void_p * n
    w_arr = vp(*[k.data_ptr() for k in kernels])
    y_arr = vp(*[y.data_ptr() for y in ys])
    _capi.check(L, L.lwm_gemv_multi_bf16(x.data_ptr(), x.stride(0), n, w_arr, None if f32 else y_arr, (C.c_int64 * n)(*Ns),
                                         y_arr if f32 else None, (C.c_int32 * n)(*Ns), ws.data_ptr(), rows, K,
                                         _stream_ptr()), "lwm_gemv_multi_bf16")
    return ys


def gemv(x, kernel, out_dtype=torch.bfloat16):
    return gemv_multi(x, [kernel], out_dtype)[0]


def gemv_fused(x, kernels, *, norm=None, residual=None, want_ss=False, out_dtype=torch.bfloat16):
    """lwm_gemv_fused_bf16: gemv_multi with the neighbouring launches of a decode step riding along.
    norm = (ss (rows, n <= 64) f32 partial sums of squares of x's rows, weight (K,) bf16, eps): RMSNorm on load;
    residual (rows, N) bf16 (one kernel): y = bf16(bf16(x @ W) + residual); want_ss: also return the (rows, N / 128)
    partial sums of squares of y for the next norm.  -> [y_i] or ([y_i], ss)."""
    rows, K = x.shape
    if x.dtype != torch.bfloat16 or x.stride(1) != 1 or any(
            k.dtype != torch.bfloat16 or not k.is_contiguous() or k.shape[0] != K for k in kernels):
        raise ValueError("gemv: expected bf16 x (contiguous rows) and contiguous bf16 (K, N) kernels")
    L = lib()
    n = len(kernels)
    Ns = [int(k.shape[1]) for k in kernels]
    key = (x.device, rows, K, tuple(Ns))
    ws = _GEMV_WS.get(key)
    if ws is None:
        need = sum(L.lwm_gemv_workspace_bytes(rows, K, N) for N in Ns)
        ws = _GEMV_WS[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
    ys = [torch.empty(rows, N, dtype=out_dtype, device=x.device) for N in Ns]
    a = _capi.LwmGemvArgs()
    a.x, a.ldx, a.nmat, a.rows, a.K = x.data_ptr(), x.stride(0), n, rows, K
    a.workspace = ws.data_ptr()
    for i in range(n):
        a.w[i], a.N[i] = kernels[i].data_ptr(), Ns[i]
        if out_dtype == torch.float32:
            a.y_f32[i] = ys[i].data_ptr()
        else:
            a.y[i], a.ldy[i] = ys[i].data_ptr(), Ns[i]
    if norm is not None:
        ss, w, eps = norm
        if ss.dtype != torch.float32 or not ss.is_contiguous() or ss.shape[0] != rows or ss.shape[1] > 64 or \
                w.dtype != torch.bfloat16 or not w.is_contiguous() or w.numel() != K:
            raise ValueError("gemv_fused: norm = (ss (rows, n <= 64) f32 contiguous, weight (K,) bf16, eps)")
        a.norm_weight, a.ss_in, a.ss_n, a.eps = w.data_ptr(), ss.data_ptr(), ss.shape[1], float(eps)
    if residual is not None:
        if n != 1 or residual.dtype != torch.bfloat16 or tuple(residual.shape) != (rows, Ns[0]) or residual.stride(1) != 1:
            raise ValueError("gemv_fused: residual goes with ONE kernel and has its output's shape")
        a.residual[0], a.ldres[0] = residual.data_ptr(), residual.stride(0)
    ss_out = None
    if want_ss:
        ss_out = torch.empty(rows, Ns[0] // 128, dtype=torch.float32, device=x.device)
        a.ss_out = ss_out.data_ptr()
    _capi.check(L, L.lwm_gemv_fused_bf16(C.byref(a), _stream_ptr()), "lwm_gemv_fused_bf16")
    return (ys, ss_out) if want_ss else ys


def _decode_rows(x, kernels):
    rows = x.numel() // x.shape[-1]
    ok = rows <= 4 and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 32 == 0 and x.shape[-1] <= 12288 and \
        not (torch.is_grad_enabled() and (x.requires_grad or any(k.requires_grad for k in kernels))) and \
        all(k.dtype == torch.bfloat16 and k.is_contiguous() and k.shape[1] % 8 == 0 for k in kernels)
    return rows if ok else 0


def dense(x, kernel, out_dtype=None):
    """flax nn.Dense without bias, `x @ kernel` (lwm/llama.py:427-432, :659).  A cached-decode step (at most four
    rows in all, no autograd) streams the kernel through lwm_gemv_bf16; everything else is the library GEMM."""
    return dense_multi(x, (kernel,), out_dtype)[0]


def dense_multi(x, kernels, out_dtype=None):
    """[x @ k for k in kernels] -- projections that share their input (wq | wk | wv, w1 | w3).  In a cached-decode
    step they ride in one GEMV launch pair."""
    rows = _decode_rows(x, kernels)
    if rows:
        x2 = x.reshape(rows, x.shape[-1])
        out = []
        for i in range(0, len(kernels), 3):
            out += gemv_multi(x2, list(kernels[i:i + 3]), out_dtype or torch.bfloat16)
        return [y.reshape(*x.shape[:-1], y.shape[-1]) for y in out]
    if out_dtype in (None, x.dtype):
        return [x @ k for k in kernels]
    return [x.to(out_dtype) @ _as_dtype(k, out_dtype) for k in kernels]


def _as_dtype(k, dtype):
    """kernel.to(dtype), kept while the kernel is unchanged (a sampling loop with more than four rows asks for the
    f32 copy of the same head at every step).  The copy is kept ON the tensor object (it dies with it: a cache keyed
    by address would hand the previous model's head to a new model that the allocator placed at the same address)."""
    if torch.is_grad_enabled() and k.requires_grad:
        return k.to(dtype)
    hit = getattr(k, "_lwm_cast", None)
    if hit is None or hit[0] != (k._version, dtype, k.data_ptr()):
        hit = ((k._version, dtype, k.data_ptr()), k.detach().to(dtype))
        k._lwm_cast = hit
    return hit[1]


# ---------------------------------------------------------------- library GEMMs of the TRAINING path, laid out for hipBLASLt
# The projections stay plain library GEMMs (SURVEY.md section 8f allows it); what this section decides is HOW they are
# handed to the library (profiles/r06_model_full.md: 46 % of the LWM-7B step at S = 32768 is these GEMMs):
#   * projections that share their input run as ONE GEMM (wq | wk | wv -> N = 3d, w1 | w3 -> N = 2F): dgrad becomes one
#     GEMM over K = sum N instead of a sum of GEMM results, wgrad one GEMM instead of three;
#   * hipBLASLt is fastest when both operands have the REDUCTION dimension contiguous (1.36-1.58 PF/s against 1.04-1.34
#     forward and 0.90-1.05 wgrad in the flax (in, out) layout): the kernels are re-laid as (out, in) -- lwm_transpose_bf16,
#     cached per parameter version -- for the forward, used as they are for dgrad, and the NARROW operand of each weight
#     gradient is transposed so that S is contiguous;
#   * the residual add rides in the GEMM epilogue (beta = 1) forward, and in the RMSNorm backward kernel backward.
# Parameters keep the reference's names and (in, out) shapes (lwm/llama.py:390-421, :631-655).
import os as _os

_RELAYOUT_EPOCH = [0]


def weights_changed():
    """Tell the re-layout cache that kernels were written through a path that does not bump tensor versions (a load
    into .data, an optimizer that swaps storage) -- or, in a benchmark without an optimizer step, that the next step
    must pay for its re-layouts as a training step would."""
    _RELAYOUT_EPOCH[0] += 1


def use_tuned_gemms(path=None):
    """Hand hipBLASLt / rocBLAS the solution a tuning run picked for each GEMM of the LWM-7B step at S = 32768
    (lwm_amd/gemm_tuning_gfx950.csv: PyTorch TunableOp results of scripts/gpu_tune_gemms.py over the call forms of this
    module; +9 % on the wqkv and w2 weight gradients, level elsewhere -- profiles/r06_gemm_tuning_log.txt).  Tuning itself
    stays OFF: shapes that are not in the file take the library's default, and a file of another ROCm / hipBLASLt version is
    ignored by its validators.  Process-wide (TunableOp is a global switch), hence a call the entry points make
    (bench.py's model legs, lwm_amd.cli.train), not an import side effect.  LWM_GEMM_TUNING=0 opts out.  -> bool."""
    if _os.environ.get("LWM_GEMM_TUNING", "1") == "0" or not torch.cuda.is_available():
        return False
    path = path or _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gemm_tuning_gfx950.csv")
    if not _os.path.exists(path):
        return False
    try:
        import shutil
        import tempfile
        import torch.cuda.tunable as T
        # (TunableOp rewrites its file when the process ends: it gets a private copy, never the tracked one)
        tmp = _os.path.join(tempfile.mkdtemp(prefix="lwm_gemm_tuning_"), "results.csv")
        shutil.copyfile(path, tmp)
        T.enable(True)
        T.tuning_enable(False)
        T.set_filename(tmp)
        return bool(T.read_file(tmp))
    except Exception:       # noqa: BLE001 -- an optimisation hint: the default solutions are always there
        return False


def transpose2d(src, out=None):
    """(R, C) bf16 with contiguous rows (any row stride) -> (C, R): lwm_transpose_bf16 for multiples of 64, else torch."""
    R, Cc = src.shape
    if out is None:
        out = torch.empty(Cc, R, dtype=src.dtype, device=src.device)
    if (src.is_cuda and src.dtype == torch.bfloat16 and src.stride(1) == 1 and out.stride(1) == 1 and R % 64 == 0 and
            Cc % 64 == 0 and src.stride(0) % 8 == 0 and out.stride(0) % 8 == 0 and src.data_ptr() % 16 == 0 and
            out.data_ptr() % 16 == 0):
        L = lib()
        _capi.check(L, L.lwm_transpose_bf16(src.data_ptr(), src.stride(0), out.data_ptr(), out.stride(0), R, Cc,
                                            _stream_ptr()), "lwm_transpose_bf16")
    else:
        out.copy_(src.t())
    return out


_WGRAD_WS = {}


def wgrad(x2, g2):
    """dW (K, N) = x2^T g2 for x2 (M, K), g2 (M, N): the weight gradient of a flax Dense kernel (lwm/llama.py:390-421,
    :631-655).  bf16 on the device with M % 32 == 0, K % 256 == 0, N % 256 == 0 -> lwm_wgrad_bf16 (hand-written: both
    operands read where they lie, csrc/gemm_wgrad.h; 1.1-1.2 PF/s against the library's 0.9-1.05 in this layout and
    1.1-1.17 behind a transposed copy of the narrow operand -- profiles/r06_wgrad.md); anything else -> the narrow operand
    transposed by lwm_transpose_bf16 and a library GEMM.  LWM_WGRAD_HIP=0: the library form throughout."""
    M, K = x2.shape
    N = g2.shape[1]
    if x2.dtype == torch.float32:
        return x2.t() @ g2                              # the fp32 flavour: a plain library GEMM
    if (_os.environ.get("LWM_WGRAD_HIP", "1") == "1" and x2.is_cuda and x2.dtype == torch.bfloat16 and g2.dtype == torch.bfloat16
            and M % 32 == 0 and K % 256 == 0 and N % 256 == 0 and x2.stride(1) == 1 and g2.stride(1) == 1
            and x2.stride(0) % 8 == 0 and g2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0 and g2.data_ptr() % 16 == 0
            and 64 * max(x2.stride(0), g2.stride(0)) < 2 ** 31):
        L = lib()
        dw = torch.empty(K, N, dtype=torch.bfloat16, device=x2.device)
        nws = int(L.lwm_wgrad_workspace_bytes(M, K, N))
        ws = None
        if nws:
            # one workspace per device and stream: launches on a stream are ordered, the partials of one call are consumed
            # by its own second launch
            key = (x2.device.index, _stream_ptr().value)
            ws = _WGRAD_WS.get(key)
            if ws is None or ws.numel() < nws:
                ws = _WGRAD_WS[key] = torch.empty(nws, dtype=torch.uint8, device=x2.device)
        _capi.check(L, L.lwm_wgrad_bf16(x2.data_ptr(), x2.stride(0), g2.data_ptr(), g2.stride(0), dw.data_ptr(), N, M, K, N,
                                        ws.data_ptr() if ws is not None else None, nws, _stream_ptr()), "lwm_wgrad_bf16")
        return dw
    if K <= N:
        return transpose2d(x2) @ g2                     # x transposed: S contiguous in the A operand
    return x2.t() @ transpose2d(g2).t()                 # g transposed: S contiguous in the B operand


def _relayout(kernels):
    """[(K, N_i)] flax kernels -> (wt (sum N_i, K) for the forward, wcat (K, sum N_i) for dgrad), kept on the first kernel
    while none of them changes (tensor versions + weights_changed())."""
    k0 = kernels[0]
    key = (_RELAYOUT_EPOCH[0],) + tuple((k._version, k.data_ptr()) for k in kernels)
    hit = getattr(k0, "_lwm_relayout", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    K = k0.shape[0]
    Ns = [int(k.shape[1]) for k in kernels]
    with torch.no_grad():
        wt = torch.empty(sum(Ns), K, dtype=k0.dtype, device=k0.device)
        off = 0
        for k, n in zip(kernels, Ns):
            transpose2d(k.detach(), wt[off:off + n])
            off += n
        wcat = k0.detach() if len(kernels) == 1 else torch.cat([k.detach() for k in kernels], dim=1)
    k0._lwm_relayout = (key, wt, wcat)
    return wt, wcat


def fused_dense_ok(x, kernels):
    """The fused / re-laid path serves bf16 GEMMs on the device with more rows than a decode step has."""
    return (_os.environ.get("LWM_DENSE_FUSED", "1") == "1" and x.is_cuda and x.dtype == torch.bfloat16 and
            x.numel() // x.shape[-1] > 4 and
            all(k.is_cuda and k.dtype == torch.bfloat16 and k.is_contiguous() and k.dim() == 2 and
                k.shape[0] == x.shape[-1] for k in kernels))


def _dense_fwd(x2, kernels, residual2=None, out=None):
    wt, _ = _relayout(kernels)
    if residual2 is None:
        return torch.matmul(x2, wt.t(), out=out) if out is not None else x2 @ wt.t()
    return torch.addmm(residual2, x2, wt.t(), out=out) if out is not None else torch.addmm(residual2, x2, wt.t())


def _dense_bwd(x2, kernels, g2, need_dx=True, need_dw=True):
    """-> (dx2 (M, K) or None, [dW_i (K, N_i)] or None) from g2 (M, sum N_i)"""
    _, wcat = _relayout(kernels)
    dx2 = g2 @ wcat.t() if need_dx else None
    if not need_dw:
        return dx2, None
    dw = wgrad(x2, g2)
    if len(kernels) == 1:
        return dx2, [dw]
    return dx2, list(dw.split([int(k.shape[1]) for k in kernels], dim=1))


class _DenseFused(torch.autograd.Function):
    """y = x @ [k_0 | k_1 | ...] (+ residual): one library GEMM for projections that share their input."""

    @staticmethod
    def forward(ctx, x, residual, *kernels):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        Nt = sum(int(k.shape[1]) for k in kernels)
        r2 = None if residual is None else residual.reshape(-1, Nt)
        y = _dense_fwd(x2, kernels, r2)
        ctx.save_for_backward(x2, *kernels)
        ctx.xshape, ctx.has_res = x.shape, residual is not None
        return y.reshape(*x.shape[:-1], Nt)

    @staticmethod
    def backward(ctx, g):
        x2, *kernels = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        dx2, dws = _dense_bwd(x2, kernels, g2, ctx.needs_input_grad[0], any(ctx.needs_input_grad[2:]))
        dws = dws or [None] * len(kernels)
        return (None if dx2 is None else dx2.reshape(ctx.xshape), g if ctx.has_res else None, *dws)


def dense_fused(x, kernels, residual=None):
    """x @ concat(kernels, axis=1) (+ residual), the kernels flax Dense kernels (in, out_i) that share their input
    (lwm/llama.py:427-432 wq / wk / wv, :659 w1 / w3) or a single one (wo, w2); the residual add (lwm/llama.py:726, :743)
    is the GEMM's epilogue."""
    return _DenseFused.apply(x, residual, *kernels)


class _QKVRope(torch.autograd.Function):
    """(x, wq, wk, wv) -> (xq, xk, xv) as (B,S,H,D) views of ONE (B,S,3,H,D) buffer, RoPE applied to xq and xk in place in
    ONE launch (they are neighbours: a (B,S,2H,D) tensor).  lwm/llama.py:494-520.  Backward: the conjugate rotation writes
    dq / dk straight into the (B,S,3,H,D) gradient buffer the fused dgrad / wgrad GEMMs read."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, table, pos, H):
        B, S, d = x.shape
        D = wq.shape[1] // H
        x2 = x.reshape(-1, d)
        qkv = torch.empty(B, S, 3, H, D, dtype=x.dtype, device=x.device)
        _dense_fwd(x2, (wq, wk, wv), out=qkv.view(B * S, 3 * H * D))
        qk = qkv[:, :, 0:2].reshape(B, S, 2 * H, D)          # (a view: q and k heads are neighbours)
        assert qk.data_ptr() == qkv.data_ptr()
        L = lib()
        _capi.check(L, L.lwm_rope_bf16(_t4(qk, "qk"), _t4(qk, "qk"), table.data_ptr(), pos.data_ptr(), B, S, 2 * H, D,
                                       table.shape[0], 0, _stream_ptr()), "lwm_rope_bf16")
        ctx.save_for_backward(x2, wq, wk, wv, table, pos)
        ctx.dims = (B, S, d, H, D)
        return qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]

    @staticmethod
    def backward(ctx, gq, gk, gv):
        x2, wq, wk, wv, table, pos = ctx.saved_tensors
        B, S, d, H, D = ctx.dims
        g = torch.empty(B, S, 3, H, D, dtype=x2.dtype, device=x2.device)
        L = lib()
        for i, t in ((0, gq), (1, gk)):
            t = t if t.stride(3) == 1 else t.contiguous()
            _capi.check(L, L.lwm_rope_bf16(_t4(t, "g"), _t4(g[:, :, i], "g"), table.data_ptr(), pos.data_ptr(), B, S, H, D,
                                           table.shape[0], 1, _stream_ptr()), "lwm_rope_bf16")
        g[:, :, 2].copy_(gv)
        dx2, dws = _dense_bwd(x2, (wq, wk, wv), g.view(B * S, 3 * H * D), ctx.needs_input_grad[0],
                              any(ctx.needs_input_grad[1:4]))
        dws = dws or [None] * 3
        return (None if dx2 is None else dx2.reshape(B, S, d), *dws, None, None, None)


def qkv_rope(x, wq, wk, wv, freqs_cis, position_ids, num_heads):
    """FlaxLLaMAAttention's projections + head split + apply_rotary_emb (lwm/llama.py:494-520) as one operator."""
    B, S = x.shape[:2]
    if position_ids is None:
        position_ids = torch.arange(S, device=x.device, dtype=torch.int32)[None].expand(B, S)
    pos = position_ids.to(torch.int32).contiguous()
    if not freqs_cis.is_cuda or freqs_cis.dtype != torch.float32 or not freqs_cis.is_contiguous():
        raise ValueError("qkv_rope: freqs_cis must be the contiguous f32 device table of precompute_freqs_cis")
    return _QKVRope.apply(x, wq, wk, wv, freqs_cis, pos, int(num_heads))


class _RmsNormRes(torch.autograd.Function):
    """(x) -> (RMSNorm(x), x): the block's `x` feeds the norm AND the residual add behind it (lwm/llama.py:704-744); handing
    it through here lets the backward add the residual branch's gradient inside the RMSNorm backward kernel
    (lwm_rmsnorm_bwd_res_bf16) instead of in a separate pass over (rows, C)."""

    @staticmethod
    def forward(ctx, x, weight, eps):
        y = _RmsNorm.forward(ctx, x, weight, eps)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, g, g_pass):
        x, w, rstd = ctx.saved_tensors
        if g is None:
            return g_pass, None, None
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        g = g.contiguous()
        res = None if g_pass is None else g_pass.contiguous()
        dx = torch.empty_like(x)
        dw = torch.empty(Cc, dtype=torch.bfloat16, device=x.device)
        L = lib()
        ws = torch.empty(max(L.lwm_rmsnorm_bwd_workspace_bytes(rows, Cc), 16) // 4, dtype=torch.float32, device=x.device)
        _capi.check(L, L.lwm_rmsnorm_bwd_res_bf16(x.data_ptr(), w.data_ptr(), g.data_ptr(), rstd.data_ptr(),
                                                  None if res is None else res.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                  ws.data_ptr(), rows, Cc, _stream_ptr()), "lwm_rmsnorm_bwd_res_bf16")
        return dx, dw.to(ctx.wdtype), None


def rmsnorm_residual(norm, x):
    """-> (norm(x), x) with the residual branch's gradient folded into the norm's backward kernel."""
    return _RmsNormRes.apply(x.to(norm.dtype), norm.kernel, norm.eps)


class _SwiGLUHalves(torch.autograd.Function):
    """silu(y[..., :F]) * y[..., F:] on the two halves of one (rows, 2F) GEMM output; d gate | d up land in the halves of
    one (rows, 2F) buffer (lwm_swiglu_*_ld_bf16)."""

    @staticmethod
    def forward(ctx, y13):
        F2 = y13.shape[-1]
        F = F2 // 2
        y2 = y13.reshape(-1, F2)
        if not y2.is_cuda or y2.dtype != torch.bfloat16 or y2.stride(1) != 1 or F % 8:
            raise ValueError("swiglu_halves: expected a bf16 ROCm tensor (..., 2F) with F % 8 == 0")
        rows = y2.shape[0]
        out = torch.empty(rows, F, dtype=y2.dtype, device=y2.device)
        L = lib()
        _capi.check(L, L.lwm_swiglu_fwd_ld_bf16(y2.data_ptr(), y2.stride(0), y2.data_ptr() + 2 * F, y2.stride(0),
                                                out.data_ptr(), F, rows, F, _stream_ptr()), "lwm_swiglu_fwd_ld_bf16")
        ctx.save_for_backward(y2)
        ctx.shape = y13.shape
        return out.reshape(*y13.shape[:-1], F)

    @staticmethod
    def backward(ctx, g):
        (y2,) = ctx.saved_tensors
        rows, F2 = y2.shape
        F = F2 // 2
        g2 = g.reshape(rows, F)
        if g2.stride(1) != 1 or g2.stride(0) % 8 or g2.data_ptr() % 16:
            g2 = g2.contiguous()
        d13 = torch.empty(rows, F2, dtype=y2.dtype, device=y2.device)
        L = lib()
        _capi.check(L, L.lwm_swiglu_bwd_ld_bf16(y2.data_ptr(), y2.stride(0), y2.data_ptr() + 2 * F, y2.stride(0),
                                                g2.data_ptr(), g2.stride(0), d13.data_ptr(), F2, d13.data_ptr() + 2 * F, F2,
                                                rows, F, _stream_ptr()), "lwm_swiglu_bwd_ld_bf16")
        return d13.reshape(ctx.shape)


def swiglu_halves(y13):
    return _SwiGLUHalves.apply(y13)


class LLaMAMLP(torch.nn.Module):
    """FlaxLLaMAMLP (lwm/llama.py:623-661): w2(silu(w1 x) * w3 x), flax Dense kernels
    (in, out), no bias.  The three GEMMs are library GEMMs (hipBLASLt via torch.matmul); in a cached-decode
    step (<= 4 rows) they are lwm_gemv_bf16 launches (`dense`)."""

    def __init__(self, hidden_size, intermediate_size, dtype=torch.bfloat16, initializer_range=0.02):
        super().__init__()
        mk = lambda i, o: torch.nn.Parameter(torch.randn(i, o, dtype=dtype) * initializer_range)
        self.w1, self.w2, self.w3 = mk(hidden_size, intermediate_size), mk(intermediate_size, hidden_size), \
            mk(hidden_size, intermediate_size)

    def forward(self, x, residual=None):
        """residual (extension): added to the result (the block's `x + ff`, lwm/llama.py:743) -- in the w2 GEMM's epilogue
        on the fused path."""
        if fused_dense_ok(x, (self.w1, self.w3)) and self.w2.dtype == torch.bfloat16 and self.w2.is_contiguous() and \
                self.w1.shape[1] % 8 == 0:
            # w1 | w3 as one GEMM, the gate on the halves of its output, w2 with the residual as its epilogue
            return dense_fused(swiglu_halves(dense_fused(x, (self.w1, self.w3))), (self.w2,), residual)
        gate, up = dense_multi(x, (self.w1, self.w3))
        out = dense(swiglu(gate.contiguous(), up.contiguous()), self.w2)
        return out if residual is None else residual + out
