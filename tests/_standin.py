"""CPU stand-in for the per-block kernels, built on the oracle's mask/softmax
definitions.  TEST INFRASTRUCTURE: lets the ring driver's schedule, carries and
communication be exercised under gloo without a GPU.  Never used by lwm_amd."""
import math

import numpy as np
import torch

from oracle.attention_ref import visible_mask


def _positions(S, start, cuts):
    """positions of S rows under a piecewise map: rows from 0 at `start`, rows from a cut (row, position) on at its position"""
    pos = start + np.arange(S)
    for r, p in ([cuts] if isinstance(cuts, tuple) else list(cuts or [])):
        pos[r:] = p + np.arange(S - r)
    return pos


def _vis(q, k, q_start, k_start, causal, seg_q, seg_k, key_valid, q_piece2=None, k_piece2=None):
    B, Sq = q.shape[0], q.shape[1]
    Sk = k.shape[1]
    n = lambda t: None if t is None else t.cpu().numpy()
    m = visible_mask(Sq, Sk, causal=False, seg_q=n(seg_q), seg_k=n(seg_k), key_valid=n(key_valid), B=B)
    if causal:
        m = m & (_positions(Sk, k_start, k_piece2)[None, :] <= _positions(Sq, q_start, q_piece2)[:, None])[None]
    return torch.from_numpy(m)[:, None]  # B,1,Sq,Sk


class OracleBlockOps:
    piece_align = 1          # piecewise position maps at any row (the HIP kernels: multiples of 256)

    @staticmethod
    def empty(shape, dtype, like):
        return torch.empty(shape, dtype=dtype)

    @staticmethod
    def zeros(shape, dtype, like):
        return torch.zeros(shape, dtype=dtype)

    @staticmethod
    def cast(src, dst=None):
        if dst is None:
            return src.to(torch.bfloat16)
        dst.copy_(src.to(torch.bfloat16))
        return dst

    @staticmethod
    def sum_cast(srcs, dst=None):
        acc = srcs[0].clone()
        for x in srcs[1:]:
            acc = acc + x            # f32 adds in argument order, as the kernel
        if dst is None:
            return acc.to(torch.bfloat16)
        dst.copy_(acc.to(dst.dtype).reshape(dst.shape))
        return dst

    @staticmethod
    def fwd(q, k, v, *, q_start=0, k_start=0, causal=True, seg_q=None, seg_k=None, key_valid=None,
            scale=None, out=None, lse=None, out_acc=None, lse_acc=None, carry_in=False, final=True, q_piece2=None,
            k_piece2=None):
        D = q.shape[-1]
        scale = scale or 1.0 / math.sqrt(D)
        s = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * scale
        vis = _vis(q, k, q_start, k_start, causal, seg_q, seg_k, key_valid, q_piece2, k_piece2)
        s = s.masked_fill(~vis, float("-inf"))
        m = s.amax(dim=-1, keepdim=True)
        m = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        ob = torch.einsum("bhqk,bkhd->bqhd", p / l.clamp_min(1e-300), v.double())
        lb = torch.where(l[..., 0] > 0, m[..., 0] + torch.log(l[..., 0].clamp_min(1e-300)),
                         torch.full_like(l[..., 0], float("-inf")))
        if carry_in:
            la = lse_acc.double()
            ln = torch.logaddexp(la, lb)
            wa = torch.where(torch.isfinite(la), torch.exp(la - ln), torch.zeros_like(la))
            wb = torch.where(torch.isfinite(lb), torch.exp(lb - ln), torch.zeros_like(lb))
            wa, wb = (w.nan_to_num(0.0).permute(0, 2, 1)[..., None] for w in (wa, wb))
            ob = out_acc.double() * wa + ob * wb
            lb = ln
        if final:
            out.copy_(ob.to(out.dtype))
            lse.copy_(lb.float())
            return out, lse
        out_acc.copy_(ob.float())
        lse_acc.copy_(lb.float())
        return out_acc, lse_acc

    @staticmethod
    def bwd_delta(out, dout, lse=None, delta=None):      # (the stand-in keeps plain delta: its own kernels consume it)
        d = torch.einsum("bqhd,bqhd->bhq", out.double(), dout.double()).float()
        if delta is not None:
            delta.copy_(d)
            return delta
        return d

    @staticmethod
    def _ds(q, k, v, dout, lse, delta, q_start, k_start, causal, seg_q, seg_k, key_valid, scale, q_piece2=None, k_piece2=None):
        D = q.shape[-1]
        scale = scale or 1.0 / math.sqrt(D)
        s = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) * scale
        vis = _vis(q, k, q_start, k_start, causal, seg_q, seg_k, key_valid, q_piece2, k_piece2)
        ok = vis & torch.isfinite(lse)[..., None]
        p = torch.where(ok, torch.exp(torch.where(ok, s - lse.double()[..., None].nan_to_num(0, 0, 0), torch.zeros_like(s))),
                        torch.zeros_like(s))
        dp = torch.einsum("bqhd,bkhd->bhqk", dout.double(), v.double())
        ds = p * (dp - delta.double()[..., None]) * scale
        return p, ds

    @classmethod
    def bwd_dq(cls, q, k, v, dout, lse, delta, *, q_start=0, k_start=0, causal=True, seg_q=None,
               seg_k=None, key_valid=None, scale=None, dq=None, dq_acc=None, carry_in=False, final=True, q_piece2=None,
               k_piece2=None):
        _, ds = cls._ds(q, k, v, dout, lse, delta, q_start, k_start, causal, seg_q, seg_k, key_valid, scale, q_piece2, k_piece2)
        r = torch.einsum("bhqk,bkhd->bqhd", ds, k.double())
        if carry_in:
            r = r + dq_acc.double()
        if final:
            if dq is None:
                dq = torch.empty(q.shape, dtype=q.dtype)
            dq.copy_(r.to(dq.dtype))
            return dq
        dq_acc.copy_(r.float())
        return dq_acc

    @classmethod
    def bwd_dkdv(cls, q, k, v, dout, lse, delta, *, q_start=0, k_start=0, causal=True, seg_q=None,
                 seg_k=None, key_valid=None, scale=None, dk=None, dv=None, dk_acc=None, dv_acc=None,
                 carry_in=False, final=True, q_piece2=None, k_piece2=None):
        p, ds = cls._ds(q, k, v, dout, lse, delta, q_start, k_start, causal, seg_q, seg_k, key_valid, scale, q_piece2, k_piece2)
        rk = torch.einsum("bhqk,bqhd->bkhd", ds, q.double())
        rv = torch.einsum("bhqk,bqhd->bkhd", p, dout.double())
        if carry_in:
            rk = rk + dk_acc.double()
            rv = rv + dv_acc.double()
        if final:
            if dk is None:
                dk = torch.empty(k.shape, dtype=k.dtype)
                dv = torch.empty(k.shape, dtype=k.dtype)
            dk.copy_(rk.to(dk.dtype))
            dv.copy_(rv.to(dv.dtype))
            return dk, dv
        dk_acc.copy_(rk.float())
        dv_acc.copy_(rv.float())
        return dk_acc, dv_acc

    # ---- inference / cache (stand-ins for fwd_splitk, combine, cache_write)
    @staticmethod
    def fwd_splitk(q, k, v, *, k_splits, q_start=0, k_start=0, causal=False, seg_q=None, seg_k=None,
                   key_valid=None, dense_mask=None, scale=None):
        D = q.shape[-1]
        scale = scale or 1.0 / math.sqrt(D)
        Sk = k.shape[1]
        per = -(-Sk // k_splits)
        outs, lses = [], []
        for s_ in range(k_splits):
            sl = slice(s_ * per, min(Sk, (s_ + 1) * per))
            kk, vv = k[:, sl].double(), v[:, sl].double()
            sc = torch.einsum("bqhd,bkhd->bhqk", q.double(), kk) * scale
            if dense_mask is not None:
                sc = sc.masked_fill(~(dense_mask[:, None, :, sl] != 0), float("-inf"))
            m = sc.amax(dim=-1, keepdim=True) if kk.shape[1] else torch.full(sc.shape[:-1] + (1,), float("-inf"))
            m = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
            p = torch.exp(sc - m)
            l = p.sum(dim=-1, keepdim=True)
            outs.append(torch.einsum("bhqk,bkhd->bqhd", p / l.clamp_min(1e-300), vv).float())
            lses.append(torch.where(l[..., 0] > 0, m[..., 0] + torch.log(l[..., 0].clamp_min(1e-300)),
                                    torch.full_like(l[..., 0], float("-inf"))).float())
        return torch.stack(outs), torch.stack(lses)

    @staticmethod
    def combine(o_parts, lse_parts, *, want_bf16=True, **_):
        l = lse_parts.double()
        mx = l.amax(dim=0)
        w = torch.where(torch.isfinite(l), torch.exp(l - torch.where(torch.isfinite(mx), mx, torch.zeros_like(mx))),
                        torch.zeros_like(l))
        den = w.sum(dim=0)
        wq = (w / den.clamp_min(1e-300)).permute(0, 1, 3, 2)[..., None]       # P,B,Sq,H,1
        out = (o_parts.double() * wq).sum(dim=0)
        lse = torch.where(den > 0, mx + torch.log(den.clamp_min(1e-300)), torch.full_like(den, float("-inf")))
        return (out.to(torch.bfloat16) if want_bf16 else out.float()), lse.float()

    @staticmethod
    def cache_write(cache, src, *, dst_row0, src_row0=0, nrows=None):
        if nrows is None:
            nrows = src.shape[1] - src_row0
        cache[:, dst_row0:dst_row0 + nrows] = src[:, src_row0:src_row0 + nrows]
        return cache
