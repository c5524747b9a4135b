"""Synthetic needle-in-a-haystack at the operator level (the reference's functional check is
scripts/eval_needle.py, which needs trained weights; SURVEY.md section 8c asks for a synthetic
substitute): a "needle" key is planted at a chosen depth of a very long context, the query
points at it, and the attention output must return the needle's value -- through the decode
path over a 1,048,576-token KV cache and through the training forward kernel at 131,072
tokens, at several depths; with packed documents a needle in ANOTHER document must stay
invisible."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _haystack(S, H, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    k = torch.randn(1, S, H, 128, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    v = torch.randn(1, S, H, 128, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    return k, v


def _plant(k, v, pos, code):
    """needle: key = 6 * e_code direction (unit-norm pattern), value = one-hot(code) * 8"""
    import torch
    H = k.shape[2]
    pat = torch.zeros(128, device="cuda")
    pat[code] = 1.0
    pat[(code * 7 + 3) % 128] = -1.0
    k[0, pos] = (pat * 6.0).to(torch.bfloat16)[None].expand(H, 128)
    val = torch.zeros(128, device="cuda")
    val[code] = 8.0
    v[0, pos] = val.to(torch.bfloat16)[None].expand(H, 128)
    return (pat * 6.0).to(torch.bfloat16)


@pytest.mark.parametrize("depth", [0.0, 0.25, 0.5, 0.999999])
def test_decode_needle_in_1m_token_cache(depth):
    """Q = 1 against a 1,048,576-token cache (8 heads: 4 GiB of K+V)."""
    import torch
    from lwm_amd import ops
    from lwm_amd.ring import _pick_splits
    S, H, code = 1 << 20, 8, 37
    k, v = _haystack(S, H, 1)
    pos = min(S - 1, int(depth * S))
    pat = _plant(k, v, pos, code)
    q = (pat * 2.0)[None, None, None].expand(1, 1, H, 128).contiguous()      # q.k_needle = 2*72 -> logit 12.7
    mask = torch.ones(1, 1, S, dtype=torch.uint8, device="cuda")
    out, lse = ops.attn_combine(*ops.attn_fwd_splitk(q, k, v, k_splits=_pick_splits(1, 1, H, S), dense_mask=mask))
    o = out.float()[0, 0]                                                     # (H, 128)
    assert (o.argmax(-1) == code).all(), (depth, o.argmax(-1).tolist())
    assert (o[:, code] > 0.5).all()                 # the needle holds a visible share of the 1M-way softmax
    # hide the needle with the mask: the answer must disappear
    mask[0, 0, pos] = 0
    out2, _ = ops.attn_combine(*ops.attn_fwd_splitk(q, k, v, k_splits=_pick_splits(1, 1, H, S), dense_mask=mask))
    assert (out2.float()[0, 0, :, code].abs() < 0.25).all()


@pytest.mark.parametrize("depth", [0.0, 0.37, 0.99])
def test_training_forward_needle_128k(depth):
    """Causal forward at S = 131072 (4 heads): the LAST query row retrieves a needle planted
    anywhere before it; an earlier row (before the needle) cannot see it."""
    import torch
    from lwm_amd import ops
    S, H, code = 131072, 4, 90
    k, v = _haystack(S, H, 2)
    pos = min(S - 2, int(depth * S))
    pat = _plant(k, v, pos, code)
    q = torch.randn(1, S, H, 128, device="cuda", dtype=torch.float32).mul_(0.1).to(torch.bfloat16)
    q[0, S - 1] = (pat * 2.0)[None].expand(H, 128)
    early = max(0, pos - 1)
    if early < pos:
        q[0, early] = (pat * 2.0)[None].expand(H, 128)
    out, _ = ops.attn_fwd_block(q, k, v, causal=True)
    last = out.float()[0, S - 1]
    assert (last.argmax(-1) == code).all() and (last[:, code] > 0.5).all()
    if early < pos:
        assert (out.float()[0, early, :, code].abs() < 0.5).all()      # causality: needle is in its future
    # packed documents: put the needle in a different document than the last query
    seg = torch.zeros(1, S, dtype=torch.int32, device="cuda")
    seg[:, pos + 1:] = 1
    out_p, _ = ops.attn_fwd_block(q, k, v, causal=True, seg_q=seg, seg_k=seg)
    assert (out_p.float()[0, S - 1, :, code].abs() < 0.5).all()
