"""ringattention_inference flavour on MI355X: dense boolean mask, split-K +
combine, q_len != kv_len, KV-cache writes -- against the fp64 oracle.
Tolerance: bf16 operands/outputs, f32 accumulation -> rel max err <= 2e-2, lse 2e-3."""
import numpy as np
import pytest

from oracle import attention_ref as R

pytestmark = pytest.mark.gpu


def _rand(shape, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(torch.bfloat16)


def _np(t):
    return t.detach().float().cpu().numpy()


@pytest.mark.parametrize("B,Q,K,H,splits,cache_index", [
    (2, 1, 4096, 4, 8, 4000),
    (1, 1, 1000, 2, 3, 999),
    (1, 7, 2048, 2, 4, 1500),
    (1, 1, 8192, 32, 16, 100),     # mostly-empty cache: whole pieces masked
    (1, 300, 1024, 2, 1, 700),     # short prefill block, two q tiles, one piece
])
def test_decode_vs_oracle(B, Q, K, H, splits, cache_index):
    import torch
    from lwm_amd import ops
    q, k, v = _rand((B, Q, H, 128), 1), _rand((B, K, H, 128), 2), _rand((B, K, H, 128), 3)
    am = (np.random.default_rng(4).random((B, K)) > 0.1).astype(np.uint8)
    am[:, 0] = 1
    mask = R.decode_mask(B, Q, K, cache_index, am)
    md = torch.from_numpy(mask.astype(np.uint8)).cuda()
    o_parts, l_parts = ops.attn_fwd_splitk(q.cuda(), k.cuda(), v.cuda(), k_splits=splits, dense_mask=md)
    out, lse = ops.attn_combine(o_parts, l_parts)
    ro, rl = R.dense_attention(_np(q), _np(k), _np(v), causal=False, dense_mask=mask)
    assert np.abs(_np(out) - ro).max() / np.abs(ro).max() <= 2e-2
    assert np.abs(_np(lse) - rl).max() <= 2e-3


def test_ringattention_inference_api_and_cache():
    """The reference call pattern: prefill into the cache, then one decode step."""
    import torch
    from lwm_amd.ringattention import concatenate_to_cache, ringattention_inference
    B, H, D, max_len, P = 1, 4, 128, 2048, 1024
    ck = torch.zeros(B, max_len, H, D, dtype=torch.bfloat16, device="cuda")
    cv = torch.zeros_like(ck)
    k0, v0 = _rand((B, P, H, D), 1).cuda(), _rand((B, P, H, D), 2).cuda()
    idx = concatenate_to_cache(ck, cv, k0, v0, 0)
    k1, v1, q1 = _rand((B, 1, H, D), 3).cuda(), _rand((B, 1, H, D), 4).cuda(), _rand((B, 1, H, D), 5).cuda()
    idx = concatenate_to_cache(ck, cv, k1, v1, idx)
    assert idx == P + 1
    assert torch.equal(ck[:, :P], k0) and torch.equal(cv[:, P], v1[:, 0]) and not ck[:, P + 1:].any()
    am = torch.ones(B, max_len, dtype=torch.bool, device="cuda")
    am[:, 10:20] = False
    mask = (torch.arange(max_len, device="cuda")[None, None, None, :] <= (idx - 1)) & am[:, None, None, :]
    out = ringattention_inference(q1, ck, cv, mask)
    rmask = R.decode_mask(B, 1, max_len, idx - 1, am.cpu().numpy())
    ro, _ = R.dense_attention(_np(q1), _np(ck), _np(cv), causal=False, dense_mask=rmask)
    assert np.abs(_np(out) - ro).max() / np.abs(ro).max() <= 2e-2


def test_structured_prefill_mask_equals_the_dense_mask():
    """ringattention_inference(causal_offset=, key_valid=) -- the cache-present mask of lwm/llama.py:577-592
    handed over as its structure -- against the dense (B,1,Q,K) mask of the reference signature and the
    oracle: a 700-token block prefilled at cache_index 300 of a 4096-row cache, padded keys included."""
    import torch
    from lwm_amd.ringattention import ringattention_inference
    B, Q, K, H, idx = 2, 700, 4096, 4, 300
    q, k, v = _rand((B, Q, H, 128), 11).cuda(), _rand((B, K, H, 128), 12).cuda(), _rand((B, K, H, 128), 13).cuda()
    am = torch.ones(B, K, dtype=torch.int32, device="cuda")
    am[0, :17] = 0
    am[1, 450:460] = 0
    dense = ((torch.arange(K, device="cuda")[None, :] <= (torch.arange(Q, device="cuda") + idx)[:, None])[None, None]
             & (am[:, None, None, :] > 0))
    o_dense = ringattention_inference(q, k, v, dense)
    o_struct = ringattention_inference(q, k, v, None, causal_offset=idx, key_valid=am)
    ro, _ = R.dense_attention(_np(q), _np(k), _np(v), causal=False, dense_mask=_np(dense[:, 0]).astype(np.uint8))
    for o in (o_dense, o_struct):
        assert np.abs(_np(o) - ro).max() / np.abs(ro).max() <= 2e-2
    assert (o_dense.float() - o_struct.float()).abs().max().item() <= 1.6e-2 * np.abs(ro).max()
    # no key tile beyond the diagonal's own is read: poison the cache from the next 64-key tile on
    k2, v2 = k.clone(), v.clone()
    tail = -(-(idx + Q) // 64) * 64
    k2[:, tail:], v2[:, tail:] = float("nan"), float("nan")
    o2 = ringattention_inference(q, k2, v2, None, causal_offset=idx, key_valid=am)
    assert torch.equal(o2, o_struct)


def test_decode_full_size_cache_properties():
    """LWM-7B decode shapes: 32 heads, 131072-token cache shard (1 GiB of K+V per
    batch row): V = ones -> output exactly ones; masked tail has no influence."""
    import torch
    from lwm_amd import ops
    from lwm_amd.ring import _pick_splits
    B, Q, K, H = 1, 1, 131072, 32
    g = torch.Generator(device="cuda").manual_seed(0)
    k = torch.randn(B, K, H, 128, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    v = torch.randn(B, K, H, 128, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn(B, Q, H, 128, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    mask = torch.zeros(B, Q, K, dtype=torch.uint8, device="cuda")
    mask[:, :, :100000] = 1
    ns = _pick_splits(B, Q, H, K)
    out, lse = ops.attn_combine(*ops.attn_fwd_splitk(q, k, v, k_splits=ns, dense_mask=mask))
    o1, _ = ops.attn_combine(*ops.attn_fwd_splitk(q, k, torch.ones_like(v), k_splits=ns, dense_mask=mask))
    assert (o1.float() - 1).abs().max().item() <= 8e-3
    k2, v2 = k.clone(), v.clone()
    k2[:, 100000:] = 7.0
    v2[:, 100000:] = -5.0
    o2, l2 = ops.attn_combine(*ops.attn_fwd_splitk(q, k2, v2, k_splits=ns, dense_mask=mask))
    assert torch.equal(o2, out) and torch.equal(l2, lse)
    # one head against the oracle
    h = 11
    ro, rl = R.dense_attention(_np(q[:, :, h:h + 1]), _np(k[:, :, h:h + 1]), _np(v[:, :, h:h + 1]), causal=False,
                               dense_mask=mask.cpu().numpy())
    assert np.abs(_np(out[:, :, h:h + 1]) - ro).max() / np.abs(ro).max() <= 2e-2
    assert np.abs(_np(lse[:, h:h + 1]) - rl).max() <= 2e-3
