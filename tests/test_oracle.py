"""The oracle checked against itself along the structural identities the
reference's code implies (the reference ships no golden
vectors, so parity is UNPINNED -- see oracle/attention_ref.py):
  blockwise == dense branch (lwm/llama.py:525-570 vs :571-614),
  ring n == ring 1, packed == per-segment, analytic backward == autograd."""
import numpy as np
import pytest
import torch

from oracle import attention_ref as R


def _data(B, S, H, D, seed):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((B, S, H, D)).astype(np.float32) for _ in range(4)]


@pytest.mark.parametrize("ring", [1, 2, 4])
@pytest.mark.parametrize("causal", [True, False])
def test_blockwise_ring_equals_dense(ring, causal):
    q, k, v, _ = _data(2, 256, 2, 16, 0)
    ref, rlse = R.dense_attention(q, k, v, causal=causal)
    out, lse = R.blockwise_ring_attention(q, k, v, ring=ring, q_chunk=32, k_chunk=16, causal=causal,
                                          return_stats=True)
    assert np.abs(out - ref).max() < 2e-5
    assert np.abs(lse - rlse).max() < 2e-5


def test_packed_equals_per_segment_and_padding():
    B, S, H, D = 1, 192, 2, 16
    q, k, v, do = _data(B, S, H, D, 1)
    bounds = [0, 50, 120, 192]
    seg = np.zeros((B, S), np.int32)
    for i in range(3):
        seg[:, bounds[i]:bounds[i + 1]] = i
    out, _ = R.dense_attention(q, k, v, causal=True, seg_q=seg, seg_k=seg)
    blk = R.blockwise_ring_attention(q, k, v, ring=2, q_chunk=32, k_chunk=32, causal=True, segment_ids=seg)
    dq, dk, dv = R.dense_attention_bwd(q, k, v, do, causal=True, seg_q=seg, seg_k=seg)
    for i in range(3):
        sl = slice(bounds[i], bounds[i + 1])
        o_i, _ = R.dense_attention(q[:, sl], k[:, sl], v[:, sl], causal=True)
        assert np.abs(out[:, sl] - o_i).max() < 1e-12
        assert np.abs(blk[:, sl] - o_i).max() < 2e-5
        gq, gk, gv = R.dense_attention_bwd(q[:, sl], k[:, sl], v[:, sl], do[:, sl], causal=True)
        assert np.abs(dq[:, sl] - gq).max() < 1e-10 and np.abs(dk[:, sl] - gk).max() < 1e-10 \
            and np.abs(dv[:, sl] - gv).max() < 1e-10
    # key padding: padded keys never contribute; fully-masked rows are 0 / -inf
    kvm = np.ones((B, S), np.uint8)
    kvm[:, :7] = 0
    out2, lse2 = R.dense_attention(q, k, v, causal=True, key_valid=kvm)
    assert np.all(out2[:, :7] == 0) and np.all(np.isneginf(lse2[:, :, :7]))
    o_ref, _ = R.dense_attention(q[:, 7:], k[:, 7:], v[:, 7:], causal=True)
    assert np.abs(out2[:, 7:] - o_ref).max() < 1e-12


def test_analytic_backward_matches_autograd():
    B, S, H, D = 1, 96, 2, 16
    q, k, v, do = _data(B, S, H, D, 2)
    seg = np.zeros((B, S), np.int32)
    seg[:, 40:] = 1
    dq, dk, dv = R.dense_attention_bwd(q, k, v, do, causal=True, seg_q=seg, seg_k=seg)
    tq, tk, tv = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", tq, tk) / np.sqrt(D)
    vis = torch.from_numpy(R.visible_mask(S, S, causal=True, seg_q=seg, seg_k=seg, B=B))[:, None]
    p = torch.softmax(s.masked_fill(~vis, float("-inf")), dim=-1)
    out = torch.einsum("bhqk,bkhd->bqhd", p, tv)
    out.backward(torch.tensor(do, dtype=torch.float64))
    assert np.abs(dq - tq.grad.numpy()).max() < 1e-10
    assert np.abs(dk - tk.grad.numpy()).max() < 1e-10
    assert np.abs(dv - tv.grad.numpy()).max() < 1e-10


def test_blockwise_backward_equals_dense_backward():
    q, k, v, do = _data(1, 128, 2, 16, 3)
    ref = R.dense_attention_bwd(q, k, v, do, causal=True)
    got = R.blockwise_ring_attention_bwd(q, k, v, do, ring=2, q_chunk=32, k_chunk=16, causal=True)
    for a, b in zip(got, ref):
        assert np.abs(a - b).max() < 5e-5


def test_q_len_differs_from_kv_len_offsets():
    """prefill-into-cache shape (lwm/llama.py:485-487): q block at a global offset."""
    q, k, v, _ = _data(1, 160, 1, 16, 4)
    full, _ = R.dense_attention(q, k, v, causal=True)
    part, _ = R.dense_attention(q[:, 100:], k, v, causal=True, q_start=100, k_start=0)
    assert np.abs(full[:, 100:] - part).max() < 1e-12


def test_torch_cpu_port_matches_dense():
    from oracle.attention_torch_cpu import blockwise_fwd_bwd
    q, k, v, do = _data(1, 384, 2, 32, 5)
    o, dq, dk, dv = blockwise_fwd_bwd(*(torch.from_numpy(x) for x in (q, k, v, do)), q_chunk=128, k_chunk=64)
    ro, _ = R.dense_attention(q, k, v)
    rq, rk, rv = R.dense_attention_bwd(q, k, v, do)
    for a, b in ((o, ro), (dq, rq), (dk, rk), (dv, rv)):
        assert np.abs(a.numpy() - b).max() < 2e-5


def test_bf16_helpers_round_to_nearest_even():
    x = np.array([1.0, 1.00390625, 1.0078125, -3.1415926, 65504.0, 1e-40, np.inf], np.float32)
    got = R.round_bf16(x)
    ref = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(got, ref)
    rng = np.random.default_rng(0)
    y = rng.standard_normal(10000).astype(np.float32) * 100
    assert np.array_equal(R.round_bf16(y), torch.from_numpy(y).to(torch.bfloat16).float().numpy())


# ---- property tests (hypothesis): random shapes, chunkings, ring sizes, packings
from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=25, deadline=None)
@given(ring=st.sampled_from([1, 2, 4]), chunks=st.integers(1, 4), qc=st.sampled_from([8, 16, 32]),
       kc=st.sampled_from([8, 16, 32]), causal=st.booleans(), nseg=st.integers(1, 4), pad=st.booleans(),
       seed=st.integers(0, 2 ** 16))
def test_property_blockwise_ring_equals_dense(ring, chunks, qc, kc, causal, nseg, pad, seed):
    """For any ring size, chunking, document packing and key padding: the blockwise/ring restatement
    (forward AND backward) equals dense masked attention, and a packed batch equals its documents run
    one by one."""
    S = ring * chunks * 32
    rng = np.random.default_rng(seed)
    q, k, v, do = (rng.standard_normal((1, S, 1, 8)).astype(np.float32) for _ in range(4))
    cuts = np.sort(rng.choice(np.arange(1, S), size=nseg - 1, replace=False)) if nseg > 1 else np.array([], int)
    seg = np.searchsorted(cuts, np.arange(S), side="right").astype(np.int32)[None]
    kv = np.ones((1, S), np.uint8)
    if pad:
        kv[0, rng.choice(S, size=S // 8, replace=False)] = 0
    kw = dict(causal=causal, seg_q=seg, seg_k=seg, key_valid=kv)
    ref, _ = R.dense_attention(q, k, v, **kw)
    bkw = dict(causal=causal, segment_ids=seg, key_valid=kv)
    out = R.blockwise_ring_attention(q, k, v, ring=ring, q_chunk=qc, k_chunk=kc, **bkw)
    assert np.abs(out - ref).max() < 5e-5
    rq, rk, rv = R.dense_attention_bwd(q, k, v, do, **kw)
    bq, bk, bv = R.blockwise_ring_attention_bwd(q, k, v, do, ring=ring, q_chunk=qc, k_chunk=kc, **bkw)
    for a, b in ((bq, rq), (bk, rk), (bv, rv)):
        assert np.abs(a - b).max() < 2e-4 * max(1.0, np.abs(b).max())
    # packed == per document
    bounds = [0, *cuts.tolist(), S]
    for a, b in zip(bounds[:-1], bounds[1:]):
        sl = slice(a, b)
        one, _ = R.dense_attention(q[:, sl], k[:, sl], v[:, sl], causal=causal, key_valid=kv[:, sl])
        rows = kv[0, sl].astype(bool) if not causal else np.ones(b - a, bool)
        assert np.abs(one - ref[:, sl])[:, rows].max() < 1e-9


@pytest.mark.parametrize("case", ["causal", "dense", "packed+padding", "offset"])
def test_dense_oracle_matches_torch_sdpa(case):
    """An anchor that is not ours: PyTorch's own scaled_dot_product_attention (float64, math path) with the
    boolean mask of lwm/llama.py:572-592 spelled out by hand, and its autograd gradients, against the fp64 dense
    oracle and its analytic backward -- output, lse-free, dq, dk, dv."""
    B, S, H, D = 2, 96, 3, 16
    q, k, v, do = [a.astype(np.float64) for a in _data(B, S, H, D, 11)]
    kw = {}
    mask = np.ones((B, S, S), bool)
    ii, jj = np.arange(S)[:, None], np.arange(S)[None, :]
    if case in ("causal", "packed+padding"):
        kw["causal"] = True
        mask &= (jj <= ii)[None]
    else:
        kw["causal"] = False
    if case == "packed+padding":
        seg = np.zeros((B, S), np.int32)
        seg[:, 30:70] = 1
        seg[:, 70:] = 2
        valid = np.ones((B, S), np.uint8)
        valid[1, 90:] = 0
        kw.update(seg_q=seg, seg_k=seg, key_valid=valid)
        mask &= seg[:, :, None] == seg[:, None, :]
        mask &= valid[:, None, :] != 0
    if case == "offset":                       # a ring step: this q block sits 40 tokens after the k block
        kw.update(causal=True, q_start=40, k_start=0)
        mask &= (jj <= ii + 40)[None]
    tq, tk, tv = [torch.tensor(a.transpose(0, 2, 1, 3), requires_grad=True) for a in (q, k, v)]      # (B,H,S,D)
    out_t = torch.nn.functional.scaled_dot_product_attention(tq, tk, tv, attn_mask=torch.tensor(mask)[:, None])
    out_t.backward(torch.tensor(do.transpose(0, 2, 1, 3)))
    out, _ = R.dense_attention(q, k, v, **kw)
    dq, dk, dv = R.dense_attention_bwd(q, k, v, do, **kw)
    rows = mask.any(-1)                        # (a row with no visible key: the oracle returns 0, SDPA returns NaN)
    sel = np.broadcast_to(rows[:, :, None, None], out.shape)
    ref = out_t.detach().numpy().transpose(0, 2, 1, 3)
    assert np.abs(out - ref)[sel].max() < 1e-12
    for got, t in ((dq, tq), (dk, tk), (dv, tv)):
        g = np.nan_to_num(t.grad.numpy().transpose(0, 2, 1, 3))
        assert np.abs(got - g).max() < 1e-11, case


def test_the_reference_additive_bias_and_the_one_known_divergence():
    """SURVEY.md Appendix A.1 applies the masks as additive float32 biases (key padding: lwm/llama.py:533-537, 0 / finfo.min).
    (1) For every query row that sees at least one key this gives what the oracle's where-masking gives.  (2) A row that
    sees NONE -- the left-padded queries of lwm/vision_chat.py:136-140 -- comes out of the reference's arithmetic as the
    uniform average of V over the once-masked keys of the processed chunks, where the oracle (and the product) define
    out = 0, lse = -inf: the one place the outputs knowingly differ.  (3) Nothing downstream sees it: those rows are
    masked KEYS in every later layer, so a second attention layer over either version of the first layer's output is
    identical on every valid row."""
    rng = np.random.default_rng(5)
    B, S, H, D, pad, qc = 2, 96, 2, 16, 13, 32
    q, k, v = (R.round_bf16(rng.standard_normal((B, S, H, D)).astype(np.float32)) for _ in range(3))
    key_valid = np.ones((B, S), np.uint8)
    key_valid[:, :pad] = 0                                           # left padding
    o_where, lse_where = R.blockwise_ring_attention(q, k, v, ring=1, q_chunk=qc, k_chunk=qc, key_valid=key_valid, return_stats=True)
    o_add, _ = R.blockwise_ring_attention(q, k, v, ring=1, q_chunk=qc, k_chunk=qc, key_valid=key_valid, return_stats=True,
                                          additive_bias=True)
    assert np.array_equal(o_add[:, pad:], o_where[:, pad:])          # (1) rows with a visible key: the same, bit for bit
    assert np.all(o_where[:, :pad] == 0) and np.all(np.isneginf(lse_where[:, :, :pad]))
    # (2) row i < pad, one q chunk [0, qc): the chunk pairs not wholly above the diagonal are k chunk 0 only; in it the keys
    # j <= i violate padding alone, the keys pad <= j < qc violate causality alone (logit finfo.min each), the keys
    # i < j < pad violate both (-inf): the reference's value is the mean of V over the first kind and the second
    for i in (0, 5, pad - 1):
        once = np.r_[0:i + 1, pad:qc]
        assert np.allclose(o_add[:, i], v[:, once].mean(axis=1), rtol=1e-5, atol=1e-6)
        assert np.abs(o_add[:, i]).max() > 0
    # (3) a second layer (q = k = v = layer-1 output) under the same masks: identical on the valid rows
    h_ref, h_ours = R.round_bf16(o_add), R.round_bf16(o_where)
    o2_ref = R.blockwise_ring_attention(h_ref, h_ref, h_ref, ring=1, q_chunk=qc, k_chunk=qc, key_valid=key_valid, additive_bias=True)
    o2_ours = R.blockwise_ring_attention(h_ours, h_ours, h_ours, ring=1, q_chunk=qc, k_chunk=qc, key_valid=key_valid)
    assert np.array_equal(o2_ref[:, pad:], o2_ours[:, pad:])
    # the ring (2 ranks) agrees with one rank in both modes on the valid rows
    o_add2 = R.blockwise_ring_attention(q, k, v, ring=2, q_chunk=16, k_chunk=16, key_valid=key_valid, additive_bias=True)
    assert np.allclose(o_add2[:, pad:], o_where[:, pad:], rtol=2e-6, atol=2e-6)
