"""CPU oracle for the RingAttention hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under lwm_amd/ does.

PARITY UNPINNED: the reference's arithmetic for this path lives in the
un-vendored, un-pinned pip package `ringattention`
(git+https://github.com/haoliuhl/ringattention.git, gpu_requirements.txt:8,
imported at lwm/llama.py:30) and the reference ships no tests or golden vectors
(SURVEY.md section 4, 8c).  jax/flax are not installable here, so the reference
cannot be executed to generate fixtures.  This module therefore restates

  (a) the in-tree *specification* of the mask -- the dense branch of
      FlaxLLaMAAttention, lwm/llama.py:572-592 (causal AND same-segment AND
      key-valid) with the bias constants of lwm/llama.py:527-537 -- as a dense
      float64 softmax attention, and
  (b) the published blockwise/ring algorithm the call site asks for
      (lwm/llama.py:539-569: float32_logits=True, causal_block_size=1,
      query/key chunk sizes, axis "sp"): ring loop over kv blocks, scan over q
      chunks x k chunks with (numerator, denominator, max) carry, skip of chunk
      pairs wholly above the diagonal, out = numerator/denominator; and the
      custom-VJP backward (recompute p from the saved statistics, rotate
      k,v,dk,dv) -- SURVEY.md Appendix A.1,

(an independent executable anchor exists one level up: HF transformers' LlamaForCausalLM --
the PyTorch route the reference documents, scripts/sample_pyt.py:8 -- is reproduced by the
oracle MODEL, whose attention is checked against dense_attention() below; tests/test_weights.py.
That pins the dense semantics, not the JAX package's blockwise order of operations)

and anchors parity on the structural identities the reference's own code
implies: blockwise == dense branch, ring n == ring 1, packed == per-segment.

PINNED since round 5: the MASK.  tests/golden/gen_ref_run_golden.py executes the reference's own mask statements
(lwm/llama.py:425, :527-537, :572-592, where they lie, numpy standing in for jax.numpy) and commits the visibility they
produce (tests/golden/ref_run.npz); visible_mask() / decode_mask() below reproduce it pair by pair
(tests/test_golden.py::test_mask_oracle_reproduces_the_reference_run).  The softmax-attention ARITHMETIC stays unpinned.

Layouts follow the reference: q,k,v,out are (B, S, H, D) (heads split by
reshape, lwm/llama.py:434-438); segment_ids (B, S); key-padding mask (B, S).
Rows with no visible key are defined to give out = 0, lse = -inf (the reference
returns a uniform average of masked keys there; such rows are left-padding
queries whose outputs are never used, lwm/vision_chat.py:138-140).
"""
from __future__ import annotations

import numpy as np

NEG_INF = -np.inf


def visible_mask(Sq, Sk, *, causal=True, q_start=0, k_start=0, seg_q=None, seg_k=None,
                 key_valid=None, B=1, dense_mask=None):
    """Boolean (B, Sq, Sk): the dense mask of lwm/llama.py:577-592 on global positions."""
    vis = np.ones((B, Sq, Sk), dtype=bool)
    if causal:
        qp = q_start + np.arange(Sq)[:, None]
        kp = k_start + np.arange(Sk)[None, :]
        vis &= (kp <= qp)[None]
    if seg_q is not None:
        vis &= (np.asarray(seg_q)[:, :, None] == np.asarray(seg_k)[:, None, :])
    if key_valid is not None:
        vis &= (np.asarray(key_valid)[:, None, :] != 0)
    if dense_mask is not None:      # arbitrary (B,Sq,Sk) boolean mask: ringattention_inference,
        vis &= (np.asarray(dense_mask) != 0)   # lwm/llama.py:577-614
    return vis


def _hm(x):
    """(B, S, H, D) -> (B, H, S, D) view"""
    return x.transpose(0, 2, 1, 3)


def _scores(a, b):
    """einsum("bqhd,bkhd->bhqk") through the BLAS (np.einsum's own loops are ~20x slower on long key axes)"""
    return np.matmul(_hm(a), _hm(b).transpose(0, 1, 3, 2))


def _apply(p, x):
    """einsum("bhqk,bkhd->bqhd")"""
    return np.matmul(p, _hm(x)).transpose(0, 2, 1, 3)


def _apply_t(p, x):
    """einsum("bhqk,bqhd->bkhd")"""
    return np.matmul(p.transpose(0, 1, 3, 2), _hm(x)).transpose(0, 2, 1, 3)


def dense_attention(q, k, v, *, causal=True, q_start=0, k_start=0, seg_q=None, seg_k=None,
                    key_valid=None, scale=None, dtype=np.float64, dense_mask=None):
    """Dense masked softmax attention.  Returns (out (B,Sq,H,D), lse (B,H,Sq))."""
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    if scale is None:
        scale = 1.0 / np.sqrt(D)
    s = _scores(q, k) * dtype(scale)
    vis = visible_mask(Sq, Sk, causal=causal, q_start=q_start, k_start=k_start, seg_q=seg_q,
                       seg_k=seg_k, key_valid=key_valid, B=B, dense_mask=dense_mask)[:, None]
    s = np.where(vis, s, NEG_INF)
    m = s.max(axis=-1, keepdims=True) if Sk > 0 else np.full(s.shape[:-1] + (1,), NEG_INF)
    m_safe = np.where(np.isfinite(m), m, 0.0)
    p = np.exp(s - m_safe)
    l = p.sum(axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        pn = np.where(l > 0, p / np.where(l > 0, l, 1.0), 0.0)
        lse = np.where(l[..., 0] > 0, m_safe[..., 0] + np.log(np.where(l[..., 0] > 0, l[..., 0], 1.0)),
                       NEG_INF)
    out = _apply(pn, v)
    return out, lse


def dense_attention_bwd(q, k, v, dout, *, causal=True, q_start=0, k_start=0, seg_q=None,
                        seg_k=None, key_valid=None, scale=None, dtype=np.float64, out_saved=None, dense_mask=None):
    """Analytic gradients of dense_attention w.r.t. q, k, v (float64 by default): (dq, dk, dv).

    out_saved: the forward output AS THE IMPLEMENTATION SAVED IT (bf16, the reference saves `out` cast to v.dtype for
    its custom VJP too -- SURVEY.md Appendix A.1).  With it the call returns (dq_saved, dk, dv, dq): dq_saved takes
    delta = rowsum(dout * out_saved), i.e. the residual every implementation of this VJP really has, so that a dq row
    can be checked against its own scale with no allowance for the rounding of `out` (in rows that see few keys
    dp - delta cancels almost completely and that rounding is the whole gradient); dq, dk, dv differentiate the exact
    function."""
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    dout = np.asarray(dout, dtype=dtype)
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    if scale is None:
        scale = 1.0 / np.sqrt(D)
    out, lse = dense_attention(q, k, v, causal=causal, q_start=q_start, k_start=k_start,
                               seg_q=seg_q, seg_k=seg_k, key_valid=key_valid, scale=scale,
                               dtype=dtype, dense_mask=dense_mask)
    s = _scores(q, k) * dtype(scale)
    vis = visible_mask(Sq, Sk, causal=causal, q_start=q_start, k_start=k_start, seg_q=seg_q,
                       seg_k=seg_k, key_valid=key_valid, B=B, dense_mask=dense_mask)[:, None]
    lse_safe = np.where(np.isfinite(lse), lse, 0.0)[..., None]
    p = np.where(vis & np.isfinite(lse)[..., None], np.exp(np.where(vis, s, 0.0) - lse_safe), 0.0)
    dv = _apply_t(p, dout)
    dp = _scores(dout, v)
    delta = np.einsum("bqhd,bqhd->bhq", dout, out)[..., None]
    ds = p * (dp - delta) * dtype(scale)
    dq = _apply(ds, k)
    dk = _apply_t(ds, q)
    if out_saved is None:
        return dq, dk, dv
    delta_s = np.einsum("bqhd,bqhd->bhq", dout, np.asarray(out_saved, dtype=dtype))[..., None]
    # dq_saved = dq - scale * (delta_saved - delta) * (P K): no second pass over the scores
    pk = _apply(p, k)
    dq_saved = dq - dtype(scale) * np.moveaxis(delta_s - delta, 1, 2) * pk
    return dq_saved, dk, dv, dq


def decode_mask(B, Q, K, cache_index, attention_mask=None):
    """The mask the reference builds for cached decoding (lwm/llama.py:574-592):
    key k visible to query i iff k <= i + cache_index, AND attention_mask[b,k]."""
    m = (np.arange(K)[None, :] <= (np.arange(Q)[:, None] + cache_index))[None].repeat(B, 0)
    if attention_mask is not None:
        m = m & (np.asarray(attention_mask)[:, None, :] != 0)
    return m


def ring_inference(q, k, v, mask, *, ring=1, scale=None):
    """ringattention_inference restated: the K/V cache is
    sharded contiguously over `ring` devices, every device holds all queries, one
    un-chunked online-softmax update per ring step with the boolean mask sliced to
    the block; float32.  Returns out (B,Q,H,D)."""
    q = np.asarray(q, np.float32)
    k = np.asarray(k, np.float32)
    v = np.asarray(v, np.float32)
    B, Q, H, D = q.shape
    K = k.shape[1]
    c = K // ring
    if scale is None:
        scale = 1.0 / np.sqrt(D)
    num = np.zeros((B, Q, H, D), np.float32)
    den = np.zeros((B, H, Q), np.float32)
    mx = np.full((B, H, Q), NEG_INF, np.float32)
    fmin = np.finfo(np.float32).min
    for t in range(ring):
        sl = slice(t * c, (t + 1) * c)
        s = np.einsum("bqhd,bkhd->bhqk", q, k[:, sl]) * np.float32(scale)
        vis = (np.asarray(mask)[:, :, sl] != 0)[:, None]
        s = np.where(vis, s, fmin)
        m_new = np.maximum(mx, s.max(axis=-1))
        p = np.where(vis, np.exp(s - m_new[..., None]), np.float32(0))
        corr = np.where(np.isfinite(mx), np.exp(mx - m_new), np.float32(0))
        num = num * np.transpose(corr, (0, 2, 1))[..., None] + np.einsum("bhqk,bkhd->bqhd", p, v[:, sl])
        den = den * corr + p.sum(axis=-1)
        mx = np.where(m_new <= fmin / 2, NEG_INF, m_new)
    den_t = np.transpose(den, (0, 2, 1))[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(den_t > 0, num / np.where(den_t > 0, den_t, 1), 0)


# --------------------------------------------------------------------------
# Blockwise / ring restatement (float32, the reference's float32_logits path)
# --------------------------------------------------------------------------

def _chunk_bias(vis):
    """{0, finfo(float32).min} additive bias, as lwm/llama.py:533-537 builds it."""
    return np.where(vis, np.float32(0.0), np.finfo(np.float32).min).astype(np.float32)


def blockwise_ring_attention(q, k, v, *, ring=1, q_chunk=1024, k_chunk=1024, causal=True,
                             segment_ids=None, key_valid=None, scale=None,
                             return_stats=False, additive_bias=False):
    """Forward of the ring/blockwise algorithm, simulated for `ring` devices.

    additive_bias=True: the masks act ONLY as the reference applies them -- additive float32 biases, key padding
    (lwm/llama.py:533-537: 0 / finfo.min) + segment + causal terms, summed in float32 (two
    violated conditions overflow to -inf) -- with no special case for rows that never see a key.  For every row with at
    least one visible key the result is the default mode's (exp(finfo.min - max) underflows to 0); a row with NONE -- a
    left-padded query, lwm/vision_chat.py:136-140 -- comes out as the UNIFORM AVERAGE of V over the keys of the processed
    chunks that violate exactly one condition (all their logits equal finfo.min: |q.k| is far below its ulp), where the
    default mode -- and the product -- define out = 0, lse = -inf.  This is the one place where the product's output is
    knowingly not the reference's; tests/test_oracle_ring.py and tests/test_gpu_attention.py state both values.

    q,k,v are the GLOBAL (B,S,H,D) arrays; rank r owns rows [r*c,(r+1)*c)
    (contiguous sharding, lwm/llama.py:560-562).  For ring step t, rank r holds
    kv block (r - t) mod n; each (q chunk, k chunk) pair wholly above the
    diagonal is skipped; otherwise
        s = q.k/sqrt(D) (f32) + bias;  m' = max(m, rowmax s);  p = exp(s - m')
        numerator = numerator*exp(m-m') + p.v;  denominator likewise.
    Output numerator/denominator.  Everything is float32.
    """
    q = np.asarray(q, dtype=np.float32)
    k = np.asarray(k, dtype=np.float32)
    v = np.asarray(v, dtype=np.float32)
    B, S, H, D = q.shape
    assert S % ring == 0
    c = S // ring
    qc = min(q_chunk, c)
    kc = min(k_chunk, c)
    assert c % qc == 0 and c % kc == 0
    if scale is None:
        scale = 1.0 / np.sqrt(D)
    scale = np.float32(scale)
    out = np.zeros_like(q)
    lse = np.full((B, H, S), NEG_INF, dtype=np.float32)
    fmin = np.finfo(np.float32).min
    for r in range(ring):
        num = np.zeros((B, c, H, D), np.float32)
        den = np.zeros((B, H, c), np.float32)
        mx = np.full((B, H, c), NEG_INF, np.float32)
        for t in range(ring):
            kb = (r - t) % ring
            for qi in range(c // qc):
                q0 = r * c + qi * qc
                qs = q[:, q0:q0 + qc]
                for ki in range(c // kc):
                    k0 = kb * c + ki * kc
                    if causal and k0 > q0 + qc - 1:
                        continue  # wholly above the diagonal (causal_block_size=1)
                    ks = k[:, k0:k0 + kc]
                    vs = v[:, k0:k0 + kc]
                    s = np.einsum("bqhd,bkhd->bhqk", qs, ks) * scale
                    vis = visible_mask(
                        qc, kc, causal=causal, q_start=q0, k_start=k0,
                        seg_q=None if segment_ids is None else np.asarray(segment_ids)[:, q0:q0 + qc],
                        seg_k=None if segment_ids is None else np.asarray(segment_ids)[:, k0:k0 + kc],
                        key_valid=None if key_valid is None else np.asarray(key_valid)[:, k0:k0 + kc],
                        B=B)[:, None]
                    sl = slice(qi * qc, qi * qc + qc)
                    m_old = mx[:, :, sl]
                    if additive_bias:
                        terms = [visible_mask(qc, kc, causal=causal, q_start=q0, k_start=k0, B=B)[:, None]]
                        if segment_ids is not None:
                            sg = np.asarray(segment_ids)
                            terms.append((sg[:, q0:q0 + qc, None] == sg[:, None, k0:k0 + kc])[:, None])
                        if key_valid is not None:
                            terms.append((np.asarray(key_valid)[:, None, k0:k0 + kc] != 0)[:, None])
                        with np.errstate(over="ignore"):
                            for tm in terms:
                                s = s + _chunk_bias(tm)
                        m_new = np.maximum(m_old, s.max(axis=-1))
                        with np.errstate(invalid="ignore"):
                            p = np.where(np.isfinite(m_new)[..., None], np.exp(s - m_new[..., None]), np.float32(0.0))
                            corr = np.where(np.isfinite(m_old), np.exp(m_old - m_new), np.float32(0.0))
                        num[:, sl] = num[:, sl] * np.transpose(corr, (0, 2, 1))[..., None] + \
                            np.einsum("bhqk,bkhd->bqhd", p, vs)
                        den[:, :, sl] = den[:, :, sl] * corr + p.sum(axis=-1)
                        mx[:, :, sl] = m_new
                        continue
                    s = s + _chunk_bias(vis)
                    m_new = np.maximum(m_old, s.max(axis=-1))
                    p = np.exp(s - m_new[..., None])
                    # a chunk whose every entry carries the finfo.min bias must not
                    # contribute (in the reference exp(min - max) underflows to 0 as
                    # soon as the row has seen one visible key; rows that never do
                    # are defined as 0 here)
                    p = np.where(vis, p, np.float32(0.0))
                    corr = np.exp(np.where(np.isfinite(m_old), m_old - m_new, NEG_INF))
                    corr = np.where(np.isfinite(m_new), corr, np.float32(0.0))
                    num[:, sl] = num[:, sl] * np.transpose(corr, (0, 2, 1))[..., None] + \
                        np.einsum("bhqk,bkhd->bqhd", p, vs)
                    den[:, :, sl] = den[:, :, sl] * corr + p.sum(axis=-1)
                    # rows whose running max is still the finfo.min bias have seen no key
                    mx[:, :, sl] = np.where(m_new <= fmin / 2, NEG_INF, m_new)
        den_t = np.transpose(den, (0, 2, 1))[..., None]
        with np.errstate(divide="ignore", invalid="ignore"):
            out[:, r * c:(r + 1) * c] = np.where(den_t > 0, num / np.where(den_t > 0, den_t, 1), 0)
            lse[:, :, r * c:(r + 1) * c] = np.where(
                den > 0, mx + np.log(np.where(den > 0, den, 1)), NEG_INF)
    if return_stats:
        return out, lse
    return out


def blockwise_ring_attention_bwd(q, k, v, dout, *, ring=1, q_chunk=1024, k_chunk=1024,
                                 causal=True, segment_ids=None, key_valid=None, scale=None):
    """Backward of the ring/blockwise algorithm (custom VJP restated, float32).

    Residuals: out and the softmax statistics (as lse).  Per (q chunk, k chunk):
    p = exp(s - lse); dv += p^T do; dp = do v^T; ds = p*(dp - rowsum(do*out))*scale;
    dq += ds k; dk += ds^T q.  dk, dv accumulate in float32 and rotate with k, v.
    """
    q = np.asarray(q, dtype=np.float32)
    k = np.asarray(k, dtype=np.float32)
    v = np.asarray(v, dtype=np.float32)
    dout = np.asarray(dout, dtype=np.float32)
    B, S, H, D = q.shape
    c = S // ring
    qc = min(q_chunk, c)
    kc = min(k_chunk, c)
    if scale is None:
        scale = 1.0 / np.sqrt(D)
    scale = np.float32(scale)
    out, lse = blockwise_ring_attention(q, k, v, ring=ring, q_chunk=q_chunk, k_chunk=k_chunk,
                                        causal=causal, segment_ids=segment_ids,
                                        key_valid=key_valid, scale=scale, return_stats=True)
    delta = np.einsum("bqhd,bqhd->bhq", dout, out)
    dq = np.zeros_like(q)
    dk = np.zeros_like(k)
    dv = np.zeros_like(v)
    for r in range(ring):
        for t in range(ring):
            kb = (r - t) % ring
            for qi in range(c // qc):
                q0 = r * c + qi * qc
                for ki in range(c // kc):
                    k0 = kb * c + ki * kc
                    if causal and k0 > q0 + qc - 1:
                        continue
                    qs, dos = q[:, q0:q0 + qc], dout[:, q0:q0 + qc]
                    ks, vs = k[:, k0:k0 + kc], v[:, k0:k0 + kc]
                    s = np.einsum("bqhd,bkhd->bhqk", qs, ks) * scale
                    vis = visible_mask(
                        qc, kc, causal=causal, q_start=q0, k_start=k0,
                        seg_q=None if segment_ids is None else np.asarray(segment_ids)[:, q0:q0 + qc],
                        seg_k=None if segment_ids is None else np.asarray(segment_ids)[:, k0:k0 + kc],
                        key_valid=None if key_valid is None else np.asarray(key_valid)[:, k0:k0 + kc],
                        B=B)[:, None]
                    ls = lse[:, :, q0:q0 + qc][..., None]
                    ok = vis & np.isfinite(ls)
                    p = np.where(ok, np.exp(np.where(ok, s - np.where(np.isfinite(ls), ls, 0), 0)),
                                 np.float32(0)).astype(np.float32)
                    dv[:, k0:k0 + kc] += np.einsum("bhqk,bqhd->bkhd", p, dos)
                    dp = np.einsum("bqhd,bkhd->bhqk", dos, vs)
                    ds = p * (dp - delta[:, :, q0:q0 + qc][..., None]) * scale
                    dq[:, q0:q0 + qc] += np.einsum("bhqk,bkhd->bqhd", ds, ks)
                    dk[:, k0:k0 + kc] += np.einsum("bhqk,bqhd->bkhd", ds, qs)
    return dq, dk, dv


# --------------------------------------------------------------------------
# bf16 helpers (numpy has no bfloat16): round-to-nearest-even on the top 16 bits
# --------------------------------------------------------------------------

def to_bf16_bits(x):
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    u = x.view(np.uint32)
    rounded = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    # NaN stays NaN
    nan = np.isnan(x)
    rounded = np.where(nan, np.uint32(0x7FC0), rounded)
    return rounded.astype(np.uint16)


def from_bf16_bits(b):
    b = np.asarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def round_bf16(x):
    return from_bf16_bits(to_bf16_bits(x))
