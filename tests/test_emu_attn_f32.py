"""The float32 flavour of the attention op (lwm_amd/csrc/attn_f32.h: lwm_attn_fwd_f32 / _bwd_delta_f32 / _bwd_dq_f32 /
_bwd_dkdv_f32 -- the reference's `--dtype=fp32`, lwm/train.py:36, BASELINE configs[0]) compiled for the host and run one
fiber per lane (tests/emu/), through the C ABI, against the fp64 oracle.  The bound is SURVEY.md section 8c's for an
fp32 kernel path: max|err| <= 1e-5 of the reference's maximum (the bf16 flavour's is 8e-3)."""
import ctypes as C

import numpy as np
import pytest

from lwm_amd import _capi
from oracle import attention_ref as R
from tests import _emu

TOL = 1e-5


def _rnd(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)


def _masks(B, S, Sk, seg, kv):
    rng = np.random.default_rng(7)
    seg_q = seg_k = key_valid = None
    if seg:
        s = np.zeros((B, S), np.int32)
        for c in np.sort(rng.choice(np.arange(1, S), size=3, replace=False)):
            s[:, c:] += 1
        seg_q = seg_k = s
    if kv:
        key_valid = (rng.random((B, Sk)) > 0.2).astype(np.uint8)
    return dict(seg_q=seg_q, seg_k=seg_k, key_valid=key_valid)


@pytest.mark.parametrize("B,Sq,Sk,H,causal,seg,kv", [
    (1, 256, 256, 1, True, False, False),
    (1, 320, 320, 2, True, False, False),     # ragged last workgroup, ragged last tile
    (2, 300, 300, 1, True, True, True),       # packed documents + padded keys
    (1, 100, 290, 1, False, False, True),     # q_len != kv_len
    (1, 1, 65, 1, False, False, False),
])
def test_emulated_f32_fwd_bwd(B, Sq, Sk, H, causal, seg, kv):
    q, k, v, do = _rnd((B, Sq, H, 128), 1), _rnd((B, Sk, H, 128), 2), _rnd((B, Sk, H, 128), 3), _rnd((B, Sq, H, 128), 4)
    kw = dict(causal=causal, **_masks(B, Sq, Sk, seg, kv))
    out, lse = _emu.attn_fwd_f32(q, k, v, **kw)
    ro, rl = R.dense_attention(q, k, v, **kw)
    assert _rel(out, ro) < TOL
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(lse), fin)
    assert np.abs(lse[fin] - rl[fin]).max() < 1e-5
    dq, dk, dv = _emu.attn_bwd_f32(q, k, v, out, lse, do, **kw)
    rq, rk, rv = R.dense_attention_bwd(q, k, v, do, **kw)[:3]
    assert _rel(dq, rq) < TOL and _rel(dk, rk) < TOL and _rel(dv, rv) < TOL


@pytest.mark.parametrize("Sq,Sk", [(0, 64), (64, 0), (1, 1), (33, 1), (129, 3)])
def test_emulated_f32_empty_and_degenerate_shapes(Sq, Sk):
    B, H = 1, 2
    q, k, v, do = _rnd((B, Sq, H, 128), 1), _rnd((B, Sk, H, 128), 2), _rnd((B, Sk, H, 128), 3), _rnd((B, Sq, H, 128), 4)
    out, lse = _emu.attn_fwd_f32(q, k, v, causal=False)
    if Sk == 0:
        assert not out.any() and np.isneginf(lse).all()
    elif Sq:
        ro, rl = R.dense_attention(q, k, v, causal=False)
        assert _rel(out, ro) < TOL and np.abs(lse - rl).max() < 1e-5
    dq, dk, dv = _emu.attn_bwd_f32(q, k, v, out, lse, do, causal=False)
    if Sq == 0 or Sk == 0:     # the side that exists gets exact zeros (the buffers were NaN before the call)
        assert not dq.any() and not dk.any() and not dv.any()
    else:
        rq, rk, rv = R.dense_attention_bwd(q, k, v, do, causal=False)[:3]
        near = lambda a, b: np.abs(a - b).max() <= TOL * max(np.abs(b).max(), 1.0)
        assert near(dq, rq) and near(dk, rk) and near(dv, rv)


def test_emulated_f32_ring_carries():
    """Two K/V blocks chained through the f32 carries (a 2-step ring on one query block) == one shot, forward and
    backward, at global position offsets."""
    B, S, H = 1, 160, 2
    q, k, v, do = (_rnd((B, S, H, 128), s) for s in (11, 12, 13, 14))
    k2, v2 = _rnd((B, S, H, 128), 15), _rnd((B, S, H, 128), 16)
    kf, vf = np.concatenate([k2, k], 1), np.concatenate([v2, v], 1)
    ro, rl = R.dense_attention(q, kf, vf, causal=True, q_start=S, k_start=0)
    acc = _emu.attn_fwd_f32(q, k, v, causal=True, q_start=S, k_start=S, final=False)
    out, lse = _emu.attn_fwd_f32(q, k2, v2, causal=True, q_start=S, k_start=0, carry=acc, final=True)
    assert _rel(out, ro) < TOL and np.abs(lse - rl).max() < 1e-5
    rq, rk, rv = R.dense_attention_bwd(q, kf, vf, do, causal=True, q_start=S, k_start=0)[:3]
    c0 = _emu.attn_bwd_f32(q, k, v, out, lse, do, causal=True, q_start=S, k_start=S, final=False)
    assert _rel(c0[1], rk[:, S:]) < TOL and _rel(c0[2], rv[:, S:]) < TOL
    zk, zv = _emu.aligned(c0[1].shape, np.float32), _emu.aligned(c0[2].shape, np.float32)
    c1 = _emu.attn_bwd_f32(q, k2, v2, out, lse, do, causal=True, q_start=S, k_start=0, carry=(c0[0], zk, zv), final=False)
    assert _rel(c1[0], rq) < TOL
    assert _rel(c1[1], rk[:, :S]) < TOL and _rel(c1[2], rv[:, :S]) < TOL
    # a carry into dk / dv (the block's gradients travelling on): added, not overwritten
    ck, cv = _emu.f32_array(_rnd(c0[1].shape, 17)), _emu.f32_array(_rnd(c0[2].shape, 18))
    base_k, base_v = ck.copy(), cv.copy()
    c2 = _emu.attn_bwd_f32(q, k2, v2, out, lse, do, causal=True, q_start=S, k_start=0,
                           carry=(_emu.f32_array(c0[0]), ck, cv), final=True)
    assert _rel(c2[1] - base_k, rk[:, :S]) < 5 * TOL and _rel(c2[2] - base_v, rv[:, :S]) < 5 * TOL


@pytest.mark.parametrize("q_start,k_start,Sq,Sk,causal", [
    (100, 37, 200, 290, True),       # the diagonal crosses the block at an offset that is no multiple of a tile
    (0, 64, 300, 260, True),         # keys start in the queries' future
    (512, 0, 70, 520, True),         # every key visible (an earlier ring block), ragged both ways
])
def test_emulated_f32_offsets_and_many_heads(q_start, k_start, Sq, Sk, causal):
    B, H = 1, 3
    q, k, v, do = _rnd((B, Sq, H, 128), 41), _rnd((B, Sk, H, 128), 42), _rnd((B, Sk, H, 128), 43), _rnd((B, Sq, H, 128), 44)
    kw = dict(causal=causal, q_start=q_start, k_start=k_start)
    out, lse = _emu.attn_fwd_f32(q, k, v, **kw)
    ro, rl = R.dense_attention(q, k, v, **kw)
    fin = np.isfinite(rl)
    assert _rel(out, ro) < TOL and np.array_equal(np.isfinite(lse), fin)
    got = _emu.attn_bwd_f32(q, k, v, out, lse, do, **kw)
    for a, ref in zip(got, R.dense_attention_bwd(q, k, v, do, **kw)[:3]):
        assert _rel(a, ref) < TOL


def test_emulated_f32_future_block_is_fully_masked():
    B, S, H = 1, 128, 1
    q, k, v = (_rnd((B, S, H, 128), s) for s in (21, 22, 23))
    out, lse = _emu.attn_fwd_f32(q, k, v, causal=True, q_start=0, k_start=4096)
    assert np.all(out == 0) and np.all(np.isneginf(lse))


def test_f32_entry_points_refuse_what_they_do_not_take():
    """Piecewise position maps, dense masks and split-K belong to the bf16 kernels: the f32 flavour says so instead of
    ignoring the fields."""
    L = _emu.lib()
    q = _emu.f32_array(_rnd((1, 256, 1, 128), 1))
    a, _ = _emu._base_args_f32(q, q, q, causal=True, q_start=0, k_start=0, seg_q=None, seg_k=None, key_valid=None, scale=None)
    out, lse = _emu.aligned(q.shape, np.float32), _emu.aligned((1, 1, 256), np.float32)
    a.out, a.lse, a.final_out = _emu._t4f(out), lse.ctypes.data, 1
    _capi.set_pieces(a, "k", [(0, 0), (256 - 128, 1024)])
    assert L.lwm_attn_fwd_f32(C.byref(a), None) == _capi.LWM_EUNSUPPORTED
    a.k_pieces = 0
    a.k_splits = 4
    assert L.lwm_attn_fwd_f32(C.byref(a), None) == _capi.LWM_EUNSUPPORTED
    a.k_splits = 0
    a.D = 64
    assert L.lwm_attn_fwd_f32(C.byref(a), None) == _capi.LWM_EUNSUPPORTED
    a.D = 128
    assert L.lwm_attn_fwd_f32(C.byref(a), None) == _capi.LWM_OK
