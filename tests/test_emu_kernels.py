"""The HIP kernel sources (lwm_amd/csrc/attn_*.h) compiled for the host and run
one fiber per lane (tests/emu/), through the same C ABI, against the fp64
oracle.  Confirms tile/fragment index math, masks, online softmax, carries and
ragged edges without a GPU; the lane maps the emulator assumes are themselves
confirmed on hardware by tests/test_gpu_probe.py."""
import numpy as np
import pytest

from oracle import attention_ref as R
from tests import _emu


def _rnd(shape, seed):
    return R.round_bf16(np.random.default_rng(seed).standard_normal(shape).astype(np.float32))


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
    (1, 320, 320, 2, True, False, False),     # ragged q tile + ragged kv tile
    (2, 300, 300, 1, True, True, True),
    (1, 100, 290, 1, False, False, True),     # q_len != kv_len
    (1, 1, 65, 1, False, False, False),
])
def test_emulated_fwd_bwd(B, Sq, Sk, H, causal, seg, kv):
    q, k, v, do = _rnd((B, Sq, H, 128), 1), _rnd((B, Sk, H, 128), 2), _rnd((B, Sk, H, 128), 3), \
        _rnd((B, Sq, H, 128), 4)
    kw = dict(causal=causal, **_masks(B, Sq, Sk, seg, kv))
    out, lse = _emu.attn_fwd(q, k, v, **kw)
    ro, rl = R.dense_attention(q, k, v, **kw)
    assert _rel(out, ro) < 1e-2
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(lse), fin)
    assert np.abs(lse[fin] - rl[fin]).max() < 1e-4
    dq, dk, dv = _emu.attn_bwd(q, k, v, out, lse, do, **kw)
    rq, rk, rv = R.dense_attention_bwd(q, k, v, do, **kw)
    assert _rel(dq, rq) < 1e-2 and _rel(dk, rk) < 1e-2 and _rel(dv, rv) < 1e-2


@pytest.mark.parametrize("Sq,Sk", [(0, 64), (64, 0), (1, 1), (33, 1), (257, 3)])
def test_emulated_empty_and_degenerate_shapes(Sq, Sk):
    """Empty query / key blocks (a rank whose shard sees nothing yet, an empty cache) and one-row
    blocks: no out-of-bounds access, out = 0 and lse = -inf where no key exists, zero gradients."""
    B, H = 1, 2
    q, k, v, do = _rnd((B, Sq, H, 128), 1), _rnd((B, Sk, H, 128), 2), _rnd((B, Sk, H, 128), 3), _rnd((B, Sq, H, 128), 4)
    out, lse = _emu.attn_fwd(q, k, v, causal=False)
    assert out.shape == (B, Sq, H, 128) and lse.shape == (B, H, Sq)
    if Sk == 0:
        assert not out.any() and np.isneginf(lse).all()
    elif Sq:
        ro, rl = R.dense_attention(q, k, v, causal=False)
        assert _rel(out, ro) < 1e-2 and np.abs(lse - rl).max() < 1e-4
    dq, dk, dv = _emu.attn_bwd(q, k, v, out, lse, do, causal=False)
    assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape
    if Sq == 0 or Sk == 0:
        assert not dq.any() and not dk.any() and not dv.any()
    else:
        rq, rk, rv = R.dense_attention_bwd(q, k, v, do, causal=False)
        # with one key p == 1 and dS == 0 exactly: dq, dk are pure rounding noise around a zero reference
        near = lambda a, b: np.abs(a - b).max() <= 1e-2 * max(np.abs(b).max(), 1.0)
        assert near(dq, rq) and near(dk, rk) and near(dv, rv)


def test_emulated_ring_carries():
    """two kv blocks with f32 carries (a 2-step ring on one q block) == one shot,
    forward and backward, with global position offsets."""
    B, S, H = 1, 256, 1
    q, k, v, do = (_rnd((B, S, H, 128), s) for s in (11, 12, 13, 14))
    # local queries are the SECOND half of a 512-token sequence; keys: both halves
    k2, v2 = _rnd((B, S, H, 128), 15), _rnd((B, S, H, 128), 16)
    kf, vf = np.concatenate([k2, k], 1), np.concatenate([v2, v], 1)
    ro, rl = R.dense_attention(q, kf, vf, causal=True, q_start=S, k_start=0)
    acc = _emu.attn_fwd(q, k, v, causal=True, q_start=S, k_start=S, final=False)   # step 0: own block
    out, lse = _emu.attn_fwd(q, k2, v2, causal=True, q_start=S, k_start=0, carry=acc, final=True)
    assert _rel(out, ro) < 1e-2 and np.abs(lse - rl).max() < 1e-4
    rq, rk, rv = R.dense_attention_bwd(q, kf, vf, do, causal=True, q_start=S, k_start=0)
    c0 = _emu.attn_bwd(q, k, v, out, lse, do, causal=True, q_start=S, k_start=S, final=False)
    dq_acc = c0[0]
    assert _rel(c0[1], rk[:, S:]) < 1e-2 and _rel(c0[2], rv[:, S:]) < 1e-2
    zk = _emu.aligned(c0[1].shape, np.float32)
    zv = _emu.aligned(c0[2].shape, np.float32)
    c1 = _emu.attn_bwd(q, k2, v2, out, lse, do, causal=True, q_start=S, k_start=0,
                       carry=(dq_acc, zk, zv), final=False)
    assert _rel(c1[0], rq) < 1e-2
    assert _rel(c1[1], rk[:, :S]) < 1e-2 and _rel(c1[2], rv[:, :S]) < 1e-2


@pytest.mark.parametrize("q_start,k_start,Sq,Sk,causal", [
    (100, 37, 200, 290, True),       # diagonal crosses the block at an offset that is no multiple of 32
    (0, 64, 300, 260, True),         # keys start in the queries' future: the first key blocks' early steps are skipped
    (512, 0, 70, 520, True),         # every key visible (an earlier ring block), ragged both ways
    (0, 0, 90, 300, False),
])
def test_emulated_backward_offsets_and_many_heads(q_start, k_start, Sq, Sk, causal):
    """The backward at position offsets that are not multiples of a tile / step, ragged both ways, with more
    (batch*head) slices than XCDs (10: the non-XCD-aware block mapping), against the oracle."""
    B, H = 1, 10
    q, k, v, do = _rnd((B, Sq, H, 128), 41), _rnd((B, Sk, H, 128), 42), _rnd((B, Sk, H, 128), 43), _rnd((B, Sq, H, 128), 44)
    kw = dict(causal=causal, q_start=q_start, k_start=k_start)
    out, lse = _emu.attn_fwd(q, k, v, **kw)
    rq, rk, rv = R.dense_attention_bwd(q, k, v, do, **kw)
    got = _emu.attn_bwd(q, k, v, out, lse, do, **kw)
    for a, ref in zip(got, (rq, rk, rv)):
        assert _rel(a, ref) < 1e-2


def test_emulated_future_block_is_fully_masked():
    B, S, H = 1, 128, 1
    q, k, v = (_rnd((B, S, H, 128), s) for s in (21, 22, 23))
    out, lse = _emu.attn_fwd(q, k, v, causal=True, q_start=0, k_start=4096)
    assert np.all(out == 0) and np.all(np.isneginf(lse))


def test_emulated_cast_f32_to_bf16():
    """lwm_cast_f32_to_bf16 (end of the backward ring): every element, ragged tail included."""
    import ctypes as C
    L = _emu.lib()
    for n in (8, 2048 + 8, 100003):
        x = _emu.aligned((n,), np.float32)
        x[...] = np.random.default_rng(n).standard_normal(n)
        y = _emu.aligned((n,), np.uint16)
        assert L.lwm_cast_f32_to_bf16(x.ctypes.data, y.ctypes.data, n, None) == 0
        assert np.array_equal(R.from_bf16_bits(y), R.round_bf16(x))


def test_emulated_sum_f32_to_bf16():
    """lwm_sum_f32_to_bf16 (owner-side reduction of returned dK/dV partials): f32 adds in
    argument order, one rounding to bf16, ragged tail included."""
    import ctypes as C
    L = _emu.lib()
    L.lwm_sum_f32_to_bf16.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    for n, k in ((8, 1), (2048 + 8, 3), (100003, 8)):
        xs = []
        for i in range(k):
            x = _emu.aligned((n,), np.float32)
            x[...] = np.random.default_rng(n + i).standard_normal(n) * 10.0 ** (i % 3)
            xs.append(x)
        y = _emu.aligned((n,), np.uint16)
        ptrs = (C.c_void_p * k)(*[x.ctypes.data for x in xs])
        assert L.lwm_sum_f32_to_bf16(ptrs, k, y.ctypes.data, n, None) == 0
        ref = xs[0].copy()
        for x in xs[1:]:
            ref = (ref + x).astype(np.float32)
        assert np.array_equal(R.from_bf16_bits(y), R.round_bf16(ref))
    assert L.lwm_sum_f32_to_bf16(ptrs, 17, y.ctypes.data, 8, None) != 0


@pytest.mark.parametrize("monotone", [True, False])
def test_emulated_packed_documents_are_skipped_not_changed(monotone):
    """Packed batch of several documents spanning many tiles: with the segment-block hints the
    kernels walk only their own documents' tiles; results equal the oracle and the hint-free run."""
    B, S, H = 1, 1024, 1
    q, k, v, do = (_rnd((B, S, H, 128), s) for s in (31, 32, 33, 34))
    seg = np.zeros((B, S), np.int32)
    for i, c in enumerate((130, 460, 465, 860)):
        seg[:, c:] = i + 1
    if not monotone:                       # ids are only compared for equality (lwm/llama.py:582-584)
        seg = np.array([7, 3, 9, 3, 1], np.int32)[seg]     # document 1 and 3 share an id
    kv = np.ones((B, S), np.uint8)
    kv[:, 450:480] = 0
    kw = dict(causal=True, seg_q=seg, seg_k=seg, key_valid=kv)
    ro, rl = R.dense_attention(q, k, v, **kw)
    rq, rk, rv = R.dense_attention_bwd(q, k, v, do, **kw)
    outs = {}
    for skip in (True, False):
        _emu.SEGMENT_SKIP = skip
        try:
            out, lse = _emu.attn_fwd(q, k, v, **kw)
            dq, dk, dv = _emu.attn_bwd(q, k, v, out, lse, do, **kw)
        finally:
            _emu.SEGMENT_SKIP = True
        outs[skip] = (out, lse, dq, dk, dv)
        assert _rel(out, ro) < 1e-2 and _rel(dq, rq) < 1e-2 and _rel(dk, rk) < 1e-2 and _rel(dv, rv) < 1e-2
        fin = np.isfinite(rl)
        assert np.array_equal(np.isfinite(lse), fin) and np.abs(lse[fin] - rl[fin]).max() < 1e-4
    for i, (a, b) in enumerate(zip(outs[True], outs[False])):
        assert np.array_equal(a, b)        # skipping removes only tiles that contribute exact zeros (the fused
                                           # backward's skipped partials are zeros in a fixed-order f32 sum)


def test_segment_block_table():
    import ctypes as C
    L = _emu.lib()
    seg = np.repeat(np.arange(5, dtype=np.int32), 30)[None]            # (1, 150)
    valid = np.ones((1, 150), np.uint8)
    valid[:, 64:128] = 0
    out = _emu.aligned((1, 5, 2), np.int32)
    assert L.lwm_attn_segment_blocks(seg.ctypes.data, valid.ctypes.data, out.ctypes.data, 1, 150, None) == 0
    assert out[0, 0].tolist() == [0, 1] and out[0, 1].tolist() == [1, 2]
    assert out[0, 2].tolist() == [2 ** 31 - 1, -2 ** 31] and out[0, 3].tolist() == out[0, 2].tolist()
    assert out[0, 4].tolist() == [4, 4]


# ---------------------------------------------------------------- two-piece position maps (lwm_version() >= 500)
def _embed(x, pos, n):
    """rows of x placed at their positions in a length-n sequence (zeros elsewhere)"""
    out = np.zeros((x.shape[0], n) + x.shape[2:], x.dtype)
    out[:, pos] = x
    return out


def _two_piece_case(Sq, q1, qg, Sk, k1, kg):
    """q rows [0,q1) at qg[0]+r, [q1,Sq) at qg[1]+(r-q1); likewise keys.  -> (positions, _emu keyword arguments)"""
    qpos = np.concatenate([qg[0] + np.arange(q1), qg[1] + np.arange(Sq - q1)])
    kpos = np.concatenate([kg[0] + np.arange(k1), kg[1] + np.arange(Sk - k1)])
    kw = dict(q_start=qg[0], k_start=kg[0])
    if q1 < Sq:
        kw["q_piece2"] = (q1, qg[1])
    if k1 < Sk:
        kw["k_piece2"] = (k1, kg[1])
    return qpos, kpos, kw


@pytest.mark.parametrize("name,Sq,q1,qg,Sk,k1,kg", [
    # a zigzag shard against itself (rank 1 of 4, half-chunks of 256): lo x lo diagonal, hi x lo full, hi x hi diagonal
    ("local", 512, 256, (256, 1536), 512, 256, (256, 1536)),
    # ... against what it gathered from its peers: [below its low half | between its halves]; the low queries see piece 1 only
    ("remote", 512, 256, (256, 1536), 1280, 256, (0, 512)),
    # one-piece queries (contiguous ownership) against two-piece keys, ragged ends on both sides
    ("ragged", 300, 300, (900, 0), 570, 256, (0, 700)),
    # two-piece queries, one-piece keys that end inside the first query piece's range
    ("qsplit", 512, 256, (0, 2048), 384, 384, (100, 0)),
])
def test_emulated_two_piece_position_maps(name, Sq, q1, qg, Sk, k1, kg):
    """One launch over a (two-piece q) x (two-piece k) block == dense attention on the positions the maps name: the
    operands embedded at their positions in the global sequence, absent keys masked out (fp64 oracle)."""
    H = 1
    q, k, v, do = _rnd((1, Sq, H, 128), 11), _rnd((1, Sk, H, 128), 12), _rnd((1, Sk, H, 128), 13), _rnd((1, Sq, H, 128), 14)
    qpos, kpos, kw = _two_piece_case(Sq, q1, qg, Sk, k1, kg)
    out, lse = _emu.attn_fwd(q, k, v, causal=True, **kw)
    n = int(max(qpos.max(), kpos.max())) + 1
    present = np.zeros((1, n), np.uint8)
    present[:, kpos] = 1
    ro, rl = R.dense_attention(_embed(q, qpos, n), _embed(k, kpos, n), _embed(v, kpos, n), causal=True, key_valid=present)
    ro, rl = ro[:, qpos], rl[:, :, qpos]
    assert _rel(out, ro) < 1e-2
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(lse), fin) and np.abs(lse[fin] - rl[fin]).max() < 1e-4
    assert not out[~np.isfinite(lse).transpose(0, 2, 1)].any()          # rows that see no key: out = 0
    dq, dk, dv = _emu.attn_bwd(q, k, v, out, lse, do, causal=True, **kw)
    rq, rk, rv = R.dense_attention_bwd(_embed(q, qpos, n), _embed(k, kpos, n), _embed(v, kpos, n), _embed(do, qpos, n),
                                       causal=True, key_valid=present)
    assert _rel(dq, rq[:, qpos]) < 1e-2 and _rel(dk, rk[:, kpos]) < 1e-2 and _rel(dv, rv[:, kpos]) < 1e-2


def test_emulated_four_piece_maps():
    """a balanced ownership of a packed batch: four chunks of 256 rows per rank (q), three runs of fetched chunks (k),
    packed documents on top -- against the oracle on the embedded operands"""
    H = 1
    qcuts, kcuts = [(256, 1024), (512, 2304), (768, 3840)], [(512, 1280), (1024, 2560)]
    Sq, Sk = 1024, 1792
    q, k, v, do = _rnd((1, Sq, H, 128), 31), _rnd((1, Sk, H, 128), 32), _rnd((1, Sk, H, 128), 33), _rnd((1, Sq, H, 128), 34)
    rows = lambda start, cuts, S: np.concatenate([p + np.arange(r1 - r0) for (r0, p), r1 in zip([(0, start)] + cuts, [c[0] for c in cuts] + [S])])
    qpos, kpos = rows(256, qcuts, Sq), rows(0, kcuts, Sk)
    n = int(max(qpos.max(), kpos.max())) + 1
    seg_full = (np.arange(n) >= 700).astype(np.int32) + (np.arange(n) >= 2400).astype(np.int32)
    kw = dict(causal=True, q_start=256, k_start=0, q_piece2=qcuts, k_piece2=kcuts, seg_q=seg_full[qpos][None], seg_k=seg_full[kpos][None])
    out, lse = _emu.attn_fwd(q, k, v, **kw)
    present = np.zeros((1, n), np.uint8)
    present[:, kpos] = 1
    okw = dict(causal=True, key_valid=present, seg_q=seg_full[None], seg_k=seg_full[None])
    ro, rl = R.dense_attention(_embed(q, qpos, n), _embed(k, kpos, n), _embed(v, kpos, n), **okw)
    assert _rel(out, ro[:, qpos]) < 1e-2
    fin = np.isfinite(rl[:, :, qpos])
    assert np.array_equal(np.isfinite(lse), fin) and np.abs(lse[fin] - rl[:, :, qpos][fin]).max() < 1e-4
    dq, dk, dv = _emu.attn_bwd(q, k, v, out, lse, do, **kw)
    rq, rk, rv = R.dense_attention_bwd(_embed(q, qpos, n), _embed(k, kpos, n), _embed(v, kpos, n), _embed(do, qpos, n), **okw)
    assert _rel(dq, rq[:, qpos]) < 1e-2 and _rel(dk, rk[:, kpos]) < 1e-2 and _rel(dv, rv[:, kpos]) < 1e-2


def test_emulated_adjacent_pieces_are_the_single_piece_launch_bit_for_bit():
    """two pieces that happen to be adjacent (start2 = start + split) name the same positions as one piece: same tiles,
    same instruction stream, same bits -- forward, dq, dk, dv, with a packed batch on top"""
    Sq = Sk = 768
    q, k, v, do = _rnd((1, Sq, 1, 128), 21), _rnd((1, Sk, 1, 128), 22), _rnd((1, Sk, 1, 128), 23), _rnd((1, Sq, 1, 128), 24)
    seg = np.zeros((1, Sq), np.int32)
    seg[:, 300:] = 1
    m = dict(seg_q=seg, seg_k=seg)
    one = _emu.attn_fwd(q, k, v, causal=True, q_start=4096, k_start=4096, **m)
    two = _emu.attn_fwd(q, k, v, causal=True, q_start=4096, k_start=4096, q_piece2=(256, 4096 + 256), k_piece2=(512, 4096 + 512), **m)
    assert np.array_equal(one[0], two[0]) and np.array_equal(one[1], two[1])
    g1 = _emu.attn_bwd(q, k, v, one[0], one[1], do, causal=True, q_start=4096, k_start=4096, **m)
    g2 = _emu.attn_bwd(q, k, v, one[0], one[1], do, causal=True, q_start=4096, k_start=4096, q_piece2=(512, 4096 + 512),
                       k_piece2=(256, 4096 + 256), **m)
    for a, b in zip(g1, g2):
        assert np.array_equal(a, b)


def test_two_piece_maps_are_validated():
    import ctypes as C
    from lwm_amd import _capi
    L = _emu.lib()
    q = _emu.bf16_array(np.zeros((1, 512, 1, 128), np.float32))
    out = _emu.aligned((1, 512, 1, 128), np.uint16)
    lse = _emu.aligned((1, 1, 512), np.float32)
    for bad in (dict(q_piece2=(100, 4096)), dict(q_piece2=(512, 4096)), dict(k_piece2=(256, 100)), dict(q_piece2=(256, 255)),
                dict(q_piece2=[(256, 300), (256, 600)]), dict(k_piece2=[(256, 1000), (384, 900)])):
        a, _ = _emu.base_args(q, q, q, causal=True, q_start=0, k_start=0, seg_q=None, seg_k=None, key_valid=None, scale=None, **bad)
        a.out, a.lse, a.final_out = _emu._t4(out), lse.ctypes.data, 1
        assert L.lwm_attn_fwd(C.byref(a), None) == _capi.LWM_EINVAL, bad
        assert b"piece" in L.lwm_last_error()
    # the backward refuses a statistics buffer of the pre-400 size when told how large it is
    a, _ = _emu.base_args(q, q, q, causal=True, q_start=0, k_start=0, seg_q=None, seg_k=None, key_valid=None, scale=None)
    small = _emu.aligned((512,), np.float32)
    a.out = a.dout = a.dq = _emu._t4(out)
    a.lse, a.delta, a.delta_bytes, a.final_out = lse.ctypes.data, small.ctypes.data, small.nbytes, 1
    assert L.lwm_attn_bwd_delta(C.byref(a), None) == _capi.LWM_EINVAL and b"lwm_attn_bwd_delta_bytes" in L.lwm_last_error()
    assert L.lwm_attn_bwd_dq(C.byref(a), None) == _capi.LWM_EINVAL
