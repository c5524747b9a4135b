"""The float32 flavour of the HBM-bound kernels (lwm_amd/csrc/elem_f32.h: RoPE, RMSNorm, SwiGLU, softmax cross entropy,
the ordered sum -- the reference's `--dtype=fp32`) emulated on the host through the C ABI against the numpy oracle
(oracle/llama_ops_ref.py).  Nothing is rounded to bf16 on the way, so the bounds are f32 rounding bounds."""
import ctypes as C

import numpy as np
import pytest

from lwm_amd import _capi
from lwm_amd.llama_ops import precompute_freqs_cis
from oracle import llama_ops_ref as R
from tests import _emu


def _rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def _a(x, dtype=np.float32):
    a = _emu.aligned(np.shape(x), dtype)
    a[...] = x
    return a


def _ok(L, rc, what):
    _capi.check(L, rc, what)


def _rope(x, tab, pos, conj=False):
    L = _emu.lib()
    xa, ta, pa = _a(x), _a(tab), _a(pos, np.int32)
    y = _emu.aligned(xa.shape, np.float32)
    B, S, H, D = xa.shape
    _ok(L, L.lwm_rope_f32(_emu._t4f(xa), _emu._t4f(y), ta.ctypes.data, pa.ctypes.data, B, S, H, D, ta.shape[0], int(conj), None),
        "lwm_rope_f32")
    return y


@pytest.mark.parametrize("theta,max_pos", [(10000.0, 4096), (5e7, 1 << 18)])
def test_rope_f32(theta, max_pos):
    B, S, H, D = 2, 37, 3, 128
    x = _rnd((B, S, H, D), 1)
    pos = np.random.default_rng(2).integers(0, max_pos, (B, S)).astype(np.int32)
    pos[0, 0], pos[0, 1] = 0, max_pos - 1
    fc = R.precompute_freqs_cis(D, max_pos, theta)
    tab = precompute_freqs_cis(D, max_pos, theta).numpy()
    got = _rope(x, tab, pos)
    ref = R.apply_rotary_emb(x, fc, pos, out_bf16=False)
    assert np.abs(got - ref).max() <= 4e-7 * np.abs(ref).max()
    back = _rope(got, tab, pos, conj=True)               # the rotation is orthogonal
    assert np.abs(back - x).max() <= 1e-6 * np.abs(x).max()
    gref = R.rope_bwd(x, fc, pos)
    assert np.abs(_rope(x, tab, pos, conj=True) - gref).max() <= 4e-7 * np.abs(gref).max()


@pytest.mark.parametrize("rows,C_", [(5, 4096), (3, 256), (2, 8192), (7, 1000)])
def test_rmsnorm_f32_fwd_bwd(rows, C_):
    L = _emu.lib()
    x, g = _a(_rnd((rows, C_), 3, 2.0)), _a(_rnd((rows, C_), 5))
    w = _a((1 + 0.1 * np.random.default_rng(4).standard_normal(C_)).astype(np.float32))
    y, rstd = _emu.aligned((rows, C_), np.float32), _emu.aligned((rows,), np.float32)
    _ok(L, L.lwm_rmsnorm_fwd_f32(x.ctypes.data, w.ctypes.data, y.ctypes.data, rstd.ctypes.data, rows, C_, 1e-6, None), "rmsnorm_fwd_f32")
    ref = R.rmsnorm(x, w, out_bf16=False)
    assert np.abs(y - ref).max() <= 1e-6 * np.abs(ref).max()
    r64 = 1.0 / np.sqrt(np.mean(x.astype(np.float64) ** 2, axis=-1) + 1e-6)
    assert np.abs(rstd - r64).max() <= 1e-6 * r64.max()
    dx, dw = _emu.aligned((rows, C_), np.float32), _emu.aligned((C_,), np.float32)
    dw[...] = np.nan
    ws = _emu.aligned((max(L.lwm_rmsnorm_bwd_workspace_bytes(rows, C_), 16) // 4,), np.float32)
    _ok(L, L.lwm_rmsnorm_bwd_f32(x.ctypes.data, w.ctypes.data, g.ctypes.data, rstd.ctypes.data, dx.ctypes.data, dw.ctypes.data,
                                 ws.ctypes.data, rows, C_, None), "rmsnorm_bwd_f32")
    rdx, rdw = R.rmsnorm_bwd(x, w, g)
    assert np.abs(dx - rdx).max() <= 2e-6 * np.abs(rdx).max()
    assert np.abs(dw - rdw).max() <= 2e-6 * max(np.abs(rdw).max(), 1e-6)


def test_swiglu_f32():
    L = _emu.lib()
    n = 4 * 1000
    a, b, g = _a(_rnd((n,), 1, 3.0)), _a(_rnd((n,), 2)), _a(_rnd((n,), 3))
    y, da, db = (_emu.aligned((n,), np.float32) for _ in range(3))
    _ok(L, L.lwm_swiglu_fwd_f32(a.ctypes.data, b.ctypes.data, y.ctypes.data, n, None), "swiglu_fwd_f32")
    ref = R.swiglu(a, b)
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max()
    _ok(L, L.lwm_swiglu_bwd_f32(a.ctypes.data, b.ctypes.data, g.ctypes.data, da.ctypes.data, db.ctypes.data, n, None), "swiglu_bwd_f32")
    ra, rb = R.swiglu_bwd(a, b, g)
    assert np.abs(da - ra).max() <= 2e-6 * np.abs(ra).max() and np.abs(db - rb).max() <= 2e-6 * np.abs(rb).max()
    assert L.lwm_swiglu_fwd_f32(a.ctypes.data, b.ctypes.data, y.ctypes.data, 6, None) == _capi.LWM_EINVAL


@pytest.mark.parametrize("B,S,V", [(2, 5, 32000), (1, 3, 8448), (2, 4, 64)])
def test_softmax_cross_entropy_f32(B, S, V):
    L = _emu.lib()
    g = np.random.default_rng(9)
    logits = _a((g.standard_normal((B, S, V)) * 3).astype(np.float32))
    tokens = g.integers(0, V, (B, S))
    tokens[0, 0] = int(logits[0, 0].argmax())
    valid = (g.random((B, S)) > 0.3).astype(np.float32)
    valid[0, 0] = 1
    loss, acc, dref = R.cross_entropy_loss_and_accuracy(logits, tokens, valid)
    w = _a((valid / (np.maximum(valid.sum(-1, keepdims=True), 1e-10) * B)).reshape(-1))
    tg = _a(tokens.reshape(-1), np.int32)
    nll, cor, dl = _emu.aligned((B * S,), np.float32), _emu.aligned((B * S,), np.int32), _emu.aligned((B * S, V), np.float32)
    _ok(L, L.lwm_softmax_ce_f32(logits.ctypes.data, tg.ctypes.data, w.ctypes.data, nll.ctypes.data, cor.ctypes.data, dl.ctypes.data,
                                B * S, V, None), "softmax_ce_f32")
    assert abs(float((nll * w).sum()) - loss) <= 2e-6 * max(1.0, abs(loss))
    assert abs(float((cor * w).sum()) - acc) <= 1e-6 and cor[0] == 1
    assert np.abs(dl.reshape(B, S, V) - dref).max() <= 2e-6 * np.abs(dref).max() + 1e-12
    assert L.lwm_softmax_ce_f32(logits.ctypes.data, tg.ctypes.data, None, nll.ctypes.data, None, None, 1, 40000, None) == \
        _capi.LWM_EUNSUPPORTED


def test_sum_f32_is_the_ordered_sum():
    L = _emu.lib()
    n = 4 * 777
    srcs = [_a(_rnd((n,), s, 10.0 ** (s - 2))) for s in range(5)]
    dst = _emu.aligned((n,), np.float32)
    ptrs = (C.c_void_p * len(srcs))(*[s.ctypes.data for s in srcs])
    _ok(L, L.lwm_sum_f32(ptrs, len(srcs), dst.ctypes.data, n, None), "lwm_sum_f32")
    ref = srcs[0].copy()
    for s in srcs[1:]:
        ref = ref + s                  # f32 adds in argument order, as the kernel
    assert np.array_equal(dst, ref)
    one = _emu.aligned((n,), np.float32)
    _ok(L, L.lwm_sum_f32(ptrs, 1, one.ctypes.data, n, None), "lwm_sum_f32")
    assert np.array_equal(one, srcs[0])
