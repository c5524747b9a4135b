"""RoPE and RMSNorm kernels (lwm_amd/csrc/llama_elem.h) emulated on the host through
the C ABI against the numpy oracle (oracle/llama_ops_ref.py)."""
import numpy as np
import pytest

from lwm_amd.llama_ops import precompute_freqs_cis
from oracle import llama_ops_ref as R
from oracle.attention_ref import round_bf16
from tests import _emu


def _rnd(shape, seed, scale=1.0):
    return round_bf16((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize("theta,max_pos", [(10000.0, 4096), (5e7, 1 << 18)])   # (1M positions: tests/test_gpu_llama_ops.py)
def test_rope_table_and_rotation(theta, max_pos):
    B, S, H, D = 2, 37, 3, 128
    x = _rnd((B, S, H, D), 1)
    pos = np.random.default_rng(2).integers(0, max_pos, (B, S)).astype(np.int32)
    pos[0, 0], pos[0, 1] = 0, max_pos - 1
    fc = R.precompute_freqs_cis(D, max_pos, theta)
    tab = precompute_freqs_cis(D, max_pos, theta).numpy()
    # the product's host table is the reference's complex table, split into (cos, sin)
    assert np.array_equal(tab[..., 0], fc.real) and np.array_equal(tab[..., 1], fc.imag)
    got = _emu.rope(x, tab, pos)
    ref = R.apply_rotary_emb(x, fc, pos)
    assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(ref).max()      # one bf16 ulp at most
    assert np.mean(got != ref) < 1e-3                                  # (fma vs mul+sub rounding)
    # backward = conjugate rotation; rotation is orthogonal: round trip restores x
    back = _emu.rope(got, tab, pos, conj=True)
    assert np.abs(back - x).max() <= 2e-2 * np.abs(x).max()
    gref = R.rope_bwd(x, fc, pos)
    assert np.abs(_emu.rope(x, tab, pos, conj=True) - gref).max() <= 2 ** -7 * np.abs(gref).max()


@pytest.mark.parametrize("rows,C", [(5, 4096), (3, 256), (2, 8192), (7, 1000 // 8 * 8)])
def test_rmsnorm_fwd_bwd(rows, C):
    x = _rnd((rows, C), 3, 2.0)
    w = round_bf16((1 + 0.1 * np.random.default_rng(4).standard_normal(C)).astype(np.float32))
    g = _rnd((rows, C), 5)
    y, rstd = _emu.rmsnorm_fwd(x, w)
    ref = R.rmsnorm(x, w)
    assert np.abs(y - ref).max() <= 2 ** -7 * np.abs(ref).max()
    assert np.mean(y != ref) < 5e-3
    r64 = 1.0 / np.sqrt(np.mean(x.astype(np.float64) ** 2, axis=-1) + 1e-6)
    assert np.abs(rstd - r64).max() <= 1e-6 * r64.max()
    dx, dw = _emu.rmsnorm_bwd(x, w, g, rstd)
    rdx, rdw = R.rmsnorm_bwd(x, w, g)
    assert np.abs(dx - rdx).max() <= 1e-2 * np.abs(rdx).max()
    assert np.abs(dw - rdw).max() <= 1e-2 * max(np.abs(rdw).max(), 1e-6)


def test_validation():
    import ctypes as C
    from lwm_amd import _capi
    L = _emu.lib()
    assert L.lwm_rmsnorm_fwd_bf16(16, 16, 16, None, 4, 100, 1e-6, None) == _capi.LWM_EUNSUPPORTED
    t = _capi.LwmTensor4(None, 0, 0, 0)
    assert L.lwm_rope_bf16(t, t, None, None, 1, 1, 1, 128, 16, 0, None) == _capi.LWM_EINVAL


@pytest.mark.parametrize("B,S,V", [(2, 5, 32000), (1, 3, 8448), (2, 4, 64)])
def test_softmax_cross_entropy(B, S, V):
    g = np.random.default_rng(9)
    logits = round_bf16((g.standard_normal((B, S, V)) * 3).astype(np.float32))
    tokens = g.integers(0, V, (B, S))
    tokens[0, 0] = int(logits[0, 0].argmax())              # one guaranteed-correct prediction
    valid = (g.random((B, S)) > 0.3).astype(np.float32)
    valid[0, 0] = 1
    loss, acc, dref = R.cross_entropy_loss_and_accuracy(logits, tokens, valid)
    w = valid / (np.maximum(valid.sum(-1, keepdims=True), 1e-10) * B)
    nll, cor, dl = _emu.softmax_ce(logits.reshape(-1, V), tokens.reshape(-1), w.reshape(-1))
    assert abs(float((nll * w.reshape(-1)).sum()) - loss) <= 1e-5 * max(1.0, abs(loss))
    assert abs(float((cor * w.reshape(-1)).sum()) - acc) <= 1e-6
    assert cor[0] == 1
    assert np.abs(dl.reshape(B, S, V) - dref).max() <= 2 ** -8 * np.abs(dref).max() + 1e-9


@pytest.mark.parametrize("rows,K,N", [(1, 256, 512), (2, 96, 40), (4, 160, 1032), (3, 128, 8)])
def test_gemv_decode_projection(rows, K, N):
    """lwm_gemv_bf16 (the projections of a cached-decode step): ragged K tile (K % 128 != 0), ragged and tiny N
    tiles, 1..4 rows; f32 accumulation, so the f32 result equals the exact product to f32 rounding and the bf16
    result is its rounding."""
    x, w = _rnd((rows, K), 21), _rnd((K, N), 22, 0.1)
    yb, yf = _emu.gemv(x, w, want_f32=True)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    assert np.abs(yf - ref).max() <= 2e-6 * np.abs(ref).max() * np.sqrt(K)
    assert np.array_equal(yb, round_bf16(yf))
    assert np.array_equal(_emu.gemv(x, w), yb)                      # bf16-only output: same values


def test_gemv_multi_shares_x():
    """wq | wk | wv (or w1 | w3) in one launch pair: each output equals the single-matrix call bit for bit."""
    x = _rnd((2, 160), 31)
    ws = [_rnd((160, n), 32 + i, 0.1) for i, n in enumerate((520, 64, 1032))]
    got = _emu.gemv_multi(x, ws, want_f32=True)
    for w, g in zip(ws, got):
        assert np.array_equal(g, _emu.gemv(x, w, want_f32=True)[1])
    gb = _emu.gemv_multi(x, ws[:2])
    assert np.array_equal(gb[0], _emu.gemv(x, ws[0])) and np.array_equal(gb[1], _emu.gemv(x, ws[1]))


def test_gemv_fused_neighbours_equal_the_separate_launches():
    """lwm_gemv_fused_bf16: RMSNorm on load (rstd from partial sums of squares), residual add in the reduction and the
    partial sums of squares of the result, against rmsnorm_fwd -> gemv -> bf16 add run one by one: identical bits when
    rstd is identical (one partial = the kernel's own total), and the partials sum to the row's sum of squares."""
    rng = np.random.default_rng(5)
    rows, K, N = 2, 256, 384
    x = R.round_bf16(rng.standard_normal((rows, K)).astype(np.float32))
    gam = R.round_bf16((1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    w1, w2 = (R.round_bf16((rng.standard_normal((K, n)) * 0.05).astype(np.float32)) for n in (N, 128))
    res = R.round_bf16(rng.standard_normal((rows, N)).astype(np.float32))
    eps = 1e-6
    xn, rstd = _emu.rmsnorm_fwd(x, gam, eps)
    # partials whose tree sum reproduces the standalone kernel's total exactly: the total itself + zeros
    tot = (1.0 / rstd.astype(np.float64) ** 2 - eps) * K
    ss = np.zeros((rows, 32), np.float32)
    ss[:, 0] = tot.astype(np.float32)
    (y1, y2) = _emu.gemv_fused(x, [w1, w2], norm=(ss, gam, eps))
    r1, r2 = _emu.gemv(xn, w1), _emu.gemv(xn, w2)
    rstd_f = 1.0 / np.sqrt(ss.sum(1) / np.float32(K) + np.float32(eps), dtype=np.float32)
    if np.array_equal(rstd_f, rstd):        # (the f64 -> f32 round trip of the total may move rstd by an ulp)
        assert np.array_equal(y1, r1) and np.array_equal(y2, r2)
    else:
        assert np.abs(y1 - r1).max() <= 2.0 ** -7 * np.abs(r1).max()
    # residual + partial sums of squares
    (z,), sso = _emu.gemv_fused(x, [w1], residual=res, want_ss=True)
    want = R.round_bf16(_emu.gemv(x, w1) + res)
    assert np.array_equal(z, want)
    assert sso.shape == (rows, N // 128)
    assert np.allclose(sso.sum(1), (want.astype(np.float64) ** 2).sum(1), rtol=1e-5)
    # the two together: a whole "x + wo(attn)" -> "norm -> w1" hand-off
    w3 = R.round_bf16((rng.standard_normal((N, 128)) * 0.05).astype(np.float32))
    ones = np.ones(N, np.float32)
    (q,) = _emu.gemv_fused(z, [w3], norm=(sso, ones, eps))
    zn, _ = _emu.rmsnorm_fwd(z, ones, eps)
    ref = _emu.gemv(zn, w3)
    assert np.abs(q - ref).max() <= 2.0 ** -7 * np.abs(ref).max()       # (rstd: other summation order, at most an ulp)


def test_gemv_validation():
    import ctypes as C
    L = _emu.lib()
    buf = _emu.aligned((4096,), np.float32)
    p = buf.ctypes.data
    assert L.lwm_gemv_bf16(p, 64, p, p, 64, None, p, 5, 64, 64, None) == -2       # rows > 4
    assert L.lwm_gemv_bf16(p, 64, p, p, 60, None, p, 1, 64, 60, None) == -2       # N % 8
    assert L.lwm_gemv_bf16(p, 64, p, None, 64, None, p, 1, 64, 64, None) == -1    # no output
    assert L.lwm_gemv_workspace_bytes(2, 4096, 11008) == 32 * 2 * 11008 * 4


# ---- round 6: the entry points the re-laid training GEMMs lean on (lwm_amd/llama_ops.py, "library GEMMs of the TRAINING path")
@pytest.mark.parametrize("R_,C_,pad_s,pad_d", [(64, 64, 0, 0), (128, 192, 8, 16), (256, 64, 24, 0)])
def test_transpose_is_exact(R_, C_, pad_s, pad_d):
    from oracle.attention_ref import to_bf16_bits
    L = _emu.lib()
    src = _emu.aligned((R_, C_ + pad_s), np.uint16)
    src[...] = to_bf16_bits(_rnd((R_, C_ + pad_s), 11))
    dst = _emu.aligned((C_, R_ + pad_d), np.uint16)
    dst[...] = 0x7fc0                                                   # (what the kernel must not leave / must not touch)
    assert L.lwm_transpose_bf16(src.ctypes.data, C_ + pad_s, dst.ctypes.data, R_ + pad_d, R_, C_, None) == 0
    assert np.array_equal(dst[:, :R_], src[:, :C_].T)
    assert np.all(dst[:, R_:] == 0x7fc0)                                # the padding of the destination rows stays
    from lwm_amd import _capi
    assert L.lwm_transpose_bf16(src.ctypes.data, C_ + pad_s, dst.ctypes.data, R_ + pad_d, R_ - 1, C_, None) == _capi.LWM_EUNSUPPORTED
    assert L.lwm_transpose_bf16(src.ctypes.data, C_ - 8, dst.ctypes.data, R_ + pad_d, R_, C_, None) == _capi.LWM_EINVAL


@pytest.mark.parametrize("rows,F", [(16, 64), (5, 11008 // 8), (1, 8)])
def test_swiglu_on_halves_equals_the_flat_kernels(rows, F):
    """gate | up as the halves of one (rows, 2F) buffer, d gate | d up into the halves of another: bit for bit the flat
    kernels on contiguous copies (lwm/llama.py:659)."""
    from oracle.attention_ref import to_bf16_bits
    L = _emu.lib()
    y13 = _emu.aligned((rows, 2 * F), np.uint16)
    y13[...] = to_bf16_bits(_rnd((rows, 2 * F), 12, 2.0))
    g = _emu.aligned((rows, F), np.uint16)
    g[...] = to_bf16_bits(_rnd((rows, F), 13))
    a, b = _emu.aligned((rows, F), np.uint16), _emu.aligned((rows, F), np.uint16)
    a[...], b[...] = y13[:, :F], y13[:, F:]
    y0, y1, da0, db0 = (_emu.aligned((rows, F), np.uint16) for _ in range(4))
    d13 = _emu.aligned((rows, 2 * F), np.uint16)
    assert L.lwm_swiglu_fwd_bf16(a.ctypes.data, b.ctypes.data, y0.ctypes.data, rows * F, None) == 0
    assert L.lwm_swiglu_fwd_ld_bf16(y13.ctypes.data, 2 * F, y13.ctypes.data + 2 * F, 2 * F, y1.ctypes.data, F, rows, F, None) == 0
    assert np.array_equal(y0, y1)
    assert L.lwm_swiglu_bwd_bf16(a.ctypes.data, b.ctypes.data, g.ctypes.data, da0.ctypes.data, db0.ctypes.data, rows * F, None) == 0
    assert L.lwm_swiglu_bwd_ld_bf16(y13.ctypes.data, 2 * F, y13.ctypes.data + 2 * F, 2 * F, g.ctypes.data, F, d13.ctypes.data, 2 * F,
                                    d13.ctypes.data + 2 * F, 2 * F, rows, F, None) == 0
    assert np.array_equal(d13[:, :F], da0) and np.array_equal(d13[:, F:], db0)
    # against the oracle's SwiGLU as well
    from oracle.attention_ref import from_bf16_bits
    ref = R.swiglu(from_bf16_bits(a), from_bf16_bits(b)) if hasattr(R, "swiglu") else None
    if ref is not None:
        assert np.abs(from_bf16_bits(y1) - ref).max() <= 2 ** -6 * np.abs(ref).max()


@pytest.mark.parametrize("rows,C", [(300, 256), (2049, 64), (5, 4096)])
def test_rmsnorm_bwd_with_the_residual_gradient_folded_in(rows, C):
    """dx = bf16(bf16(dx_norm) + res): the roundings of autograd's separate add (FlaxLLaMABlock's `x` feeds the norm and the
    residual add, lwm/llama.py:704-744); dw unchanged; and the re-worked dW reduction against a float64 sum."""
    from oracle.attention_ref import from_bf16_bits, to_bf16_bits
    L = _emu.lib()
    x, g, res = _rnd((rows, C), 21, 2.0), _rnd((rows, C), 22), _rnd((rows, C), 23)
    w = round_bf16((1 + 0.1 * np.random.default_rng(24).standard_normal(C)).astype(np.float32))
    _, rstd = _emu.rmsnorm_fwd(x, w)
    dx0, dw0 = _emu.rmsnorm_bwd(x, w, g, rstd)
    xb, wb, gb, rb = (_emu.bf16_array(t) for t in (x, w, g, res))
    dx1, dw1 = _emu.aligned((rows, C), np.uint16), _emu.aligned((C,), np.uint16)
    ws = _emu.aligned((max(L.lwm_rmsnorm_bwd_workspace_bytes(rows, C), 16) // 4,), np.float32)
    r = _emu.aligned((rows,), np.float32)
    r[...] = rstd
    assert L.lwm_rmsnorm_bwd_res_bf16(xb.ctypes.data, wb.ctypes.data, gb.ctypes.data, r.ctypes.data, rb.ctypes.data, dx1.ctypes.data,
                                      dw1.ctypes.data, ws.ctypes.data, rows, C, None) == 0
    assert np.array_equal(dx1, to_bf16_bits(dx0 + res))
    assert np.array_equal(from_bf16_bits(dw1), dw0)
    dwr = (g.astype(np.float64) * x.astype(np.float64) * rstd[:, None].astype(np.float64)).sum(0)
    assert np.abs(dw0 - dwr).max() <= 6e-3 * np.abs(dwr).max()          # (one bf16 rounding of the result)


# ---- round 6: the hand-written weight-gradient GEMM (lwm_amd/csrc/gemm_wgrad.h)
def _wgrad_emu(x, g, K, N, S, pad_x=0, pad_g=0, pad_w=0):
    from oracle.attention_ref import from_bf16_bits, to_bf16_bits
    L = _emu.lib()
    xb = _emu.aligned((S, K + pad_x), np.uint16)
    gb = _emu.aligned((S, N + pad_g), np.uint16)
    xb[...] = 0x7fc0
    gb[...] = 0x7fc0                                                      # (padding columns must never be read into a sum)
    xb[:, :K], gb[:, :N] = to_bf16_bits(x), to_bf16_bits(g)
    dw = _emu.aligned((K, N + pad_w), np.uint16)
    dw[...] = 0x7fc0
    nbytes = L.lwm_wgrad_workspace_bytes(S, K, N)
    ws = _emu.aligned((max(nbytes, 16) // 4,), np.float32)
    ws[...] = np.nan                                                      # (a partial nobody wrote must not be summed)
    rc = L.lwm_wgrad_bf16(xb.ctypes.data, K + pad_x, gb.ctypes.data, N + pad_g, dw.ctypes.data, N + pad_w, S, K, N,
                          ws.ctypes.data if nbytes else None, nbytes, None)
    assert rc == 0, _emu.last_error() if hasattr(_emu, "last_error") else rc
    assert np.all(dw[:, N:] == 0x7fc0)
    return from_bf16_bits(dw[:, :N]), nbytes


@pytest.mark.parametrize("S,K,N,cus,pads", [
    (160, 256, 256, 24, (0, 0, 0)),        # one tile cut into five one-stage ranges: five partials summed
    (32, 256, 512, 24, (8, 16, 24)),       # a single stage per tile (prologue only), padded leading dimensions
    (224, 1024, 2816, 24, (0, 0, 0)),      # 44 tiles on 24 "CUs": 24 whole, 20 stream-K with ranges that span two tiles; a ragged band
    (96, 512, 3328, 24, (0, 8, 0)),        # 26 tiles: two left over, cut into 1-stage ranges
    (64, 4096, 4096, 256, (0, 0, 0)),      # 256 tiles on 256 CUs: the XCD-blocked tile numbering, no stream-K
])
@pytest.mark.parametrize("waves", [8, 4])
def test_wgrad_gemm(S, K, N, cus, pads, waves, monkeypatch):
    """dW = x^T g with f32 accumulation (the flax Dense kernel's gradient, lwm/llama.py:390-421): products of bf16 values are
    exact in f32, so only the ORDER of the f32 sums differs from the oracle -- bound = bf16 rounding of the result plus f32
    summation noise."""
    monkeypatch.setenv("LWM_EMU_CUS", str(cus))
    monkeypatch.setenv("LWM_WGRAD_WAVES", str(waves))          # 8: two waves per SIMD (default); 4: one wave per SIMD, 128 x 128 each
    x, g = _rnd((S, K), 21), _rnd((S, N), 22)
    got, nbytes = _wgrad_emu(x, g, K, N, S, *pads)
    tiles = (K // 256) * (N // 256)
    assert (nbytes > 0) == (tiles % cus != 0) and nbytes in (0, 2 * cus * 256 * 256 * 4)
    ref = x.astype(np.float64).T @ g.astype(np.float64)
    assert np.abs(got - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-3
    assert np.mean(got != round_bf16(ref.astype(np.float32))) < 2e-3      # (a rounding tie now and then)


def test_wgrad_validation():
    from lwm_amd import _capi
    L = _emu.lib()
    a = _emu.aligned((64, 256), np.uint16)
    d = _emu.aligned((256, 256), np.uint16)
    ws = _emu.aligned((2 * 24 * 65536,), np.float32)
    args = lambda **kw: [kw.get("x", a.ctypes.data), kw.get("ldx", 256), a.ctypes.data, 256, d.ctypes.data, 256,
                         kw.get("S", 64), kw.get("K", 256), kw.get("N", 256), kw.get("ws", ws.ctypes.data), kw.get("wsb", ws.nbytes), None]
    assert L.lwm_wgrad_bf16(*args()) == 0
    assert L.lwm_wgrad_bf16(*args(S=48)) == _capi.LWM_EUNSUPPORTED
    assert L.lwm_wgrad_bf16(*args(K=128)) == _capi.LWM_EUNSUPPORTED
    assert L.lwm_wgrad_bf16(*args(ldx=248)) == _capi.LWM_EINVAL
    assert L.lwm_wgrad_bf16(*args(ws=None)) == _capi.LWM_EINVAL
    assert L.lwm_wgrad_bf16(*args(wsb=1024)) == _capi.LWM_EINVAL
    assert L.lwm_wgrad_workspace_bytes(64, 256, 100) == 0
