"""RoPE / RMSNorm HIP kernels on MI355X through the reference-named Python surface
(lwm_amd.llama_ops), forward and backward, against the numpy oracle.
Tolerance: bf16 outputs within one bf16 ulp (2^-7 relative) of the oracle's bf16 result;
gradients within 1e-2 relative of the float64 gradients."""
import numpy as np
import pytest

from oracle import llama_ops_ref as R
from oracle.attention_ref import round_bf16

pytestmark = pytest.mark.gpu


def _rnd(shape, seed, scale=1.0):
    return round_bf16((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _dev(a, dt=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dt) if dt is not None else t


def _np(t):
    return t.detach().float().cpu().numpy()


@pytest.mark.parametrize("theta,max_pos,S", [(10000.0, 4096, 333), (5e7, 1 << 20, 1024)])
def test_rope_fwd_bwd(theta, max_pos, S):
    import torch
    from lwm_amd.llama_ops import apply_rotary_emb, precompute_freqs_cis
    B, H, D = 2, 4, 128
    xq, xk, gq = _rnd((B, S, H, D), 1), _rnd((B, S, H, D), 2), _rnd((B, S, H, D), 3)
    pos = np.random.default_rng(4).integers(0, max_pos, (B, S)).astype(np.int32)
    tab = precompute_freqs_cis(D, max_pos, theta, device="cuda")
    fc = R.precompute_freqs_cis(D, max_pos, theta)
    q = _dev(xq, torch.bfloat16).requires_grad_(True)
    k = _dev(xk, torch.bfloat16)
    oq, ok = apply_rotary_emb(q, k, tab, _dev(pos))
    rq, rk = R.apply_rotary_emb(xq, fc, pos), R.apply_rotary_emb(xk, fc, pos)
    for got, ref in ((oq, rq), (ok, rk)):
        assert np.abs(_np(got) - ref).max() <= 2 ** -7 * np.abs(ref).max()
    oq.backward(_dev(gq, torch.bfloat16))
    rg = R.rope_bwd(gq, fc, pos)
    assert np.abs(_np(q.grad) - rg).max() <= 2 ** -7 * np.abs(rg).max()
    # default positions = arange (lwm/llama.py:1081-1082)
    o2, _ = apply_rotary_emb(k, k, tab)
    r2 = R.apply_rotary_emb(xk, fc, np.tile(np.arange(S), (B, 1)))
    assert np.abs(_np(o2) - r2).max() <= 2 ** -7 * np.abs(r2).max()


@pytest.mark.parametrize("shape", [(2, 100, 4096), (1, 7, 256), (3, 8192)])
def test_rmsnorm_module_fwd_bwd(shape):
    import torch
    from lwm_amd.llama_ops import RMSNorm
    C = shape[-1]
    x, g = _rnd(shape, 5, 2.0), _rnd(shape, 6)
    w = (1 + 0.1 * np.random.default_rng(7).standard_normal(C)).astype(np.float32)
    m = RMSNorm(C).cuda()
    with torch.no_grad():
        m.kernel.copy_(torch.from_numpy(w))
    xd = _dev(x, torch.bfloat16).requires_grad_(True)
    y = m(xd)
    ref = R.rmsnorm(x, w)
    assert np.abs(_np(y) - ref).max() <= 2 ** -7 * np.abs(ref).max()
    y.backward(_dev(g, torch.bfloat16))
    rdx, rdw = R.rmsnorm_bwd(x, round_bf16(w), g)
    assert np.abs(_np(xd.grad) - rdx).max() <= 1e-2 * np.abs(rdx).max()
    assert np.abs(_np(m.kernel.grad) - rdw).max() <= 1e-2 * np.abs(rdw).max()


def test_full_size_bandwidth_and_properties():
    """LWM-7B shapes at S = 32768: RoPE over (1,S,32,128), RMSNorm over (S,4096)."""
    import torch
    from lwm_amd.llama_ops import RMSNorm, apply_rotary_emb, precompute_freqs_cis
    S = 32768
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(1, S, 32, 128, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    tab = precompute_freqs_cis(128, S, 1e7, device="cuda")
    y, _ = apply_rotary_emb(x, x, tab)
    # the rotation preserves the norm of every (even, odd) pair up to bf16 rounding
    n0 = (x.float() ** 2).reshape(1, S, 32, 64, 2).sum(-1)
    n1 = (y.float() ** 2).reshape(1, S, 32, 64, 2).sum(-1)
    assert ((n0 - n1).abs() <= 2e-2 * n0 + 1e-6).all()
    assert torch.equal(y[:, 0], x[:, 0])             # position 0: identity
    h = x.reshape(S, 4096)
    out = RMSNorm(4096).cuda()(h)
    rms = out.float().pow(2).mean(-1).sqrt()
    assert (rms - 1).abs().max().item() <= 1e-2      # weight = ones: unit RMS rows


def test_cross_entropy_and_chunked_head():
    """tux.cross_entropy_loss_and_accuracy semantics (lwm/train.py:177-181) + the chunked lm_head:
    loss / accuracy / gradients against the float64 oracle; chunked == unchunked."""
    import torch
    from lwm_amd.llama_ops import chunked_lm_head_loss, cross_entropy_loss_and_accuracy
    B, S, Dm, V = 2, 96, 256, 32000
    g = np.random.default_rng(11)
    h = round_bf16((g.standard_normal((B, S, Dm))).astype(np.float32))
    W = round_bf16((g.standard_normal((Dm, V)) * 0.05).astype(np.float32))
    tokens = g.integers(0, V, (B, S))
    valid = (g.random((B, S)) > 0.25).astype(np.float32)
    hd = _dev(h, torch.bfloat16).requires_grad_(True)
    Wd = _dev(W, torch.bfloat16).requires_grad_(True)
    logits = hd @ Wd                                   # bf16 logits, as the model's lm_head gives
    lg = logits.detach().clone().requires_grad_(True)
    loss, acc = cross_entropy_loss_and_accuracy(lg, _dev(tokens), _dev(valid))
    loss.backward()
    rl, ra, rd = R.cross_entropy_loss_and_accuracy(_np(logits), tokens, valid)
    assert abs(loss.item() - rl) <= 1e-4 * abs(rl) and abs(acc.item() - ra) <= 1e-6
    assert np.abs(_np(lg.grad) - rd).max() <= 2 ** -7 * np.abs(rd).max() + 1e-9
    # chunked head: same loss, gradients of hidden and kernel vs float64 chain rule
    l2, a2 = chunked_lm_head_loss(hd, Wd, _dev(tokens), _dev(valid), chunk=32)
    l2.backward()
    assert abs(l2.item() - rl) <= 1e-4 * abs(rl) and abs(a2.item() - ra) <= 1e-6
    dh_ref = rd @ W.astype(np.float64).T
    dW_ref = np.einsum("bsd,bsv->dv", h.astype(np.float64), rd)
    assert np.abs(_np(hd.grad) - dh_ref).max() <= 2e-2 * np.abs(dh_ref).max()
    assert np.abs(_np(Wd.grad) - dW_ref).max() <= 2e-2 * np.abs(dW_ref).max()
    l3, _ = chunked_lm_head_loss(hd.detach(), Wd.detach(), _dev(tokens), _dev(valid), chunk=96)
    assert abs(l3.item() - l2.item()) <= 1e-6 * abs(l2.item())


def test_vision_text_loss_combination():
    import torch
    from lwm_amd.llama_ops import vision_text_loss
    B, S = 1, 64
    g = np.random.default_rng(12)
    vl = round_bf16(g.standard_normal((B, S, 8448)).astype(np.float32))
    tl = round_bf16(g.standard_normal((B, S, 32000)).astype(np.float32))
    tvm = g.random((B, S)) > 0.5
    tgt = np.where(tvm, g.integers(0, 8448, (B, S)), g.integers(0, 32000, (B, S)))
    lm = (g.random((B, S)) > 0.1).astype(np.float32)
    loss, m = vision_text_loss(_dev(vl, torch.bfloat16), _dev(tl, torch.bfloat16), _dev(tgt), _dev(lm), _dev(tvm))
    rv, _, _ = R.cross_entropy_loss_and_accuracy(vl, np.where(tvm, tgt, 0), lm * tvm)
    rt, _, _ = R.cross_entropy_loss_and_accuracy(tl, np.where(tvm, 0, tgt), lm * (1.0 - tvm))
    assert abs(loss.item() - 0.5 * (rv + rt)) <= 1e-4 * abs(0.5 * (rv + rt))
    assert abs(m["vision_loss"].item() - rv) <= 1e-4 * abs(rv)


def test_swiglu_fwd_bwd():
    import torch
    from lwm_amd.llama_ops import swiglu
    a, b, g = _rnd((3, 700, 88), 21, 2.0), _rnd((3, 700, 88), 22), _rnd((3, 700, 88), 23)
    ad, bd = _dev(a, torch.bfloat16).requires_grad_(True), _dev(b, torch.bfloat16).requires_grad_(True)
    y = swiglu(ad, bd)
    ref = R.swiglu(a, b)
    assert np.abs(_np(y) - ref).max() <= 2 ** -7 * np.abs(ref).max()
    y.backward(_dev(g, torch.bfloat16))
    da, db = R.swiglu_bwd(a, b, g)
    assert np.abs(_np(ad.grad) - da).max() <= 2 ** -7 * np.abs(da).max() + 1e-6
    assert np.abs(_np(bd.grad) - db).max() <= 2 ** -7 * np.abs(db).max() + 1e-6


@pytest.mark.parametrize("rows,K,N", [(1, 4096, 4096), (3, 4096, 11008), (1, 11008, 4096), (2, 4096, 32000), (4, 4096, 12288)])
def test_gemv_decode_projection_7b_shapes(rows, K, N):
    """lwm_gemv_bf16 at LWM-7B's projection shapes (attention, FFN in / out, lm_head) against the f32 product."""
    import torch
    from lwm_amd.llama_ops import gemv
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(rows, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(K, N, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    yf = gemv(x, w, torch.float32)
    ref = x.double() @ w.double()
    assert (yf.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    yb = gemv(x, w)
    assert torch.equal(yb, yf.to(torch.bfloat16))
    assert torch.equal(gemv(x, w, torch.float32), yf)                # deterministic


def test_gemv_fused_neighbours_equal_the_separate_launches_on_gpu():
    """lwm_gemv_fused_bf16 at LWM-7B's shapes: norm on load + residual in the reduction + the partial sums of squares,
    against RMSNorm kernel -> gemv -> bf16 add launched one by one."""
    import torch
    from lwm_amd.llama_ops import RMSNorm, gemv, gemv_fused, gemv_multi
    g = torch.Generator(device="cuda").manual_seed(9)
    rows, d, F = 2, 4096, 11008
    x = torch.randn(rows, d, generator=g, device="cuda").to(torch.bfloat16)
    res = torch.randn(rows, d, generator=g, device="cuda").to(torch.bfloat16)
    wo = (torch.randn(d, d, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    w1, w3 = ((torch.randn(d, F, generator=g, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(2))
    norm = RMSNorm(d, 1e-6).cuda()
    with torch.no_grad():
        norm.kernel.copy_(1.0 + 0.1 * torch.randn(d, generator=g, device="cuda"))
        # x2 = res + x @ wo, then norm -> w1 | w3
        (x2,), ss = gemv_fused(x, (wo,), residual=res, want_ss=True)
        want_x2 = gemv(x, wo) + res
        assert torch.equal(x2, want_x2)
        assert torch.allclose(ss.sum(-1), x2.float().pow(2).sum(-1), rtol=1e-5)
        a, b = gemv_fused(x2, (w1, w3), norm=(ss, norm.kernel.to(torch.bfloat16), 1e-6))
        ra, rb = gemv_multi(norm(x2[:, None])[:, 0].contiguous(), [w1, w3])
        for got, ref in ((a, ra), (b, rb)):      # rstd sums in another order: at most one bf16 ulp on a few inputs
            assert (got.float() - ref.float()).abs().max() <= 2 ** -7 * ref.float().abs().max()
        assert torch.equal(gemv_fused(x2, (w1, w3), norm=(ss, norm.kernel.to(torch.bfloat16), 1e-6))[0], a)    # deterministic


def test_dense_routes_decode_rows_through_gemv():
    """`dense` = flax nn.Dense without bias: <= 4 rows without autograd take lwm_gemv_bf16, anything else the
    library GEMM; both agree to bf16 rounding."""
    import torch
    from lwm_amd.llama_ops import dense
    g = torch.Generator(device="cuda").manual_seed(6)
    w = (torch.randn(4096, 4096, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    x = torch.randn(2, 1, 4096, generator=g, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        a = dense(x, w)
    b = x @ w
    assert a.shape == b.shape and (a.float() - b.float()).abs().max() <= 2 ** -7 * b.float().abs().max()
    wp = torch.nn.Parameter(w)
    y = dense(x, wp)                                                  # autograd on: library GEMM, differentiable
    y.float().sum().backward()
    assert wp.grad is not None
    big = torch.randn(2, 8, 4096, generator=g, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        assert torch.equal(dense(big, w), big @ w)


# ---- round 6: the library GEMMs of the training path, re-laid for hipBLASLt (lwm_amd/llama_ops.py)
@pytest.mark.parametrize("R_,C_", [(4096, 4096), (256, 11008), (22016, 4096), (100, 36)])
def test_transpose2d_is_exact(R_, C_):
    import torch
    from lwm_amd.llama_ops import transpose2d
    x = torch.randn(R_, C_ + 8, device="cuda").to(torch.bfloat16)[:, :C_]          # (a row stride that is not the width)
    out = torch.full((C_, R_ + 16), float("nan"), device="cuda", dtype=torch.bfloat16)
    transpose2d(x, out[:, :R_])
    assert torch.equal(out[:, :R_], x.t())
    assert bool(torch.isnan(out[:, R_:]).all())


def test_fused_operators_against_separate_ones():
    """qkv_rope / dense_fused / swiglu_halves / rmsnorm_residual -- the training path's operators -- against the separate
    operators they replace (dense_multi + apply_rotary_emb, x @ k, swiglu, RMSNorm + add), forward and every gradient:
    same math, another grouping of the library GEMMs (results agree to bf16 GEMM noise; the elementwise kernels are the same
    instructions)."""
    import torch
    from lwm_amd import llama_ops as LO
    bf = torch.bfloat16
    torch.manual_seed(0)
    B, S, H, D, F = 1, 640, 4, 128, 1408
    d = H * D
    mk = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).to(bf)
    x0 = mk(B, S, d)
    ws = {n: mk(d, d, sc=0.03) for n in ("wq", "wk", "wv", "wo")}
    w1, w3, w2 = mk(d, F, sc=0.03), mk(d, F, sc=0.03), mk(F, d, sc=0.03)
    nw = (1 + 0.1 * torch.randn(d, device="cuda")).float()
    tab = LO.precompute_freqs_cis(D, 2048, 10000.0, device="cuda")
    pos = torch.arange(S, device="cuda", dtype=torch.int32)[None]
    gout = mk(B, S, d)

    def run(fused):
        leaves = {n: t.clone().requires_grad_(True) for n, t in dict(ws, w1=w1, w3=w3, w2=w2, x=x0, nw=nw).items()}
        L = leaves
        norm = LO.RMSNorm(d).cuda()
        norm.kernel = torch.nn.Parameter(L["nw"].detach().clone())
        if fused:
            h, xr = LO.rmsnorm_residual(norm, L["x"])
            q, k, v = LO.qkv_rope(h, L["wq"], L["wk"], L["wv"], tab, pos, H)
            a = (q.float() * torch.tanh(k.float()) + v.float()).to(bf).reshape(B, S, d)       # (a stand-in for attention)
            x2 = LO.dense_fused(a, (L["wo"],), xr)
            h2, x2r = LO.rmsnorm_residual(norm, x2)
            out = LO.dense_fused(LO.swiglu_halves(LO.dense_fused(h2, (L["w1"], L["w3"]))), (L["w2"],), x2r)
        else:
            h = norm(L["x"])
            q, k, v = (t.reshape(B, S, H, D) for t in (h @ L["wq"], h @ L["wk"], h @ L["wv"]))
            q, k = LO.apply_rotary_emb(q, k, tab, pos)
            a = (q.float() * torch.tanh(k.float()) + v.float()).to(bf).reshape(B, S, d)
            x2 = L["x"] + a @ L["wo"]
            h2 = norm(x2)
            out = x2 + LO.swiglu((h2 @ L["w1"]).contiguous(), (h2 @ L["w3"]).contiguous()) @ L["w2"]
        out.backward(gout)
        grads = {n: t.grad.float() for n, t in leaves.items() if n != "nw"}
        grads["nw"] = norm.kernel.grad.float()
        return out.float(), grads

    o1, g1 = run(True)
    o0, g0 = run(False)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert rel(o1, o0) <= 1e-2, rel(o1, o0)
    for n in g0:
        assert rel(g1[n], g0[n]) <= 2e-2, (n, rel(g1[n], g0[n]))
        assert g1[n].shape == g0[n].shape


def test_relayout_cache_follows_the_weights():
    """the (out, in) copy of a kernel is rebuilt when the kernel changes in place, and on weights_changed()"""
    import torch
    from lwm_amd import llama_ops as LO
    w = (torch.randn(256, 512, device="cuda") * 0.05).to(torch.bfloat16)
    x = torch.randn(64, 256, device="cuda").to(torch.bfloat16)
    y0 = LO.dense_fused(x, (w,))
    assert torch.allclose(y0.float(), (x @ w).float(), rtol=2e-2, atol=2e-2)
    wt0 = w._lwm_relayout[1]
    assert LO._relayout((w,))[0] is wt0                               # cached
    w.mul_(2.0)
    y1 = LO.dense_fused(x, (w,))
    assert torch.allclose(y1.float(), 2 * y0.float(), rtol=2e-2, atol=2e-2)
    wt1 = w._lwm_relayout[1]
    LO.weights_changed()
    assert LO._relayout((w,))[0] is not wt1


# ---- round 6: the hand-written weight-gradient GEMM (csrc/gemm_wgrad.h) behind llama_ops.wgrad
@pytest.mark.parametrize("M,K,N", [
    (4096, 4096, 12288),      # wq|wk|wv: 768 tiles = 3 whole rounds on 256 CUs
    (2048, 4096, 22016),      # w1|w3: 1376 tiles = 5 rounds + 96 stream-K tiles
    (2048, 11008, 4096),      # w2: 2 rounds + 176 stream-K tiles, ranges that span two tiles
    (4096, 256, 512),         # 2 tiles: stream-K alone, 128 stages over 256 blocks (empty blocks too)
    (8192, 4096, 32000),      # lm_head chunk
    (96, 512, 768),           # a ragged call: three stages
])
def test_wgrad_gemm_against_fp32(M, K, N):
    """dW = x^T g (the flax Dense kernel's gradient, lwm/llama.py:390-421) with f32 accumulation: products of bf16 values are
    exact in f32, so against an f32 torch GEMM of the same operands only the ORDER of the sums differs -- bound, per element: the bf16
    rounding of the result (half an ulp: 2^-8 relative at most) plus f32 summation noise; and the kernel is deterministic."""
    import torch
    from lwm_amd import llama_ops as ops
    gen = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, K, device="cuda", generator=gen).to(torch.bfloat16)
    g = torch.randn(M, N, device="cuda", generator=gen).to(torch.bfloat16)
    dw = ops.wgrad(x, g)
    assert dw.shape == (K, N) and dw.dtype == torch.bfloat16
    assert torch.equal(dw, ops.wgrad(x, g))
    big = max(float((x.float().t() @ g[:, c0:c0 + 4096].float()).abs().max()) for c0 in range(0, N, 4096))
    for c0 in range(0, N, 4096):
        ref = x.float().t() @ g[:, c0:c0 + 4096].float()
        rel = (dw[:, c0:c0 + 4096].float() - ref).abs() / ref.abs().clamp_min(0.01 * big)
        assert float(rel.max()) <= 2 ** -8 * 1.05          # every element: the bf16 rounding of the f32 sum, nothing more


def test_wgrad_one_wave_per_simd_form_agrees(monkeypatch):
    """LWM_WGRAD_WAVES=4 (128 x 128 per wave, one wave per SIMD; kept as a switch, profiles/r06_wgrad.md): the same sums in the
    same order per element -- bit-identical to the default form, stream-K tail included."""
    import torch
    from lwm_amd import llama_ops as ops
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(2048, 4096, device="cuda", generator=gen).to(torch.bfloat16)
    g = torch.randn(2048, 22016, device="cuda", generator=gen).to(torch.bfloat16)
    base = ops.wgrad(x, g)
    monkeypatch.setenv("LWM_WGRAD_WAVES", "4")
    assert torch.equal(ops.wgrad(x, g), base)


def test_wgrad_on_strided_views_and_the_library_fallback(monkeypatch):
    """The operands the training path hands over are views: a column block of the fused (S, 3d) gradient buffer, rows with a
    padded leading dimension.  Shapes the kernel does not take go to the library; both forms agree to bf16 rounding."""
    import torch
    from lwm_amd import llama_ops as ops
    gen = torch.Generator(device="cuda").manual_seed(7)
    M, K, N = 1024, 512, 768
    xb = torch.randn(M, K + 64, device="cuda", generator=gen).to(torch.bfloat16)
    gb = torch.randn(M, 3 * N, device="cuda", generator=gen).to(torch.bfloat16)
    x, g = xb[:, :K], gb[:, N:2 * N]
    dw = ops.wgrad(x, g)
    ref = x.float().t() @ g.float()
    assert float((dw.float() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max()) + 0.05
    monkeypatch.setenv("LWM_WGRAD_HIP", "0")
    lib_form = ops.wgrad(x.contiguous(), g.contiguous())
    assert float((dw.float() - lib_form.float()).abs().max()) <= 2 ** -7 * float(ref.abs().max())
    monkeypatch.delenv("LWM_WGRAD_HIP")
    odd = ops.wgrad(xb[:1000, :K], gb[:1000, :N])          # 1000 rows: not a multiple of 32 -> the library form
    ref = xb[:1000, :K].float().t() @ gb[:1000, :N].float()
    assert float((odd.float() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
