"""The C oracle (oracle/vqgan_ref.c + oracle/vqgan_ref.py) against an INDEPENDENT
float64 PyTorch implementation of lwm/vqgan.py (F.conv2d / F.group_norm /
F.silu / nearest interpolate / cdist-free distance formula).  This is what makes
the oracle a faithful restatement; the HIP kernels are then held bit-exact to
the oracle (tests/test_emu_vqgan.py on CPU, tests/test_gpu_vqgan.py on MI355X).

Tolerances: activations rel. max err <= 2e-5 (f32 oracle vs f64 reference);
code indices: every disagreement must be a near-tie (top-2 distance margin of the
float64 reference below 1e-6 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lwm_amd.vqgan import VQGANConfig, random_params
from oracle import vqgan_ref as R


def _t(a):
    return torch.from_numpy(np.asarray(a)).double()


def t_conv(p, x, stride=1, down=False, up=False, res=None):
    """x NCHW float64; flax kernel HWIO -> OIHW."""
    w = _t(p["kernel"]).permute(3, 2, 0, 1)
    k = w.shape[-1]
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if down:
        x = F.pad(x, (0, 1, 0, 1))
        y = F.conv2d(x, w, _t(p["bias"]), stride=2)
    else:
        y = F.conv2d(x, w, _t(p["bias"]), padding=(k - 1) // 2)
    return y if res is None else y + res


def t_gn_silu(p, x):
    return F.silu(F.group_norm(x, 32, _t(p["scale"]), _t(p["bias"]), eps=1e-6))


def t_resnet(p, x):
    h = t_conv(p["Conv_0"], t_gn_silu(p["GroupNorm_0"], x))
    h = t_gn_silu(p["GroupNorm_1"], h)
    res = t_conv(p["Conv_2"], x) if "Conv_2" in p else x
    return t_conv(p["Conv_1"], h, res=res)


def t_encoder(p, x, cfg):
    h = t_conv(p["Conv_0"], x)
    for lvl in range(cfg.num_resolutions):
        bp = p[f"DownsamplingBlock_{lvl}"]
        for i in range(cfg.num_res_blocks):
            h = t_resnet(bp[f"ResnetBlock_{i}"], h)
        if lvl != cfg.num_resolutions - 1:
            h = t_conv(bp["Downsample_0"]["Conv_0"], h, down=True)
    h = t_resnet(p["MidBlock_0"]["ResnetBlock_1"], t_resnet(p["MidBlock_0"]["ResnetBlock_0"], h))
    return t_conv(p["Conv_1"], t_gn_silu(p["GroupNorm_0"], h))


def t_decoder(p, z, cfg):
    h = t_conv(p["Conv_0"], z)
    h = t_resnet(p["MidBlock_0"]["ResnetBlock_1"], t_resnet(p["MidBlock_0"]["ResnetBlock_0"], h))
    for order, lvl in enumerate(reversed(range(cfg.num_resolutions))):
        bp = p[f"UpsamplingBlock_{order}"]
        for i in range(cfg.num_res_blocks + 1):
            h = t_resnet(bp[f"ResnetBlock_{i}"], h)
        if lvl != 0:
            h = t_conv(bp["Upsample_0"]["Conv_0"], h, up=True)
    return t_conv(p["Conv_1"], t_gn_silu(p["GroupNorm_0"], h)).clamp(-1, 1)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


CFG = VQGANConfig.get_default_config(dict(resolution=32, channel_mult=(1, 2, 4), num_embeddings=1024))


@pytest.fixture(scope="module")
def params():
    return random_params(CFG, seed=3)


def test_primitives_vs_float64():
    g = np.random.default_rng(0)
    x = g.standard_normal((2, 9, 7, 128)).astype(np.float32)
    p = {"kernel": (g.standard_normal((3, 3, 128, 256)) / 34).astype(np.float32),
         "bias": g.standard_normal(256).astype(np.float32)}
    xt = _t(x).permute(0, 3, 1, 2)
    nhwc = lambda t: t.permute(0, 2, 3, 1).numpy()
    assert _rel(R.conv2d(x, p["kernel"], p["bias"]), nhwc(t_conv(p, xt))) < 2e-6
    assert _rel(R.conv2d(x, p["kernel"], p["bias"], up_shift=1), nhwc(t_conv(p, xt, up=True))) < 2e-6
    x2 = g.standard_normal((1, 8, 8, 128)).astype(np.float32)
    assert _rel(R.conv2d(x2, p["kernel"], p["bias"], stride=2, pad=0, out_hw=(4, 4)),
                nhwc(t_conv(p, _t(x2).permute(0, 3, 1, 2), down=True))) < 2e-6
    gp = {"scale": (1 + 0.1 * g.standard_normal(128)).astype(np.float32),
          "bias": (0.1 * g.standard_normal(128)).astype(np.float32)}
    assert _rel(R.groupnorm(x, gp["scale"], gp["bias"], silu=True), nhwc(t_gn_silu(gp, xt))) < 2e-6


def test_exp_is_accurate():
    xs = np.linspace(-87, 88, 20001).astype(np.float32)
    got = np.array([R.expf(v) for v in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 2.5e-7     # < 2 ulp of f32


def test_encode_decode_vs_float64(params):
    g = np.random.default_rng(1)
    px = g.uniform(-1, 1, (2, CFG.resolution, CFG.resolution, 3)).astype(np.float32)
    cfg = CFG.as_dict()
    zq, idx = R.encode(params, px, cfg)
    h = t_encoder(params["encoder"], _t(px).permute(0, 3, 1, 2), CFG)
    h = t_conv(params["quant_conv"], h).permute(0, 2, 3, 1)            # NHWC f64
    # the oracle's pre-quantisation activations (recomputed) agree with float64
    h32 = R.conv2d(R.encoder(params["encoder"], px, cfg), params["quant_conv"]["kernel"],
                   params["quant_conv"]["bias"])
    assert _rel(h32, h.numpy()) < 2e-5
    cb = _t(params["quantize"]["embeddings"])
    zf = h.reshape(-1, h.shape[-1])
    d = (zf ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1)[None] - 2 * zf @ cb.T
    ref_idx = d.argmin(1).reshape(idx.shape).numpy()
    bad = np.argwhere(ref_idx != idx)
    for b in bad:                       # every disagreement must be a float32-level near-tie
        row = d.reshape(idx.shape + (-1,))[tuple(b)]
        top2 = torch.topk(-row, 2).values
        assert abs(float(top2[0] - top2[1])) <= 1e-6 * float(row.abs().max()), (b, top2)
    assert len(bad) <= 0.02 * idx.size
    # quantised value: z + (e - z)
    e = params["quantize"]["embeddings"][idx]
    assert np.array_equal(zq, h32 + (e - h32))
    # decode path
    rec = R.decode(params, idx, cfg)
    zt = _t(e).permute(0, 3, 1, 2)
    ref = t_decoder(params["decoder"], t_conv(params["post_quant_conv"], zt), CFG).permute(0, 2, 3, 1).numpy()
    assert rec.shape == (2, CFG.resolution, CFG.resolution, 3)
    assert np.abs(rec - ref).max() < 2e-5
    assert rec.min() >= -1 and rec.max() <= 1


def test_video_5d_folds_time_into_batch(params):
    g = np.random.default_rng(2)
    px = g.uniform(-1, 1, (1, 2, CFG.resolution, CFG.resolution, 3)).astype(np.float32)
    cfg = CFG.as_dict()
    zq5, idx5 = R.encode(params, px, cfg)
    zq4, idx4 = R.encode(params, px[0], cfg)
    assert idx5.shape == (1, 2) + idx4.shape[1:] and np.array_equal(idx5[0], idx4)
    assert np.array_equal(zq5[0], zq4)
    assert R.decode(params, idx5, cfg).shape == px.shape


def test_index_mismatch_report_names_margins():
    """a code-index mismatch is reported with its top-2 margin."""
    import numpy as np
    from lwm_amd.vqgan import VQGANConfig, random_params
    from oracle import vqgan_ref as V
    cfg = VQGANConfig.get_default_config(dict(resolution=32, channel_mult=(1, 2, 4), num_embeddings=1024))
    params = random_params(cfg, seed=5)
    px = np.random.default_rng(6).uniform(-1, 1, (1, 32, 32, 3)).astype(np.float32)
    _, idx = V.encode(params, px, cfg.as_dict())
    assert V.index_mismatch_report(params, px, idx, cfg.as_dict())[:2] == (0, idx.size)
    bad = idx.copy()
    bad.reshape(-1)[3] = (bad.reshape(-1)[3] + 1) % 1024
    n, total, text = V.index_mismatch_report(params, px, bad, cfg.as_dict())
    assert (n, total) == (1, idx.size) and "top-2 margin" in text and "pos 3" in text


def test_decoder_tree_uses_flax_creation_order_names():
    """flax @nn.compact numbers submodules in creation order; the reference Decoder creates its
    UpsamplingBlocks for i_level = nres-1 .. 0 (lwm/vqgan.py:180), so in a real checkpoint
    UpsamplingBlock_0 is the 768-channel block WITH Upsample_0 and the last one (256 -> 128 channels) has
    none.  random_params, the oracle and the product all have to agree with that, or decode() cannot
    run on the reference's pickle."""
    from lwm_amd.vqgan import VQGANConfig, random_params
    cfg = VQGANConfig.get_default_config()
    dec = random_params(cfg, seed=0)["decoder"]
    n = cfg.num_resolutions
    hc, mult = cfg.hidden_channels, cfg.channel_mult
    first, last = dec["UpsamplingBlock_0"], dec[f"UpsamplingBlock_{n - 1}"]
    assert first["ResnetBlock_0"]["Conv_0"]["kernel"].shape[-1] == hc * mult[n - 1] == 768
    assert "Upsample_0" in first and "Upsample_0" not in last
    assert last["ResnetBlock_0"]["Conv_0"]["kernel"].shape[2:] == (hc * mult[1], hc * mult[0])
    assert "Conv_2" in last["ResnetBlock_0"]           # 256 -> 128 needs the 1x1 shortcut (lwm/vqgan.py:258-262)
    for order in range(n):
        lvl = n - 1 - order
        assert dec[f"UpsamplingBlock_{order}"]["ResnetBlock_2"]["Conv_1"]["kernel"].shape[-1] == hc * mult[lvl]
