"""Parity of the VQGAN HIP kernels (through the C ABI, on MI355X) against the C
oracle.  The bar is BIT-EXACT: activations compare with np.array_equal, code
indices are identical (north_star: "bit-exact for VQGAN code indices").  The
only non-structural source of a difference is the f64 GroupNorm statistics sum
(rounded to f32 after the reduction) -- none occurs on these
seeds."""
import numpy as np
import pytest

from lwm_amd.vqgan import VQGANConfig, random_params
from oracle import vqgan_ref as R

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _conv_inputs(seed, B, H, W, Cin, Cout, k):
    g = np.random.default_rng(seed)
    x = g.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (g.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = g.standard_normal(Cout).astype(np.float32)
    return x, w, b


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,kw", [
    (1, 64, 64, 128, 128, 3, {}),                                   # 128x128 tile, L2-scale layer
    (2, 16, 16, 768, 768, 3, {}),                                   # 32x128 tile, long K (6912)
    (1, 32, 32, 3, 128, 3, {}),                                     # conv_in (Cin = 3)
    (1, 32, 32, 128, 3, 3, dict(clip=True)),                        # decoder out (Cout = 3) + clip
    (3, 16, 16, 768, 64, 3, {}),                                    # encoder out conv, 128x64 tile
    (1, 16, 16, 64, 64, 1, {}),                                     # quant_conv
    (1, 64, 64, 256, 256, 3, dict(stride=2, pad=0, out_hw=(32, 32))),  # Downsample
    (1, 32, 32, 256, 256, 3, dict(up_shift=1)),                     # Upsample
    (1, 21, 13, 40, 200, 3, {}),                                    # ragged everything
    (2, 21, 13, 3, 256, 3, {}),                                     # conv_cin4: ragged M, two channel tiles
    (1, 33, 17, 3, 128, 3, dict(stride=2, pad=0, out_hw=(16, 8))),  # conv_cin4 with a stride
    (2, 8, 48, 128, 3, 3, dict(clip=True)),                         # conv_patch_c128_out3 (W % 16 == 0, H % 4 == 0)
    (1, 8, 16, 128, 3, 3, dict(up_shift=1)),                        # ... over the upsampled image, no clip
])
def test_conv_bit_exact(B, H, W, Cin, Cout, k, kw):
    import torch
    from lwm_amd import ops
    x, w, b = _conv_inputs(1, B, H, W, Cin, Cout, k)
    got = ops.conv2d_nhwc(_dev(x), _dev(w), _dev(b), **kw)
    torch.cuda.synchronize()
    ref = R.conv2d(x, w, b, **kw)
    assert np.array_equal(got.cpu().numpy(), ref)


def test_conv_residual():
    from lwm_amd import ops
    x, w, b = _conv_inputs(2, 1, 32, 32, 128, 256, 3)
    res = np.random.default_rng(3).standard_normal((1, 32, 32, 256)).astype(np.float32)
    got = ops.conv2d_nhwc(_dev(x), _dev(w), _dev(b), residual=_dev(res))
    assert np.array_equal(got.cpu().numpy(), R.conv2d(x, w, b, residual=res))


@pytest.mark.parametrize("B,H,W,Cin,Cout,kw", [
    (2, 64, 64, 128, 128, {}),                    # conv_patch_c128: 128 tiles of 4x16 pixels
    (1, 64, 64, 128, 256, dict(up_shift=1)),      # Upsample folded in, two channel tiles
    (2, 64, 64, 256, 256, {}),                    # conv_patch_c256: 128 tiles of 4x16 pixels x 256 channels
    (1, 32, 64, 256, 256, dict(up_shift=1)),      # c256 + Upsample, non-square image
    (1, 96, 80, 128, 128, {}),                    # 24 x 5 tiles: odd tile counts, every edge
])
def test_conv_patch_resident_bit_exact(B, H, W, Cin, Cout, kw):
    """3x3 stride-1 SAME with 128 / 256 input channels and enough tiles for the chip: the halo-patch kernels
    (vqgan_conv.h::conv_patch_body; A patch resident in LDS, B operands straight from L1/L2) -- bit-equal to
    the oracle, with bias, residual and the folded upsample."""
    from lwm_amd import ops
    x, w, b = _conv_inputs(21, B, H, W, Cin, Cout, 3)
    up = kw.get("up_shift", 0)
    res = np.random.default_rng(22).standard_normal((B, H << up, W << up, Cout)).astype(np.float32)
    got = ops.conv2d_nhwc(_dev(x), _dev(w), _dev(b), residual=_dev(res), **kw)
    assert np.array_equal(got.cpu().numpy(), R.conv2d(x, w, b, residual=res, **kw))


def test_conv_full_resolution_layer():
    """One encoder level-0 conv at BASELINE size: 256x256x128 -> 128 (19.3 GFLOP)."""
    from lwm_amd import ops
    x, w, b = _conv_inputs(4, 1, 256, 256, 128, 128, 3)
    got = ops.conv2d_nhwc(_dev(x), _dev(w), _dev(b))
    assert np.array_equal(got.cpu().numpy(), R.conv2d(x, w, b))


def test_conv_bdirect_ranged_staging_across_images():
    """conv_igemm_128x128_bd stages by ranged buffer loads (offsets relative to the image of a tile's first pixel; out-of-image
    taps past the descriptor's range come back as zeros): 36 images of 960 pixels -- 128-pixel tiles straddle images, 270
    tiles -- stride 1 and the stride-2 Downsample form, with the residual; bit-identical to the oracle."""
    from lwm_amd import ops
    x, w, b = _conv_inputs(41, 36, 24, 40, 64, 128, 3)
    res = np.random.default_rng(42).standard_normal((36, 24, 40, 128)).astype(np.float32)
    got = ops.conv2d_nhwc(_dev(x), _dev(w), _dev(b), residual=_dev(res))
    assert np.array_equal(got.cpu().numpy(), R.conv2d(x, w, b, residual=res))
    kw = dict(stride=2, pad=0, out_hw=(12, 20))
    got = ops.conv2d_nhwc(_dev(x), _dev(w), _dev(b), **kw)
    assert np.array_equal(got.cpu().numpy(), R.conv2d(x, w, b, **kw))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (3, 128, 128, 128, 128),      # conv_patch_c128_res: 384 tiles of 8x16 on 256 CUs -- one or two tiles per workgroup
    (3, 96, 64, 256, 256),        # conv_patch_c256_res: 288 tiles of 4x16
])
def test_conv_patch_persistent_walk(B, H, W, Cin, Cout):
    """The patch kernels are persistent (one workgroup per CU walks the tiles; the next tile's patch is requested before this
    tile's results are stored, the residual operand beside the last MFMAs): more tiles than CUs, a ragged count per workgroup,
    with and without the residual -- bit-identical to the oracle."""
    from lwm_amd import ops
    x, w, b = _conv_inputs(31, B, H, W, Cin, Cout, 3)
    res = np.random.default_rng(32).standard_normal((B, H, W, Cout)).astype(np.float32)
    got = ops.conv2d_nhwc(_dev(x), _dev(w), _dev(b), residual=_dev(res))
    assert np.array_equal(got.cpu().numpy(), R.conv2d(x, w, b, residual=res))
    got = ops.conv2d_nhwc(_dev(x), _dev(w), _dev(b))
    assert np.array_equal(got.cpu().numpy(), R.conv2d(x, w, b))


@pytest.mark.parametrize("B,HW,C,silu", [(1, 65536, 128, True), (2, 4096, 256, True), (1, 1024, 512, False),
                                          (3, 256, 768, True), (1, 77, 128, True)])
def test_groupnorm_silu_bit_exact(B, HW, C, silu):
    from lwm_amd import ops
    g = np.random.default_rng(5)
    x = (g.standard_normal((B, HW, C)) * 2 + 0.3).astype(np.float32)
    gamma = (1 + 0.1 * g.standard_normal(C)).astype(np.float32)
    beta = (0.1 * g.standard_normal(C)).astype(np.float32)
    got = ops.groupnorm_silu(_dev(x), _dev(gamma), _dev(beta), silu=silu)
    ref = R.groupnorm(x, gamma, beta, silu=silu)
    assert np.array_equal(got.cpu().numpy(), ref)


def test_silu_extremes():
    """exp clamp range and large |x| (no NaN/inf, identical to the oracle)."""
    from lwm_amd import ops
    x = np.zeros((1, 64, 128), np.float32)
    x[0, :, 0] = np.linspace(-200, 200, 64)
    x[0, :, 5] = np.linspace(-1e-3, 1e-3, 64)
    gamma, beta = np.full(128, 50.0, np.float32), np.zeros(128, np.float32)
    got = ops.groupnorm_silu(_dev(x), _dev(gamma), _dev(beta), silu=True).cpu().numpy()
    assert np.isfinite(got).all()
    assert np.array_equal(got, R.groupnorm(x, gamma, beta, silu=True))


@pytest.mark.parametrize("N,E", [(256, 8192), (1000, 8192), (33, 512)])
def test_vq_bit_exact(N, E):
    from lwm_amd import ops
    g = np.random.default_rng(7)
    cb = g.uniform(-1.0 / E, 1.0 / E, (E, 64)).astype(np.float32)
    z = (g.standard_normal((N, 64)) * 2.0 / E).astype(np.float32)
    z[3] = cb[5]
    cb[9] = cb[4]          # duplicate code: the first index must win
    z[7] = cb[9]
    cbd, zd = _dev(cb), _dev(z)
    idx = ops.vq_argmin(zd, cbd)
    ref = R.vq_argmin(z, cb)
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert int(idx[3]) == 5 and int(idx[7]) == 4
    assert np.array_equal(ops.vq_gather(cbd, idx).cpu().numpy(), R.vq_gather(cb, ref))
    assert np.array_equal(ops.vq_gather(cbd, idx, zd).cpu().numpy(), R.vq_gather(cb, ref, z))


def _model(cfg, seed):
    from lwm_amd.vqgan import VQGAN
    params = random_params(cfg, seed)
    return params, VQGAN(params=params, config=cfg)


def test_model_encode_decode_small():
    """Whole tokeniser at resolution 64 (all five levels, all channel widths)."""
    cfg = VQGANConfig.get_default_config(dict(resolution=64))
    params, vq = _model(cfg, 11)
    px = np.random.default_rng(12).uniform(-1, 1, (2, 64, 64, 3)).astype(np.float32)
    zq, idx = vq.encode(px)
    rzq, ridx = R.encode(params, px, cfg.as_dict())
    if not np.array_equal(idx.cpu().numpy(), ridx):           # code indices: identical
        raise AssertionError(R.index_mismatch_report(params, px, idx.cpu().numpy(), cfg.as_dict())[2])
    assert np.array_equal(zq.cpu().numpy(), rzq)
    rec = vq.decode(idx)
    rrec = R.decode(params, ridx, cfg.as_dict())
    assert np.array_equal(rec.cpu().numpy(), rrec)
    # 5-D video input folds T into the batch (lwm/vqgan.py:119-121, :134-136)
    zq5, idx5 = vq.encode(px[None])
    assert tuple(idx5.shape) == (1, 2, 4, 4) and np.array_equal(idx5[0].cpu().numpy(), ridx)
    assert tuple(vq.decode(idx5).shape) == (1, 2, 64, 64, 3)


def test_model_full_resolution_frame():
    """BASELINE config: one 256x256 frame, default VQGANConfig (216.6 GFLOP encode,
    477.4 GFLOP decode); indices and pixels identical to the oracle."""
    cfg = VQGANConfig.get_default_config()
    params, vq = _model(cfg, 21)
    px = np.random.default_rng(22).uniform(-1, 1, (1, 256, 256, 3)).astype(np.float32)
    zq, idx = vq.encode(px)
    rzq, ridx = R.encode(params, px, cfg.as_dict())
    assert tuple(idx.shape) == (1, 16, 16)
    if not np.array_equal(idx.cpu().numpy(), ridx):           # say how close the calls were (top-2 margins)
        raise AssertionError(R.index_mismatch_report(params, px, idx.cpu().numpy(), cfg.as_dict())[2])
    assert np.array_equal(zq.cpu().numpy(), rzq)
    rec = vq.decode(idx).cpu().numpy()
    rrec = R.decode(params, ridx, cfg.as_dict())
    assert rec.min() >= -1 and rec.max() <= 1
    assert np.array_equal(rec, rrec)


def test_errors_are_loud():
    import torch
    from lwm_amd import ops
    with pytest.raises(ValueError):
        ops.conv2d_nhwc(torch.zeros(1, 8, 8, 4), torch.zeros(3, 3, 4, 8).cuda())   # CPU tensor
    with pytest.raises(ValueError):
        ops.conv2d_nhwc(torch.zeros(1, 8, 8, 4).cuda(), torch.zeros(3, 3, 5, 8).cuda())
    with pytest.raises(Exception):
        ops.vq_argmin(torch.zeros(4, 32).cuda(), torch.zeros(16, 32).cuda())       # D != 64
