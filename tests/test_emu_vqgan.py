"""The VQGAN HIP kernel sources (lwm_amd/csrc/vqgan_*.h) compiled for the host
and run one fiber per lane (tests/emu/), through the same C ABI, against the C
oracle (oracle/vqgan_ref.c).  The contract is BIT-EXACT: the exact-f32 MFMA is
emulated as the ordered fmaf pair it is documented to be (confirmed on hardware
by tests/test_gpu_probe.py::test_mfma_f32_is_ordered_fma_chain)."""
import numpy as np
import pytest

from oracle import vqgan_ref as R
from tests import _emu


def _rng(seed):
    return np.random.default_rng(seed)


def _conv_case(seed, B, H, W, Cin, Cout, k, **kw):
    g = _rng(seed)
    x = g.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (g.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = g.standard_normal(Cout).astype(np.float32)
    return x, w, b


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,kw", [
    (1, 12, 12, 32, 128, 3, {}),                                   # 32x128 tile (few workgroups)
    (2, 10, 6, 64, 160, 3, {}),                                    # ragged M and ragged Cout tile
    (1, 8, 8, 3, 128, 3, {}),                                      # conv_in: Cin = 3 (scalar staging)
    (1, 8, 8, 128, 3, 3, dict(clip=True)),                         # decoder out: Cout = 3, clip
    (1, 8, 8, 40, 64, 3, {}),                                      # Cin not a multiple of 32; 128x64 tile
    (1, 16, 16, 64, 64, 1, {}),                                    # 1x1 (quant_conv)
    (1, 16, 16, 32, 128, 3, dict(stride=2, pad=0, out_hw=(8, 8))), # Downsample
    (1, 6, 6, 32, 128, 3, dict(up_shift=1)),                       # Upsample
    (1, 48, 32, 64, 256, 3, {}),                                   # 24 tiles of 128x128: the B-direct generic kernel
    (1, 95, 65, 32, 128, 3, dict(stride=2, pad=0, out_hw=(47, 32))),  # ... with a stride and a ragged last tile (Downsample)
    (2, 21, 13, 3, 256, 3, {}),                                    # conv_cin4: ragged M, two channel tiles
    (1, 13, 9, 3, 128, 3, dict(stride=2, pad=0, out_hw=(6, 4))),   # conv_cin4 with a stride
    (2, 8, 32, 128, 3, 3, dict(clip=True)),                        # conv_patch_c128_out3
    (1, 4, 16, 128, 3, 3, dict(up_shift=1)),                       # ... over the upsampled image
])
def test_conv_bit_exact(B, H, W, Cin, Cout, k, kw):
    x, w, b = _conv_case(1, B, H, W, Cin, Cout, k)
    got = _emu.conv2d(x, w, b, **kw)
    ref = R.conv2d(x, w, b, **kw)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), np.abs(got - ref).max()


def test_conv_residual_and_no_bias():
    x, w, b = _conv_case(2, 1, 9, 9, 32, 128, 3)
    res = _rng(3).standard_normal((1, 9, 9, 128)).astype(np.float32)
    assert np.array_equal(_emu.conv2d(x, w, b, residual=res), R.conv2d(x, w, b, residual=res))
    assert np.array_equal(_emu.conv2d(x, w, None), R.conv2d(x, w, None))


def test_conv_large_tile_variant():
    """M*N large enough for the 128x128 workgroup tile (>= 256 tiles)."""
    x, w, b = _conv_case(4, 1, 128, 128, 8, 256, 3)
    got = _emu.conv2d(x, w, b)
    assert np.array_equal(got, R.conv2d(x, w, b))


@pytest.mark.parametrize("B,H,W,Cin,Cout,kw", [
    (1, 32, 48, 128, 128, {}),                    # conv_patch_c128: 24 tiles of 4x16, every image edge
    (2, 16, 48, 128, 256, {}),                    # two images, two channel tiles
    (1, 16, 24, 128, 128, dict(up_shift=1)),      # Upsample folded in: 32x48 output
    (1, 16, 16, 256, 256, {}),                    # 4 tiles of 4x16: too few for the machine -> generic kernel
    (1, 16, 48, 256, 256, {}),                    # conv_patch_c256: 12 tiles of 4x16 pixels x 256 channels
    (1, 8, 24, 256, 256, dict(up_shift=1)),       # c256 + Upsample
    (1, 16, 48, 256, 128, {}),                    # 256 -> 128 channels: generic kernel
])
def test_conv_patch_resident_bit_exact(B, H, W, Cin, Cout, kw):
    """3x3 stride-1 SAME with 128 / 256 input channels: the halo-patch kernel (vqgan_conv.h::conv_patch_body)
    must be bit-identical to the oracle -- same taps-then-channels order as the generic kernel."""
    x, w, b = _conv_case(11, B, H, W, Cin, Cout, 3)
    res = _rng(12).standard_normal((B, H << kw.get("up_shift", 0), W << kw.get("up_shift", 0), Cout)).astype(np.float32)
    got = _emu.conv2d(x, w, b, residual=res, **kw)
    assert np.array_equal(got, R.conv2d(x, w, b, residual=res, **kw)), np.abs(got - R.conv2d(x, w, b, residual=res, **kw)).max()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_conv_patch_resident_random_shapes(seed):
    """Seeded random geometry inside (and just outside) the patch kernels' domain: batch, image height a
    multiple of 4, width a multiple of 16 or not, 128 / 256 input channels, 128 / 256 / 384 outputs, upsample."""
    g = _rng(100 + seed)
    B = int(g.integers(1, 3))
    H = 4 * int(g.integers(2, 7))
    W = 16 * int(g.integers(1, 4)) + (8 if seed == 3 else 0)     # seed 3: width not a multiple of 16 -> generic kernel
    Cin = int(g.choice([128, 256]))
    Cout = int(g.choice([128, 256, 384]))
    up = int(g.integers(0, 2))
    kw = dict(up_shift=1) if up else {}
    x, w, b = _conv_case(200 + seed, B, H, W, Cin, Cout, 3)
    res = _rng(300 + seed).standard_normal((B, H << up, W << up, Cout)).astype(np.float32)
    assert np.array_equal(_emu.conv2d(x, w, b, residual=res, **kw), R.conv2d(x, w, b, residual=res, **kw)), (B, H, W, Cin, Cout, up)


@pytest.mark.parametrize("cus", [3, 5])
@pytest.mark.parametrize("B,H,W,Cin,Cout,kw", [
    (2, 16, 32, 128, 256, {}),                    # 8 pixel tiles of 8x16 x 2 channel tiles
    (1, 8, 16, 256, 256, dict(up_shift=1)),       # 8 tiles of 4x16, Upsample folded in
])
def test_conv_patch_persistent_walk(monkeypatch, cus, B, H, W, Cin, Cout, kw):
    """The patch kernels are persistent: one workgroup per CU walks the tiles blockIdx, blockIdx + gridDim, ...; the next
    tile's patch is requested before the current tile's results are written.  On a machine of 3 or 5 CUs a workgroup
    takes several tiles (ragged counts included), with and without the residual operand: bit-identical to the oracle."""
    monkeypatch.setenv("LWM_EMU_CUS", str(cus))
    x, w, b = _conv_case(21, B, H, W, Cin, Cout, 3)
    up = kw.get("up_shift", 0)
    res = _rng(22).standard_normal((B, H << up, W << up, Cout)).astype(np.float32)
    assert np.array_equal(_emu.conv2d(x, w, b, residual=res, **kw), R.conv2d(x, w, b, residual=res, **kw))
    assert np.array_equal(_emu.conv2d(x, w, b, **kw), R.conv2d(x, w, b, **kw))


@pytest.mark.parametrize("kw", [{}, dict(stride=2, pad=0, out_hw=(4, 12))])
def test_conv_bdirect_ranged_staging_across_images(capfd, monkeypatch, kw):
    """conv_igemm_128x128_bd stages by ranged buffer loads whose offsets are relative to the image of the tile's first pixel
    and whose out-of-image taps lie past the descriptor's range (zeros from the range check).  Images of 192 pixels: a
    128-pixel tile straddles two images, the last tile is ragged, later tiles start in later images -- bit-identical."""
    monkeypatch.setenv("LWM_EMU_CUS", "2")
    monkeypatch.setenv("LWM_EMU_TRACE", "1")
    x, w, b = _conv_case(41, 3, 8, 24, 64, 128, 3)
    Ho, Wo = kw.get("out_hw", (8, 24))
    res = _rng(42).standard_normal((3, Ho, Wo, 128)).astype(np.float32)
    got = _emu.conv2d(x, w, b, residual=res, **kw)
    names = [l.split()[1] for l in capfd.readouterr().err.splitlines() if l.startswith("emu-launch")]
    assert names == ["conv_igemm_128x128_bd"], names
    assert np.array_equal(got, R.conv2d(x, w, b, residual=res, **kw))


def test_conv_patch_resident_is_what_runs(capfd, monkeypatch):
    """The dispatch really takes the patch kernels for these shapes (the emulation traces its launches)."""
    monkeypatch.setenv("LWM_EMU_TRACE", "1")
    x, w, b = _conv_case(14, 1, 32, 48, 128, 128, 3)
    _emu.conv2d(x, w, b)
    x, w, b = _conv_case(15, 1, 16, 48, 256, 256, 3)
    _emu.conv2d(x, w, b)
    x, w, b = _conv_case(16, 1, 16, 16, 256, 256, 3)     # too few tiles for the machine: generic kernel
    _emu.conv2d(x, w, b)
    err = capfd.readouterr().err
    names = [l.split()[1] for l in err.splitlines() if l.startswith("emu-launch")]
    assert names[0] == "conv_patch_c128" and names[1] == "conv_patch_c256" and names[2].startswith("conv_igemm"), names
    x, w, b = _conv_case(14, 1, 32, 48, 128, 128, 3)     # with the residual operand: the form that prefetches it
    _emu.conv2d(x, w, b, residual=np.zeros((1, 32, 48, 128), np.float32))
    err = capfd.readouterr().err
    assert [l.split()[1] for l in err.splitlines() if l.startswith("emu-launch")] == ["conv_patch_c128_res"]


def test_conv_patch_resident_no_bias_clip():
    x, w, _ = _conv_case(13, 1, 32, 48, 128, 128, 3)
    x *= 4.0
    assert np.array_equal(_emu.conv2d(x, w, None, clip=True), R.conv2d(x, w, None, clip=True))


@pytest.mark.parametrize("C,HW,silu", [(128, 100, True), (256, 64, True), (512, 33, False), (768, 16, True)])
def test_groupnorm_silu_bit_exact(C, HW, silu):
    g = _rng(5)
    x = (g.standard_normal((2, HW, C)) * 2 + 0.3).astype(np.float32)
    gamma = (1 + 0.1 * g.standard_normal(C)).astype(np.float32)
    beta = (0.1 * g.standard_normal(C)).astype(np.float32)
    got = _emu.groupnorm(x, gamma, beta, silu=silu)
    ref = R.groupnorm(x, gamma, beta, silu=silu)
    assert np.array_equal(got, ref), np.abs(got - ref).max()


def test_groupnorm_many_slices():
    g = _rng(6)
    x = g.standard_normal((1, 70 * 70, 128)).astype(np.float32)
    gamma, beta = np.ones(128, np.float32), np.zeros(128, np.float32)
    assert np.array_equal(_emu.groupnorm(x, gamma, beta, silu=True), R.groupnorm(x, gamma, beta, silu=True))


@pytest.mark.parametrize("N,E", [(40, 512), (256, 1024)])
def test_vq_argmin_and_gather(N, E):
    g = _rng(7)
    cb = g.uniform(-1.0 / E, 1.0 / E, (E, 64)).astype(np.float32)
    z = (g.standard_normal((N, 64)) * 2.0 / E).astype(np.float32)
    z[3] = cb[5]                    # exact hit
    cb[9] = cb[4]                   # duplicate code: first index must win
    z[7] = cb[9]
    idx = _emu.vq_argmin(z, cb)
    ref = R.vq_argmin(z, cb)
    assert np.array_equal(idx, ref)
    assert idx[3] == 5 and idx[7] == 4
    assert np.array_equal(_emu.vq_gather(cb, idx), R.vq_gather(cb, ref))
    assert np.array_equal(_emu.vq_gather(cb, idx, z), R.vq_gather(cb, ref, z))


def test_validation_errors_are_loud():
    import ctypes as C
    from lwm_amd import _capi
    L = _emu.lib()
    a = _capi.LwmConvArgs()
    assert L.lwm_conv2d_nhwc_f32(C.byref(a), None) == _capi.LWM_EINVAL
    assert L.lwm_vq_argmin_f32(1, 1, 1, 1, 4, 8, 32, None) == _capi.LWM_EUNSUPPORTED
    assert L.lwm_groupnorm_silu_f32(16, 16, 16, 16, 16, 1, 4, 6, 3, 1e-6, 1, None) == _capi.LWM_EUNSUPPORTED
