"""CPU oracle for the VQGAN tokeniser -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under lwm_amd/ does.

The ARITHMETIC of conv / GroupNorm is this oracle's own contract (see oracle/vqgan_ref.c: XLA's is not reproducible);
the WIRING of the networks is pinned (round 5) to the reference's own module code: every class of lwm/vqgan.py:105-351 executed
under a minimal emulation of flax.linen.Module with the primitives below standing in for nn.Conv / nn.GroupNorm / nn.silu
(tests/golden/gen_ref_run_golden.py: MiniFlax) gives bit for bit what encode() / decode() below give
(tests/test_golden.py::test_vqgan_oracle_network_reproduces_the_reference_run).  The QUANTISER is pinned to a run of the
reference's own VectorQuantizer.__call__ (lwm/vqgan.py:192-221 executed where it lies with numpy standing in for
jax.numpy: tests/golden/gen_ref_run_golden.py -> tests/golden/ref_run.npz; vq_argmin / vq_gather below reproduce its
indices, its lookup and its straight-through forward value exactly, tests/test_golden.py).  The rest of lwm/vqgan.py is
flax code that cannot be executed here (the encoder + quantiser are checked against HF transformers'
ChameleonVQVAE and the decoder network against HF's JanusVQVAEDecoder, third-party
implementations of the same architecture: tests/golden/gen_hf_vqvae_golden.py,
tests/test_golden.py); the arithmetic primitives are restated in C (libvqgan_ref.so,
built by oracle/Makefile) and this module composes them exactly as the flax
modules of lwm/vqgan.py do, walking the same parameter tree
({'encoder': {'Conv_0': {'kernel','bias'}, 'DownsamplingBlock_0': {...}}, ...},
flax auto-names, lwm/vqgan.py:105-351).

All tensors are numpy float32, NHWC.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvqgan_ref.so")
_lib = None

# lwm/vqgan.py:62-77 (VQGANConfig defaults)
DEFAULT_CONFIG = dict(resolution=256, num_channels=3, hidden_channels=128,
                      channel_mult=(1, 2, 2, 4, 6), num_res_blocks=2, attn_resolutions=(),
                      no_attn_mid_block=True, z_channels=64, num_embeddings=8192,
                      quantized_embed_dim=64, resample_with_conv=True)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "vqgan_ref.c")
        if not os.path.exists(_SO) or (os.path.exists(src)
                                       and os.path.getmtime(src) > os.path.getmtime(_SO)):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        L = C.CDLL(_SO)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.ref_conv2d.argtypes = [fp, fp, fp, fp, fp] + [C.c_int] * 13
        L.ref_conv2d.restype = None
        L.ref_groupnorm.argtypes = [fp, fp, fp, fp, C.c_int, C.c_long, C.c_int, C.c_int, C.c_float,
                                    C.c_int]
        L.ref_groupnorm.restype = None
        L.ref_vq_argmin.argtypes = [fp, fp, ip, C.c_long, C.c_int, C.c_int]
        L.ref_vq_argmin.restype = None
        L.ref_vq_gather.argtypes = [fp, ip, fp, fp, C.c_long, C.c_int]
        L.ref_vq_gather.restype = None
        L.ref_silu_array.argtypes = [fp, fp, C.c_long]
        L.ref_silu_array.restype = None
        L.ref_expf_scalar.argtypes = [C.c_float]
        L.ref_expf_scalar.restype = C.c_float
        _lib = L
    return _lib


def _f(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv2d(x, w, bias=None, residual=None, *, stride=1, pad=None, up_shift=0, out_hw=None,
           clip=False):
    """nn.Conv on NHWC x with HWIO kernel w.  pad=None means flax 'SAME' for
    stride 1 ((k-1)//2 each side).  Downsample (lwm/vqgan.py:291-300) is
    stride=2, pad=0, out_hw=(H//2, W//2): the pad of one row/column at the
    bottom/right is the zero fill outside the input."""
    x, w = _c(x), _c(w)
    B, Hin, Win, Cin = x.shape
    KH, KW, Cin2, Cout = w.shape
    assert Cin2 == Cin
    if pad is None:
        pad = (KH - 1) // 2
    Hv, Wv = Hin << up_shift, Win << up_shift
    if out_hw is None:
        out_hw = ((Hv + 2 * pad - KH) // stride + 1, (Wv + 2 * pad - KW) // stride + 1)
    Ho, Wo = out_hw
    y = np.empty((B, Ho, Wo, Cout), np.float32)
    bias = None if bias is None else _c(bias)
    residual = None if residual is None else _c(residual)
    if residual is not None:
        assert residual.shape == y.shape
    lib().ref_conv2d(_f(x), _f(w), _f(bias), _f(residual), _f(y), B, Hin, Win, Cin, Cout, KH, KW,
                     stride, pad, up_shift, Ho, Wo, int(clip))
    return y


def groupnorm(x, gamma, beta, *, groups=32, eps=1e-6, silu=False):
    x = _c(x)
    B, C_ = x.shape[0], x.shape[-1]
    HW = int(np.prod(x.shape[1:-1]))
    y = np.empty_like(x)
    lib().ref_groupnorm(_f(x), _f(_c(gamma)), _f(_c(beta)), _f(y), B, HW, C_, groups, eps,
                        int(silu))
    return y


def vq_argmin(z, codebook):
    z, codebook = _c(z), _c(codebook)
    D = z.shape[-1]
    N = z.size // D
    idx = np.empty(z.shape[:-1], np.int32)
    lib().ref_vq_argmin(_f(z), _f(codebook), idx.ctypes.data_as(C.POINTER(C.c_int32)), N,
                        codebook.shape[0], D)
    return idx


def vq_gather(codebook, idx, z=None):
    codebook = _c(codebook)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    D = codebook.shape[1]
    out = np.empty(idx.shape + (D,), np.float32)
    z = None if z is None else _c(z)
    lib().ref_vq_gather(_f(codebook), idx.ctypes.data_as(C.POINTER(C.c_int32)), _f(z), _f(out),
                        idx.size, D)
    return out


def silu(x):
    """nn.silu with the arithmetic of the fused GroupNorm + SiLU above (x * (1 / (1 + exp(-x))), the oracle's exp)."""
    x = _c(x)
    y = np.empty_like(x)
    lib().ref_silu_array(_f(x), _f(y), x.size)
    return y


def expf(x):
    return np.float32(lib().ref_expf_scalar(float(np.float32(x))))


# ---------------------------------------------------------------- model (lwm/vqgan.py)
def _conv(p, x, **kw):
    return conv2d(x, p["kernel"], p["bias"], **kw)


def _gn_silu(p, x):
    return groupnorm(x, p["scale"], p["bias"], silu=True)


def resnet_block(p, x):
    """lwm/vqgan.py:242-263 (use_conv_shortcut=False, dropout deterministic)."""
    h = _gn_silu(p["GroupNorm_0"], x)
    h = _conv(p["Conv_0"], h)
    h = _gn_silu(p["GroupNorm_1"], h)
    res = x
    if "Conv_2" in p:                       # out_channels != in_channels: 1x1 shortcut
        res = _conv(p["Conv_2"], x)
    return conv2d(h, p["Conv_1"]["kernel"], p["Conv_1"]["bias"], residual=res)


def mid_block(p, x):
    """lwm/vqgan.py:340-351 with no_attn_mid_block=True."""
    return resnet_block(p["ResnetBlock_1"], resnet_block(p["ResnetBlock_0"], x))


def encoder(p, x, cfg):
    """lwm/vqgan.py:149-164."""
    assert x.shape[1] == x.shape[2] == cfg["resolution"], x.shape
    h = _conv(p["Conv_0"], x)
    nres = len(cfg["channel_mult"])
    for lvl in range(nres):
        bp = p[f"DownsamplingBlock_{lvl}"]
        for i in range(cfg["num_res_blocks"]):
            h = resnet_block(bp[f"ResnetBlock_{i}"], h)
        if lvl != nres - 1:                 # lwm/vqgan.py:237, Downsample :286-303
            dp = bp["Downsample_0"]["Conv_0"]
            h = conv2d(h, dp["kernel"], dp["bias"], stride=2, pad=0,
                       out_hw=(h.shape[1] // 2, h.shape[2] // 2))
    h = mid_block(p["MidBlock_0"], h)
    h = _gn_silu(p["GroupNorm_0"], h)
    return _conv(p["Conv_1"], h)


def decoder(p, z, cfg, clip=True):
    """lwm/vqgan.py:167-184; the final clip is VQGANModel.decode's (:141)."""
    h = _conv(p["Conv_0"], z)
    h = mid_block(p["MidBlock_0"], h)
    nres = len(cfg["channel_mult"])
    for order, lvl in enumerate(reversed(range(nres))):
        # flax auto-names follow creation order, which is reversed(range(nres)) (lwm/vqgan.py:180):
        # UpsamplingBlock_0 is the deepest level
        bp = p[f"UpsamplingBlock_{order}"]
        for i in range(cfg["num_res_blocks"] + 1):
            h = resnet_block(bp[f"ResnetBlock_{i}"], h)
        if lvl != 0:                        # Upsample: nearest x2 + conv (:306-319)
            up = bp["Upsample_0"]["Conv_0"]
            h = conv2d(h, up["kernel"], up["bias"], up_shift=1)
    h = _gn_silu(p["GroupNorm_0"], h)
    return conv2d(h, p["Conv_1"]["kernel"], p["Conv_1"]["bias"], clip=clip)


def encode(params, pixel_values, cfg=None):
    """VQGANModel.encode (lwm/vqgan.py:117-128): returns (quantized, indices)."""
    cfg = cfg or DEFAULT_CONFIG
    x = _c(pixel_values)
    T = None
    if x.ndim == 5:
        T = x.shape[1]
        x = x.reshape((-1,) + x.shape[2:])
    h = encoder(params["encoder"], x, cfg)
    h = _conv(params["quant_conv"], h)
    cb = params["quantize"]["embeddings"]
    idx = vq_argmin(h, cb)
    zq = vq_gather(cb, idx, z=h)
    if T is not None:
        zq = zq.reshape((-1, T) + zq.shape[1:])
        idx = idx.reshape((-1, T) + idx.shape[1:])
    return zq, idx


def index_mismatch_report(params, pixel_values, got_idx, cfg=None):
    """when code indices differ from the oracle's, say how many and how close the
    call was -- for every mismatching position the oracle's best and second-best squared distances and the
    distance of the index the device chose.  -> (n_mismatch, n_total, report string)."""
    cfg = cfg or DEFAULT_CONFIG
    x = _c(pixel_values)
    if x.ndim == 5:
        x = x.reshape((-1,) + x.shape[2:])
    h = _conv(params["quant_conv"], encoder(params["encoder"], x, cfg))
    cb = np.asarray(params["quantize"]["embeddings"], np.float32)
    ref = vq_argmin(h, cb).reshape(-1)
    got = np.asarray(got_idx).reshape(-1)
    z = h.reshape(-1, h.shape[-1]).astype(np.float64)
    bad = np.nonzero(ref != got)[0]
    lines = []
    for i in bad[:8]:
        d = ((z[i][None] - cb.astype(np.float64)) ** 2).sum(-1)
        o = np.argsort(d)
        lines.append(f"pos {i}: oracle {ref[i]} (d={d[ref[i]]:.9g}), device {got[i]} (d={d[got[i]]:.9g}), "
                     f"top-2 margin {d[o[1]] - d[o[0]]:.3g}")
    return len(bad), len(ref), f"{len(bad)}/{len(ref)} code indices differ; " + "; ".join(lines)


def decode(params, encoding, cfg=None, is_codebook_indices=True):
    """VQGANModel.decode (lwm/vqgan.py:130-141)."""
    cfg = cfg or DEFAULT_CONFIG
    enc = np.asarray(encoding)
    if is_codebook_indices:
        enc = vq_gather(params["quantize"]["embeddings"], enc)
    T = None
    if enc.ndim == 5:
        T = enc.shape[1]
        enc = enc.reshape((-1,) + enc.shape[2:])
    h = _conv(params["post_quant_conv"], enc)
    out = decoder(params["decoder"], h, cfg, clip=True)
    if T is not None:
        out = out.reshape((-1, T) + out.shape[1:])
    return out
