"""VQGAN video tokeniser on MI355X: the reference's `lwm.vqgan.VQGAN` surface
(lwm/vqgan.py:14-56) over the hand-written HIP primitives of liblwm_hip.so.

    vqgan = VQGAN(vqgan_checkpoint)                  # pickle of the flax param tree (:19)
    quantized, indices = vqgan.encode(pixel_values)  # (B,256,256,3) or (B,T,256,256,3) f32 in [-1,1]
    pixels = vqgan.decode(indices)                   # (..,16,16) int -> (..,256,256,3) f32 in [-1,1]

Same argument meaning and shapes as the reference; arrays are torch tensors on
the ROCm device (numpy / CPU tensors are accepted and uploaded).  There is no
CPU path: every op is a HIP kernel (lwm_amd/csrc/vqgan_*.h) and raises if the
shared library is missing.

The module structure below follows the flax modules and their auto-generated
parameter names: Encoder (:149-164), Decoder (:167-184), ResnetBlock
(:242-263), Downsample (:286-303), Upsample (:306-319), MidBlock (:340-351,
no attention: no_attn_mid_block=True, attn_resolutions=() -- :69-70),
VectorQuantizer (:187-221), VQGANModel.encode/decode (:117-141).
"""
from __future__ import annotations


import numpy as np
import torch

from . import ops


class VQGANConfig:
    """lwm/vqgan.py:59-102 (defaults identical)."""

    def __init__(self, resolution=256, num_channels=3, hidden_channels=128,
                 channel_mult=(1, 2, 2, 4, 6), num_res_blocks=2, attn_resolutions=(),
                 no_attn_mid_block=True, z_channels=64, num_embeddings=8192,
                 quantized_embed_dim=64, dropout=0.0, resample_with_conv=True,
                 commitment_cost=0.25):
        self.resolution = resolution
        self.num_channels = num_channels
        self.hidden_channels = hidden_channels
        self.channel_mult = tuple(channel_mult)
        self.num_res_blocks = num_res_blocks
        self.attn_resolutions = tuple(attn_resolutions)
        self.no_attn_mid_block = no_attn_mid_block
        self.z_channels = z_channels
        self.num_embeddings = num_embeddings
        self.quantized_embed_dim = quantized_embed_dim
        self.dropout = dropout
        self.resample_with_conv = resample_with_conv
        self.commitment_cost = commitment_cost
        self.num_resolutions = len(self.channel_mult)

    @classmethod
    def get_default_config(cls, updates=None):
        cfg = cls()
        for k, v in dict(updates or {}).items():
            if not hasattr(cfg, k):
                raise KeyError(f"unknown VQGANConfig field {k!r}")
            setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
        cfg.num_resolutions = len(cfg.channel_mult)
        return cfg

    def as_dict(self):
        return {k: getattr(self, k) for k in (
            "resolution", "num_channels", "hidden_channels", "channel_mult", "num_res_blocks",
            "attn_resolutions", "no_attn_mid_block", "z_channels", "num_embeddings",
            "quantized_embed_dim", "resample_with_conv")}


def random_params(config: VQGANConfig | None = None, seed: int = 0):
    """A random parameter tree with the reference's structure and names (numpy
    f32).  Kernels ~ N(0, 1/fan_in) (flax lecun_normal scale), codebook
    U(-1/n_e, 1/n_e) (lwm/vqgan.py:198-200); biases / GroupNorm affine are
    perturbed from flax's zeros/ones so that tests exercise them."""
    cfg = config or VQGANConfig()
    g = np.random.default_rng(seed)

    def conv(cin, cout, k=3):
        return {"kernel": (g.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32),
                "bias": (0.05 * g.standard_normal(cout)).astype(np.float32)}

    def gn(c):
        return {"scale": (1.0 + 0.1 * g.standard_normal(c)).astype(np.float32),
                "bias": (0.05 * g.standard_normal(c)).astype(np.float32)}

    def resnet(cin, cout):
        p = {"GroupNorm_0": gn(cin), "Conv_0": conv(cin, cout), "GroupNorm_1": gn(cout),
             "Conv_1": conv(cout, cout)}
        if cin != cout:
            p["Conv_2"] = conv(cin, cout, 1)
        return p

    hc, mult, nres = cfg.hidden_channels, cfg.channel_mult, cfg.num_resolutions
    enc = {"Conv_0": conv(cfg.num_channels, hc)}
    c = hc
    for lvl in range(nres):
        bp = {}
        out_c = hc * mult[lvl]
        for i in range(cfg.num_res_blocks):
            bp[f"ResnetBlock_{i}"] = resnet(c, out_c)
            c = out_c
        if lvl != nres - 1:
            bp["Downsample_0"] = {"Conv_0": conv(c, c)}
        enc[f"DownsamplingBlock_{lvl}"] = bp
    enc["MidBlock_0"] = {"ResnetBlock_0": resnet(c, c), "ResnetBlock_1": resnet(c, c)}
    enc["GroupNorm_0"] = gn(c)
    enc["Conv_1"] = conv(c, cfg.z_channels)

    c = hc * mult[nres - 1]
    dec = {"Conv_0": conv(cfg.z_channels, c)}
    dec["MidBlock_0"] = {"ResnetBlock_0": resnet(c, c), "ResnetBlock_1": resnet(c, c)}
    for lvl in reversed(range(nres)):
        bp = {}
        out_c = hc * mult[lvl]
        for i in range(cfg.num_res_blocks + 1):
            bp[f"ResnetBlock_{i}"] = resnet(c, out_c)
            c = out_c
        if lvl != 0:
            bp["Upsample_0"] = {"Conv_0": conv(c, c)}
        # flax names compact submodules by CREATION order and the reference Decoder creates them in
        # reversed(range(num_resolutions)) order (lwm/vqgan.py:180): the block of the deepest level is
        # UpsamplingBlock_0 (768 channels, has Upsample_0), the full-resolution one UpsamplingBlock_{nres-1}
        dec[f"UpsamplingBlock_{nres - 1 - lvl}"] = bp
    dec["GroupNorm_0"] = gn(c)
    dec["Conv_1"] = conv(c, cfg.num_channels)

    n_e, e_dim = cfg.num_embeddings, cfg.quantized_embed_dim
    return {
        "encoder": enc, "decoder": dec,
        "quantize": {"embeddings": g.uniform(-1.0 / n_e, 1.0 / n_e, (n_e, e_dim)).astype(np.float32)},
        "quant_conv": conv(cfg.z_channels, e_dim, 1),
        "post_quant_conv": conv(e_dim, cfg.z_channels, 1),
    }


def _to_device(tree, device):
    if isinstance(tree, dict):
        return {k: _to_device(v, device) for k, v in tree.items()}
    return torch.as_tensor(np.ascontiguousarray(np.asarray(tree), dtype=np.float32)).to(device).contiguous()


class VQGAN:
    def __init__(self, vqgan_checkpoint="", replicate=False, *, params=None, config=None, device=None):
        """`vqgan_checkpoint`: pickle of the flax parameter tree (lwm/vqgan.py:19), or pass
        `params` (the same nested dict) directly.  `replicate` (pmap over local devices in
        the reference, :20-24) is accepted for signature compatibility: frames are
        independent, so multi-GPU use is one VQGAN per process ("replicas only")."""
        if params is None:
            assert vqgan_checkpoint != ""
            from .weights import load_pickle_tree      # also reads pickles written from jax arrays
            params = load_pickle_tree(vqgan_checkpoint)
        if not torch.cuda.is_available():
            raise RuntimeError("lwm_amd.VQGAN needs a ROCm device (there is no CPU path)")
        self.replicate = replicate
        self.config = config or VQGANConfig.get_default_config()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.params = _to_device(params, self.device)
        self._codebook = self.params["quantize"]["embeddings"]
        self._se = ops.vq_sqnorm(self._codebook)
        self._ws = None

    # ---- building blocks
    def _gn_silu(self, p, x):
        need = ops.lib().lwm_groupnorm_workspace_bytes(x.shape[0], x.numel() // (x.shape[0] * x.shape[-1]),
                                                       x.shape[-1], 32)
        if self._ws is None or self._ws.numel() * 8 < need:
            self._ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device)
        return ops.groupnorm_silu(x, p["scale"], p["bias"], groups=32, eps=1e-6, silu=True,
                                  workspace=self._ws)

    @staticmethod
    def _conv(p, x, **kw):
        return ops.conv2d_nhwc(x, p["kernel"], p["bias"], **kw)

    def _resnet(self, p, x):
        h = self._gn_silu(p["GroupNorm_0"], x)
        h = self._conv(p["Conv_0"], h)
        h = self._gn_silu(p["GroupNorm_1"], h)
        res = self._conv(p["Conv_2"], x) if "Conv_2" in p else x
        return self._conv(p["Conv_1"], h, residual=res)

    def _mid(self, p, x):
        return self._resnet(p["ResnetBlock_1"], self._resnet(p["ResnetBlock_0"], x))

    def _encoder(self, p, x):
        cfg = self.config
        if x.shape[1] != cfg.resolution or x.shape[2] != cfg.resolution:
            raise AssertionError(tuple(x.shape))          # lwm/vqgan.py:154
        h = self._conv(p["Conv_0"], x)
        for lvl in range(cfg.num_resolutions):
            bp = p[f"DownsamplingBlock_{lvl}"]
            for i in range(cfg.num_res_blocks):
                h = self._resnet(bp[f"ResnetBlock_{i}"], h)
            if lvl != cfg.num_resolutions - 1:
                h = self._conv(bp["Downsample_0"]["Conv_0"], h, stride=2, pad=0,
                               out_hw=(h.shape[1] // 2, h.shape[2] // 2))
        h = self._mid(p["MidBlock_0"], h)
        h = self._gn_silu(p["GroupNorm_0"], h)
        return self._conv(p["Conv_1"], h)

    def _decoder(self, p, z):
        cfg = self.config
        h = self._conv(p["Conv_0"], z)
        h = self._mid(p["MidBlock_0"], h)
        for order, lvl in enumerate(reversed(range(cfg.num_resolutions))):
            bp = p[f"UpsamplingBlock_{order}"]            # creation-order auto-name (lwm/vqgan.py:180)
            for i in range(cfg.num_res_blocks + 1):
                h = self._resnet(bp[f"ResnetBlock_{i}"], h)
            if lvl != 0:
                h = self._conv(bp["Upsample_0"]["Conv_0"], h, up_shift=1)
        h = self._gn_silu(p["GroupNorm_0"], h)
        return self._conv(p["Conv_1"], h, clip=True)     # clip: VQGANModel.decode, :141

    def _as_dev(self, a, dtype):
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
        return t.to(device=self.device, dtype=dtype).contiguous()

    # ---- reference API
    @torch.no_grad()
    def encode(self, pixel_values):
        """-> (quantized_states (...,h,w,64) f32, codebook_indices (...,h,w) int32)."""
        x = self._as_dev(pixel_values, torch.float32)
        T = None
        if x.dim() == 5:                                   # video: fold T into batch (:119-121)
            T = x.shape[1]
            x = x.reshape((-1,) + tuple(x.shape[2:]))
        h = self._encoder(self.params["encoder"], x)
        h = self._conv(self.params["quant_conv"], h)
        idx = ops.vq_argmin(h, self._codebook, self._se)
        zq = ops.vq_gather(self._codebook, idx, z=h)
        if T is not None:
            zq = zq.reshape((-1, T) + tuple(zq.shape[1:]))
            idx = idx.reshape((-1, T) + tuple(idx.shape[1:]))
        return zq, idx

    @torch.no_grad()
    def decode(self, encoding, is_codebook_indices=True):
        """indices (...,h,w) -> pixels (...,H,W,3) f32 clipped to [-1,1] (:130-141)."""
        if is_codebook_indices:
            idx = self._as_dev(encoding, torch.int32)
            enc = ops.vq_gather(self._codebook, idx)
        else:
            enc = self._as_dev(encoding, torch.float32)
        T = None
        if enc.dim() == 5:
            T = enc.shape[1]
            enc = enc.reshape((-1,) + tuple(enc.shape[2:]))
        h = self._conv(self.params["post_quant_conv"], enc)
        out = self._decoder(self.params["decoder"], h)
        if T is not None:
            out = out.reshape((-1, T) + tuple(out.shape[1:]))
        return out
