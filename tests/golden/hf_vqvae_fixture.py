"""The tiny VQGAN shared by tests/golden/gen_hf_vqvae_golden.py (which loads it into HF transformers'
ChameleonVQVAE encoder / JanusVQVAE decoder) and tests/test_golden.py (which feeds it to the oracle and the HIP kernels): configuration and
seeded float32 parameters in the flax auto-named tree of lwm/vqgan.py.  numpy's PCG64 stream is stable across
versions, so the parameters are regenerated instead of stored."""
import numpy as np

# a miniature of lwm/vqgan.py:62-77 (the HIP GroupNorm wants 4 channels per group: 128 is the narrowest width)
CFG = dict(resolution=32, num_channels=3, hidden_channels=128, channel_mult=(1, 2, 2), num_res_blocks=2,
           attn_resolutions=(), no_attn_mid_block=True, z_channels=64, num_embeddings=256,
           quantized_embed_dim=64, resample_with_conv=True)
SEED = 20240911


def _makers(g):
    def conv(cin, cout, k=3):
        return {"kernel": (g.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32),
                "bias": (0.1 * g.standard_normal(cout)).astype(np.float32)}

    def gn(c):
        return {"scale": (1.0 + 0.2 * g.standard_normal(c)).astype(np.float32), "bias": (0.1 * g.standard_normal(c)).astype(np.float32)}

    def resnet(cin, cout):
        p = {"GroupNorm_0": gn(cin), "Conv_0": conv(cin, cout), "GroupNorm_1": gn(cout), "Conv_1": conv(cout, cout)}
        if cin != cout:
            p["Conv_2"] = conv(cin, cout, 1)
        return p

    return conv, gn, resnet


def lwm_tree():
    g = np.random.default_rng(SEED)
    conv, gn, resnet = _makers(g)
    hc, mult = CFG["hidden_channels"], CFG["channel_mult"]
    enc = {"Conv_0": conv(CFG["num_channels"], hc)}
    c = hc
    for lvl, m in enumerate(mult):
        bp = {}
        for i in range(CFG["num_res_blocks"]):
            bp[f"ResnetBlock_{i}"] = resnet(c, hc * m)
            c = hc * m
        if lvl != len(mult) - 1:
            bp["Downsample_0"] = {"Conv_0": conv(c, c)}
        enc[f"DownsamplingBlock_{lvl}"] = bp
    enc["MidBlock_0"] = {"ResnetBlock_0": resnet(c, c), "ResnetBlock_1": resnet(c, c)}
    enc["GroupNorm_0"] = gn(c)
    enc["Conv_1"] = conv(c, CFG["z_channels"])
    return {"encoder": enc, "quant_conv": conv(CFG["z_channels"], CFG["quantized_embed_dim"], 1),
            "quantize": {"embeddings": g.standard_normal((CFG["num_embeddings"], CFG["quantized_embed_dim"])).astype(np.float32)}}


def pixels():
    g = np.random.default_rng(SEED + 1)
    return (g.random((2, CFG["resolution"], CFG["resolution"], 3)) * 2 - 1).astype(np.float32)      # NHWC, as lwm feeds it


def decoder_tree():
    """'post_quant_conv' and 'decoder' of the same model (their own random stream: the encoder's values above stay
    what the committed encoder vectors were made with).  UpsamplingBlock_<order> in flax creation order =
    reversed levels (lwm/vqgan.py:180)."""
    g = np.random.default_rng(SEED + 2)
    conv, gn, resnet = _makers(g)
    hc, mult = CFG["hidden_channels"], CFG["channel_mult"]
    c = hc * mult[-1]
    dec = {"Conv_0": conv(CFG["z_channels"], c),
           "MidBlock_0": {"ResnetBlock_0": resnet(c, c), "ResnetBlock_1": resnet(c, c)}}
    for order, lvl in enumerate(reversed(range(len(mult)))):
        bp = {}
        for i in range(CFG["num_res_blocks"] + 1):
            bp[f"ResnetBlock_{i}"] = resnet(c, hc * mult[lvl])
            c = hc * mult[lvl]
        if lvl != 0:
            bp["Upsample_0"] = {"Conv_0": conv(c, c)}
        dec[f"UpsamplingBlock_{order}"] = bp
    dec["GroupNorm_0"] = gn(c)
    dec["Conv_1"] = conv(c, CFG["num_channels"])
    return {"post_quant_conv": conv(CFG["quantized_embed_dim"], CFG["z_channels"], 1), "decoder": dec}


def latents():
    g = np.random.default_rng(SEED + 3)
    side = CFG["resolution"] // 2 ** (len(CFG["channel_mult"]) - 1)
    return g.standard_normal((2, side, side, CFG["z_channels"])).astype(np.float32)                  # NHWC decoder input
