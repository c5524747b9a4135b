"""Golden vectors from HF transformers' ChameleonVQVAE encoder + vector quantiser -- a third-party PyTorch
implementation of the SAME taming-style VQGAN encoder that lwm/vqgan.py builds in flax (conv_in, per level
ResnetBlocks [GroupNorm(32, eps 1e-6) -> swish -> conv3x3, twice; 1x1 nin shortcut when the channel count
changes], Downsample = zero pad (0,1,0,1) + conv3x3 stride 2, mid block without attention, GroupNorm -> swish
-> conv_out, 1x1 quant_conv, argmin of |z|^2 + |e|^2 - 2 z.e) -- run HERE on CPU in float64.

The reference's flax modules cannot be imported in this image; this anchors the oracle's encoder + quantiser
(layer order, padding side of the Downsample, GroupNorm grouping / eps, the shortcut rule, HWIO kernels,
the argmin) against code that is not ours.  The DECODER network is anchored the same way on HF's JanusVQVAEDecoder
(conv_in, mid block, per level num_res_blocks + 1 ResnetBlocks and nearest x2 + conv3x3 upsampling in reversed
level order, GroupNorm -> swish -> conv_out): Janus puts attention blocks into the mid block and the deepest
level, which lwm's configuration does not have (no_attn_mid_block, attn_resolutions = ()) -- each is
`x + proj_out(attn(norm(x)))`, so zeroing proj_out makes it the identity and the rest is lwm's decoder.

    python tests/golden/gen_hf_vqvae_golden.py      # needs `transformers`; writes hf_vqvae_tiny.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import hf_vqvae_fixture as F  # noqa: E402

CFG = F.CFG


def _oihw(w):                       # flax HWIO -> torch OIHW
    return torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1)))


def _put_conv(m, p):
    m.weight.copy_(_oihw(p["kernel"]))
    m.bias.copy_(torch.from_numpy(p["bias"]))


def _put_gn(m, p):
    m.weight.copy_(torch.from_numpy(p["scale"]))
    m.bias.copy_(torch.from_numpy(p["bias"]))


def _put_resnet(m, p):
    _put_gn(m.norm1, p["GroupNorm_0"]); _put_conv(m.conv1, p["Conv_0"])
    _put_gn(m.norm2, p["GroupNorm_1"]); _put_conv(m.conv2, p["Conv_1"])
    assert (m.in_channels != m.out_channels) == ("Conv_2" in p)
    if "Conv_2" in p:
        _put_conv(m.nin_shortcut, p["Conv_2"])


def load_lwm_tree(model, tree):
    """The flax auto-named parameter tree of lwm/vqgan.py (the one oracle/vqgan_ref.py walks) -> the HF module tree."""
    enc, e = model.encoder, tree["encoder"]
    _put_conv(enc.conv_in, e["Conv_0"])
    for lvl, down in enumerate(enc.down):
        bp = e[f"DownsamplingBlock_{lvl}"]
        for i, b in enumerate(down.block):
            _put_resnet(b, bp[f"ResnetBlock_{i}"])
        assert hasattr(down, "downsample") == ("Downsample_0" in bp)
        if hasattr(down, "downsample"):
            _put_conv(down.downsample.conv, bp["Downsample_0"]["Conv_0"])
    _put_resnet(enc.mid.block_1, e["MidBlock_0"]["ResnetBlock_0"])
    _put_resnet(enc.mid.block_2, e["MidBlock_0"]["ResnetBlock_1"])
    _put_gn(enc.norm_out, e["GroupNorm_0"])
    _put_conv(enc.conv_out, e["Conv_1"])
    _put_conv(model.quant_conv, tree["quant_conv"])
    model.quantize.embedding.weight.copy_(torch.from_numpy(tree["quantize"]["embeddings"]))


def main():
    import transformers
    from transformers import ChameleonVQVAEConfig
    from transformers.models.chameleon.modeling_chameleon import ChameleonVQVAE
    cfg = ChameleonVQVAEConfig(base_channels=CFG["hidden_channels"], channel_multiplier=list(CFG["channel_mult"]),
                               num_res_blocks=CFG["num_res_blocks"], resolution=CFG["resolution"], in_channels=3,
                               latent_channels=CFG["z_channels"], embed_dim=CFG["quantized_embed_dim"],
                               num_embeddings=CFG["num_embeddings"], attn_resolutions=None, attn_type="none",
                               double_latent=False, dropout=0.0)
    model = ChameleonVQVAE(cfg).eval().double()
    n_before = sum(p.numel() for p in model.parameters())
    with torch.no_grad():
        load_lwm_tree(model, F.lwm_tree())
    n_tree = sum(v.size for v in _leaves(F.lwm_tree()))
    n_unused = sum(p.numel() for p in model.post_quant_conv.parameters())      # decoder side of HF's module: not in lwm's encode
    assert n_before == n_tree + n_unused, (n_before, n_tree, n_unused)          # every HF parameter was overwritten
    x = torch.from_numpy(F.pixels())
    with torch.no_grad():
        h = model.quant_conv(model.encoder(x.double().permute(0, 3, 1, 2)))                          # NCHW float64
        z = h.permute(0, 2, 3, 1).reshape(-1, CFG["quantized_embed_dim"])
        e = model.quantize.embedding.weight
        d = (z ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * z @ e.T
        best2 = torch.topk(d, 2, dim=1, largest=False)
        idx = best2.indices[:, 0]
        _, _, hf_idx = model.quantize(h)
        assert torch.equal(hf_idx.reshape(-1), idx)                                                   # HF's own argmin
        gap = (best2.values[:, 1] - best2.values[:, 0])
    side = h.shape[-1]
    out = {"z": h.permute(0, 2, 3, 1).numpy(),                                                       # float64, pre-quantisation
           "indices": idx.reshape(2, side, side).numpy().astype(np.int32),
           "gap": gap.reshape(2, side, side).numpy(),
           "decoder_out": janus_decoder_out(),                                                        # float64, before the clip
           "transformers_version": np.array(transformers.__version__)}
    dst = os.path.join(HERE, "hf_vqvae_tiny.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes; min argmin gap", float(gap.min()), "median", float(gap.median()))


def janus_decoder_out():
    from transformers.models.janus.configuration_janus import JanusVQVAEConfig
    from transformers.models.janus.modeling_janus import JanusVQVAEAttnBlock, JanusVQVAEDecoder
    cfg = JanusVQVAEConfig(base_channels=CFG["hidden_channels"], channel_multiplier=list(CFG["channel_mult"]),
                           num_res_blocks=CFG["num_res_blocks"], latent_channels=CFG["z_channels"], in_channels=3,
                           out_channels=3, embed_dim=CFG["quantized_embed_dim"], num_embeddings=CFG["num_embeddings"],
                           double_latent=False, dropout=0.0)
    dec = JanusVQVAEDecoder(cfg).eval().double()
    tree = F.decoder_tree()["decoder"]
    attn_params = 0
    with torch.no_grad():
        _put_conv(dec.conv_in, tree["Conv_0"])
        _put_resnet(dec.mid.block_1, tree["MidBlock_0"]["ResnetBlock_0"])
        _put_resnet(dec.mid.block_2, tree["MidBlock_0"]["ResnetBlock_1"])
        for order, up in enumerate(dec.up):                      # HF appends in reversed-level order too
            bp = tree[f"UpsamplingBlock_{order}"]
            for i, b in enumerate(up.block):
                _put_resnet(b, bp[f"ResnetBlock_{i}"])
            assert hasattr(up, "upsample") == ("Upsample_0" in bp)
            if hasattr(up, "upsample"):
                _put_conv(up.upsample.conv, bp["Upsample_0"]["Conv_0"])
        _put_gn(dec.norm_out, tree["GroupNorm_0"])
        _put_conv(dec.conv_out, tree["Conv_1"])
        for m in dec.modules():                                  # attention blocks -> identity
            if isinstance(m, JanusVQVAEAttnBlock):
                m.proj_out.weight.zero_()
                m.proj_out.bias.zero_()
                attn_params += sum(p.numel() for p in m.parameters())
    n_hf = sum(p.numel() for p in dec.parameters())
    n_tree = sum(v.size for v in _leaves(tree))
    assert n_hf == n_tree + attn_params, (n_hf, n_tree, attn_params)     # every other HF parameter was overwritten
    z = torch.from_numpy(F.latents()).double().permute(0, 3, 1, 2)
    with torch.no_grad():
        y = dec(z)
    return y.permute(0, 2, 3, 1).numpy()


def _leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from _leaves(v)
        else:
            yield v


if __name__ == "__main__":
    main()
