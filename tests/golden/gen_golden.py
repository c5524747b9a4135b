"""Generates tests/golden/*.npz from the oracles (NOT from the reference: the
reference cannot be executed here and ships no vectors -- see DESIGN.md section 6).
Purpose: pin the oracles so they cannot drift silently, and give the GPU tests
fixed known-answer cases.  Re-run:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import attention_ref as A  # noqa: E402
from oracle import vqgan_ref as V  # noqa: E402
from lwm_amd.vqgan import VQGANConfig, random_params  # noqa: E402


def attention_case():
    g = np.random.default_rng(2024)
    B, S, H, D = 1, 96, 2, 128
    q, k, v, do = (A.round_bf16(g.standard_normal((B, S, H, D)).astype(np.float32)) for _ in range(4))
    seg = np.zeros((B, S), np.int32)
    seg[:, 40:] = 1
    seg[:, 77:] = 2
    kv = np.ones((B, S), np.uint8)
    kv[:, 5:9] = 0
    kw = dict(causal=True, seg_q=seg, seg_k=seg, key_valid=kv)
    out, lse = A.dense_attention(q, k, v, **kw)
    dq, dk, dv = A.dense_attention_bwd(q, k, v, do, **kw)
    np.savez_compressed(os.path.join(HERE, "attention_packed_s96.npz"), q=q, k=k, v=v, do=do, seg=seg,
                        key_valid=kv, out=out.astype(np.float32), lse=lse.astype(np.float32),
                        dq=dq.astype(np.float32), dk=dk.astype(np.float32), dv=dv.astype(np.float32))


def vqgan_case():
    g = np.random.default_rng(7)
    x = g.standard_normal((1, 6, 5, 128)).astype(np.float32)
    w = (g.standard_normal((3, 3, 128, 128)) / 34).astype(np.float32)
    b = g.standard_normal(128).astype(np.float32)
    gamma = (1 + 0.1 * g.standard_normal(128)).astype(np.float32)
    beta = (0.1 * g.standard_normal(128)).astype(np.float32)
    cb = g.uniform(-1 / 256, 1 / 256, (256, 64)).astype(np.float32)
    y = V.conv2d(x, w, b)
    yd = V.conv2d(x[:, :6, :4], w, b, stride=2, pad=0, out_hw=(3, 2))
    yu = V.conv2d(x, w, b, up_shift=1)
    h = V.groupnorm(y, gamma, beta, silu=True)
    z = np.ascontiguousarray(h[..., :64] * 0.01)
    idx = V.vq_argmin(z, cb)
    np.savez_compressed(os.path.join(HERE, "vqgan_primitives.npz"), x=x, w=w, b=b, gamma=gamma, beta=beta,
                        cb=cb, conv=y, conv_down=yd, conv_up=yu, gn_silu=h, z=z, idx=idx,
                        zq=V.vq_gather(cb, idx, z))
    # whole model, small config, parameters from the seeded generator
    cfg = VQGANConfig.get_default_config(dict(resolution=32, channel_mult=(1, 2, 4), num_embeddings=1024))
    params = random_params(cfg, seed=5)
    px = np.random.default_rng(6).uniform(-1, 1, (1, 32, 32, 3)).astype(np.float32)
    zq, idx = V.encode(params, px, cfg.as_dict())
    rec = V.decode(params, idx, cfg.as_dict())
    np.savez_compressed(os.path.join(HERE, "vqgan_model_res32.npz"), px=px, idx=idx, zq=zq, rec=rec)


if __name__ == "__main__":
    attention_case()
    vqgan_case()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
