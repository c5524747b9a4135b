"""Golden vectors produced by the REFERENCE'S OWN SOURCE LINES (not by a restatement).

Almost everything on the hot path lives in packages that cannot be imported here (jax, flax, ringattention, tux:
SURVEY.md section 8c), but four pieces of /root/reference are self-contained enough to be EXECUTED with numpy standing in
for the handful of `jax.numpy` / `jax.lax` names they use:

  precompute_freqs_cis        lwm/llama.py:344-350   module-level function (numpy code in the reference already)
  apply_rotary_emb            lwm/llama.py:353-375   module-level function
  RMSNorm._norm / .__call__   lwm/llama.py:334-341   methods of a flax Module; `self` = a plain object holding eps, dtype, weight
  VectorQuantizer.__call__    lwm/vqgan.py:191-221   method of a flax Module; `self.param(...)` returns the codebook handed in

This script cuts exactly those definitions out of the reference files with `ast` (the text is executed where it lies --
nothing is copied into the repo; decorators such as @nn.compact are dropped) and runs them.  Stand-ins, all one-to-one:
jnp.{asarray, reshape, stack, real, imag, square, sum, einsum, argmin, promote_types, float32} = numpy's;
jax.lax.complex(a, b) = a + 1j*b (complex64); jax.lax.rsqrt(x) = 1 / sqrt(x) in x's dtype; jax.lax.stop_gradient and
jax.device_put = identity; jax.nn.one_hot = an identity-matrix gather (its result is discarded by the reference).

What this PINS for the oracle, the product's host logic and the HIP kernels: the RoPE table formula, frequency dtype, pair
interleaving, reshape / stack order and position indexing (the call site's jnp.take, lwm/llama.py:515); RMSNorm's order of
casts and operations at dtype = float32 (the reference's default dtype; numpy has no bfloat16); the quantiser's distance
formula, first-index argmin, gather and output shapes.  What it does NOT pin: XLA's rounding and summation order (numpy
performs the arithmetic here) -- which is why the quantiser case is a WELL-CONDITIONED one (codes drawn N(0, 1), inputs near
codes: top-2 margins far above f32 rounding), where every summation order gives the same indices; with the reference's
random initialisation (codes U(-1/8192, 1/8192), lwm/vqgan.py:198-200) distances tie at f32 resolution and the index
depends on the order of additions, under XLA as under anything else.

Writes tests/golden/ref_run.npz.  Needs /root/reference (this container only); the tests read the .npz.
Re-run:  python tests/golden/gen_ref_run_golden.py
"""
import ast
import os
import types
from typing import Tuple

import numpy as np

REF = "/root/reference/lwm"
HERE = os.path.dirname(os.path.abspath(__file__))


def shims():
    jnp = types.SimpleNamespace(asarray=np.asarray, reshape=np.reshape, stack=np.stack, real=np.real, imag=np.imag,
                                square=np.square, sum=np.sum, einsum=np.einsum, argmin=np.argmin,
                                promote_types=np.promote_types, float32=np.float32, ndarray=np.ndarray, dtype=np.dtype)
    lax = types.SimpleNamespace(complex=lambda a, b: (a + 1j * b).astype(np.complex64),
                                rsqrt=lambda x: (1.0 / np.sqrt(x)).astype(x.dtype),
                                stop_gradient=lambda x: x)
    nn_ = types.SimpleNamespace(one_hot=lambda i, num_classes: np.eye(num_classes, dtype=np.float32)[i])
    jax = types.SimpleNamespace(lax=lax, numpy=jnp, nn=nn_, device_put=lambda x: x)
    return {"np": np, "jnp": jnp, "jax": jax, "Tuple": Tuple}


def cut(path, cls, name):
    """-> (function object compiled from the reference's text, first line, last line)"""
    src = open(path).read()
    body = ast.parse(src).body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    text = ast.get_source_segment(src, node)
    first = node.lineno
    if node.decorator_list:                       # @nn.compact: flax bookkeeping, not arithmetic
        text = text[text.index("def "):]
    import textwrap
    ns = shims()
    exec(compile(textwrap.dedent(" " * node.col_offset + text), f"{path}:{first}", "exec"), ns)
    return ns[name], first, node.end_lineno


def round_bf16(x):
    b = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def rope(out):
    pre, a0, a1 = cut(f"{REF}/llama.py", None, "precompute_freqs_cis")
    rot, b0, b1 = cut(f"{REF}/llama.py", None, "apply_rotary_emb")
    out["rope_lines"] = np.array([[a0, a1], [b0, b1]], np.int32)
    D, H = 128, 2
    g = np.random.default_rng(515)
    # theta / context pairs of the released models (README.md:112-117; lwm/llama.py:161 default)
    for tag, theta, max_pos in (("t1e4", 1e4, 32768), ("t1e7", 1e7, 262144), ("t5e7", 5e7, 1048576)):
        table = pre(D, max_pos, theta=theta, dtype=np.float32)
        assert table.dtype == np.complex64 and table.shape == (max_pos, D // 2)
        pos = np.unique(np.concatenate(([0, 1, 2, 3, 255, 1023, 4095, max_pos // 2 - 1, max_pos // 2, max_pos - 2, max_pos - 1],
                                        g.integers(0, max_pos, 21)))).astype(np.int32)
        position_ids = np.stack((pos, pos[::-1]))                       # (B = 2, S)
        xq = round_bf16(g.standard_normal((2, pos.size, H, D)).astype(np.float32))
        xk = round_bf16(g.standard_normal((2, pos.size, H, D)).astype(np.float32))
        freqs_cis = np.take(table, position_ids, axis=0)               # lwm/llama.py:515
        yq, yk = rot(xq, xk, freqs_cis=freqs_cis, dtype=np.float32)     # lwm/llama.py:517
        assert yq.dtype == np.float32 and yq.shape == xq.shape
        out.update({f"rope_{tag}_theta": np.float64(theta), f"rope_{tag}_max_pos": np.int64(max_pos), f"rope_{tag}_pos": position_ids,
                    f"rope_{tag}_rows": table[pos], f"rope_{tag}_xq": xq, f"rope_{tag}_xk": xk, f"rope_{tag}_yq": yq, f"rope_{tag}_yk": yk})
        del table


def rmsnorm(out):
    norm, a0, a1 = cut(f"{REF}/llama.py", "RMSNorm", "_norm")
    call, b0, b1 = cut(f"{REF}/llama.py", "RMSNorm", "__call__")
    out["rmsnorm_lines"] = np.array([[a0, a1], [b0, b1]], np.int32)
    g = np.random.default_rng(320)
    for tag, shape, eps in (("c4096", (3, 5, 4096), 1e-6), ("c256", (7, 256), 1e-5)):
        x = round_bf16((g.standard_normal(shape) * 2.0).astype(np.float32))
        w = round_bf16((1 + 0.1 * g.standard_normal(shape[-1])).astype(np.float32))
        self = types.SimpleNamespace(eps=eps, dtype=np.float32, param_dtype=np.float32, weight=w)
        self._norm = types.MethodType(norm, self)
        y = call(self, x)
        assert y.dtype == np.float32 and y.shape == x.shape
        out.update({f"rmsnorm_{tag}_x": x, f"rmsnorm_{tag}_w": w, f"rmsnorm_{tag}_eps": np.float64(eps), f"rmsnorm_{tag}_y": y})


def vq(out):
    call, a0, a1 = cut(f"{REF}/vqgan.py", "VectorQuantizer", "__call__")
    out["vq_lines"] = np.array([[a0, a1]], np.int32)
    g = np.random.default_rng(187)
    E, D = 8192, 64                                                    # lwm/vqgan.py:62-77 defaults
    codebook = g.standard_normal((E, D)).astype(np.float32)
    pick = g.integers(0, E, (2, 16, 16))
    pick[0, 0, :4] = (0, E - 1, 1, E - 2)
    z = (codebook[pick] + 0.05 * g.standard_normal((2, 16, 16, D))).astype(np.float32)
    self = types.SimpleNamespace(n_e=E, e_dim=D, param=lambda name, init, shape, dtype: codebook)
    z_q, idx = call(self, z)
    assert idx.shape == z.shape[:-1] and z_q.shape == z.shape and np.array_equal(idx, pick)
    looked_up = call(self, z, encoding_indices=idx)                    # the decode path, lwm/vqgan.py:204-205
    assert looked_up.shape == z.shape and np.array_equal(looked_up, codebook[idx])
    # top-2 margin of every row in float64: the case is well conditioned when it is far above f32 rounding of d (~ 64 * 2^-23)
    zf = z.reshape(-1, D).astype(np.float64)
    d = (zf ** 2).sum(1, keepdims=True) + (codebook.astype(np.float64) ** 2).sum(1)[None] - 2 * zf @ codebook.astype(np.float64).T
    part = np.partition(d, 1, axis=1)
    out.update({"vq_codebook": codebook, "vq_z": z, "vq_idx": idx.astype(np.int32), "vq_zq": np.asarray(z_q, np.float32),
                "vq_lookup": np.asarray(looked_up, np.float32), "vq_min_margin": np.float64((part[:, 1] - part[:, 0]).min())})


def main():
    out = {}
    rope(out)
    rmsnorm(out)
    vq(out)
    np.savez_compressed(os.path.join(HERE, "ref_run.npz"), **out)
    print("wrote ref_run.npz;", "lines", {k: out[k].tolist() for k in out if k.endswith("_lines")}, "vq margin", float(out["vq_min_margin"]))


if __name__ == "__main__":
    main()
