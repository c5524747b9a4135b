"""Golden vectors produced by the REFERENCE'S OWN SOURCE LINES (not by a restatement).

Almost everything on the hot path lives in packages that cannot be imported here (jax, flax, ringattention, tux:
SURVEY.md section 8c), but four pieces of /root/reference are self-contained enough to be EXECUTED with numpy standing in
for the handful of `jax.numpy` / `jax.lax` names they use:

  precompute_freqs_cis        lwm/llama.py:344-350   module-level function (numpy code in the reference already)
  apply_rotary_emb            lwm/llama.py:353-375   module-level function
  RMSNorm._norm / .__call__   lwm/llama.py:334-341   methods of a flax Module; `self` = a plain object holding eps, dtype, weight
  VectorQuantizer.__call__    lwm/vqgan.py:191-221   method of a flax Module; `self.param(...)` returns the codebook handed in
  the MASK statements of FlaxLLaMAAttention  (the in-tree specification of SURVEY.md section 8 row a4):
      setup:     self.causal_mask = make_causal_mask(...)                                   lwm/llama.py:425
      __call__:  blockwise branch, attention_mask -> additive key-padding bias               lwm/llama.py:526-537
                 dense branch, causal (with the cache's shift) AND segment AND key mask      lwm/llama.py:573-592
    These are statement RANGES inside larger methods (the rest of the methods builds flax layers and calls ringattention):
    the ranges are located in the syntax tree by what they assign, compiled as they are and executed with the locals the
    method would hold (xq, xk, hidden_states, attention_mask, segment_ids; `self` = a plain object with has_variable /
    variables / causal_mask / config / dtype).  Two flax helpers they call are stood in from flax's documented behaviour
    (flax 0.8.4, pinned in gpu_requirements.txt): combine_masks(*masks) = logical AND of the masks that are not None, cast
    to float32; make_causal_mask(x, dtype) = (1, 1, L, L) with [q, k] = q >= k.

This script cuts exactly those definitions out of the reference files with `ast` (the text is executed where it lies --
nothing is copied into the repo; decorators such as @nn.compact are dropped) and runs them.  Stand-ins, all one-to-one:
jnp.{asarray, reshape, stack, real, imag, square, sum, einsum, argmin, promote_types, float32} = numpy's;
jax.lax.complex(a, b) = a + 1j*b (complex64); jax.lax.rsqrt(x) = 1 / sqrt(x) in x's dtype; jax.lax.stop_gradient and
jax.device_put = identity; jax.nn.one_hot = an identity-matrix gather (its result is discarded by the reference).

What this PINS for the oracle, the product's host logic and the HIP kernels: the RoPE table formula, frequency dtype, pair
interleaving, reshape / stack order and position indexing (the call site's jnp.take, lwm/llama.py:515); RMSNorm's order of
casts and operations at dtype = float32 (the reference's default dtype; numpy has no bfloat16); the quantiser's distance
formula, first-index argmin, gather and output shapes; the boolean visibility of every (query, key) pair in the training
and the cached-inference branch and the bias constants the blockwise branch hands to ringattention.  What it does NOT pin: XLA's rounding and summation order (numpy
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
    # the mask statements (lwm/llama.py:425, :526-537, :573-592)
    jnp.expand_dims, jnp.full, jnp.finfo, jnp.arange, jnp.broadcast_to, jnp.ones = (
        np.expand_dims, np.full, np.finfo, np.arange, np.broadcast_to, np.ones)
    lax.select = lambda pred, a, b: np.where(pred, a, b)

    def combine_masks(*masks, dtype=np.float32):          # flax.linen.combine_masks
        masks = [m for m in masks if m is not None]
        assert masks and all(m.ndim == masks[0].ndim for m in masks)
        out = masks[0].astype(bool)
        for m in masks[1:]:
            out = np.logical_and(out, m.astype(bool))
        return out.astype(dtype)

    def make_causal_mask(x, extra_batch_dims=0, dtype=np.float32):   # flax.linen.make_causal_mask
        idxs = np.broadcast_to(np.arange(x.shape[-1], dtype=np.int32), x.shape)
        return np.greater_equal(idxs[..., None], idxs[..., None, :])[..., None, :, :].astype(dtype)

    return {"np": np, "jnp": jnp, "jax": jax, "lax": lax, "Tuple": Tuple, "combine_masks": combine_masks,
            "make_causal_mask": make_causal_mask}


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


def statements(path, cls, method, pick):
    """-> (code object of a run of consecutive statements of cls.method chosen by pick(list of statements of the method,
    source text), first line, last line)"""
    src = open(path).read()
    c = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == cls)
    f = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == method)
    stmts = pick(f, src)
    mod = ast.Module(body=stmts, type_ignores=[])
    return compile(mod, f"{path}:{stmts[0].lineno}", "exec"), stmts[0].lineno, stmts[-1].end_lineno


def assigns(node, name):
    return isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name or
                                                isinstance(t, ast.Attribute) and t.attr == name for t in node.targets)


def masks(out):
    path = f"{REF}/llama.py"
    seg = lambda src, n: ast.get_source_segment(src, n)

    def the_if(f, src):       # `if self.config.scan_attention and xq.shape[1] > max(...)`: blockwise branch / dense branch
        return next(n for n in ast.walk(f) if isinstance(n, ast.If) and "scan_attention" in seg(src, n.test))

    def upto(body, name):     # the statements of a branch up to and including the LAST assignment of `name` before attn_weights
        stop = next(i for i, n in enumerate(body) if assigns(n, "attn_weights"))
        last = max(i for i, n in enumerate(body[:stop]) if assigns(n, name))
        return body[:last + 1]

    setup_code, s0, s1 = statements(path, "FlaxLLaMAAttention", "setup", lambda f, src: [n for n in f.body if assigns(n, "causal_mask")])
    bias_code, b0, b1 = statements(path, "FlaxLLaMAAttention", "__call__", lambda f, src: upto(the_if(f, src).body, "attention_bias"))
    # dense branch: everything before the comment "During fast autoregressive decoding" = up to attention_mask = combine_masks(...)
    def dense_pick(f, src):
        body = the_if(f, src).orelse
        last = max(i for i, n in enumerate(body) if assigns(n, "attention_mask") and "combine_masks" in seg(src, n))
        return body[:last + 1]
    dense_code, d0, d1 = statements(path, "FlaxLLaMAAttention", "__call__", dense_pick)
    out["mask_lines"] = np.array([[s0, s1], [b0, b1], [d0, d1]], np.int32)

    g = np.random.default_rng(572)
    L = 48                                                         # config.max_sequence_length of this case
    ns = shims()
    self = types.SimpleNamespace(config=types.SimpleNamespace(max_sequence_length=L), dtype=np.float32, variables={})
    self.has_variable = lambda col, name: col in self.variables and name in self.variables[col]
    exec(setup_code, dict(ns, self=self, config=self.config))
    assert self.causal_mask.shape == (1, 1, L, L) and self.causal_mask.dtype == bool

    def packed(B, S):
        segs = np.zeros((B, S), np.int32)
        for b in range(B):
            cuts = np.sort(g.choice(np.arange(1, S), 3, replace=False))
            for c in cuts:
                segs[b, c:] += 1
        am = np.ones((B, S), np.int32)
        am[0, :3] = 0                                              # left padding (lwm/vision_chat.py:138-140)
        am[1, 10:13] = 0
        return segs, am

    # (1) training, dense branch (S <= chunk): causal AND same segment AND key mask
    B, S = 2, 40
    segs, am = packed(B, S)
    loc = dict(ns, self=self, xq=np.zeros((B, S, 2, 4), np.float32), xk=np.zeros((B, S, 2, 4), np.float32),
               hidden_states=np.zeros((B, S, 8), np.float32), attention_mask=am, segment_ids=segs, init_cache=False)
    exec(dense_code, loc)
    m = loc["attention_mask"]
    assert m.shape == (B, 1, S, S)
    out.update({"mask_train_seg": segs, "mask_train_am": am, "mask_train": m.astype(np.float32)})
    # (1b) the same without segment_ids
    loc = dict(ns, self=self, xq=np.zeros((B, S, 2, 4), np.float32), xk=np.zeros((B, S, 2, 4), np.float32),
               hidden_states=np.zeros((B, S, 8), np.float32), attention_mask=am, segment_ids=None, init_cache=False)
    exec(dense_code, loc)
    out["mask_train_noseg"] = loc["attention_mask"].astype(np.float32)
    # (2) cached inference: a prefill block of Q tokens at cache_index, then one decode token; the key mask covers max_length
    K = 32
    amk = np.ones((B, K), np.int32)
    amk[0, :2] = 0
    for tag, Q, idx in (("prefill", 6, 5), ("decode", 1, 17), ("first", 8, 0)):
        self.variables = {"cache": {"cache_index": np.int32(idx), "cached_key": np.zeros((B, K, 2, 4), np.float32)}}
        loc = dict(ns, self=self, xq=np.zeros((B, Q, 2, 4), np.float32), xk=np.zeros((B, Q, 2, 4), np.float32),
                   hidden_states=np.zeros((B, Q, 8), np.float32), attention_mask=amk, segment_ids=None, init_cache=False)
        exec(dense_code, loc)
        m = loc["attention_mask"]
        assert m.shape == (B, 1, Q, K)
        out.update({f"mask_{tag}": m.astype(np.float32), f"mask_{tag}_index": np.int32(idx)})
    out["mask_cache_am"] = amk
    # (3) blockwise branch: the additive key-padding bias handed to ringattention, at the reference's default dtype
    self.variables = {}
    loc = dict(ns, self=self, xq=np.zeros((B, S, 2, 4), np.float32), xk=np.zeros((B, S, 2, 4), np.float32),
               attention_mask=am, segment_ids=segs, init_cache=False)
    exec(bias_code, loc)
    bias = loc["attention_bias"]
    assert bias.shape == (B, 1, 1, S) and bias.dtype == np.float32
    out["mask_bias"] = bias


def main():
    out = {}
    rope(out)
    rmsnorm(out)
    vq(out)
    masks(out)
    np.savez_compressed(os.path.join(HERE, "ref_run.npz"), **out)
    print("wrote ref_run.npz;", "lines", {k: out[k].tolist() for k in out if k.endswith("_lines")}, "vq margin", float(out["vq_min_margin"]))


if __name__ == "__main__":
    main()
