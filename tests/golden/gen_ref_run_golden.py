"""Golden vectors produced by EXECUTING THE REFERENCE'S OWN SOURCE (not a restatement of it).

The reference as a whole cannot run here: jax, flax, ringattention and tux are not installable.  But
most of what sits on and around the hot path is plain Python over `jax.numpy` calls, and THAT can run: this script cuts
functions, methods, statement ranges and whole class hierarchies out of the files under /root/reference with `ast`,
compiles the text as it is (nothing is copied into the repo; the file name and line of every code object are the
reference's) and executes it with stand-ins for what is missing.  TEST INFRASTRUCTURE: only tests/test_golden.py reads the
result.

What is executed (function below -> reference lines; the committed .npz records the line ranges, the tests pin them):

  rope()               precompute_freqs_cis, apply_rotary_emb                      lwm/llama.py:344-375
  rmsnorm()            RMSNorm._norm / __call__ (dtype float32)                    lwm/llama.py:334-341
  vq()                 VectorQuantizer.__call__ (encode and lookup paths)          lwm/vqgan.py:192-221
  masks()              FlaxLLaMAAttention: causal_mask (setup), the blockwise branch's bias, the dense branch's mask with and
                       without a cache -- statement RANGES inside larger methods, located in the syntax tree by what they
                       assign and executed with the locals the method would hold   lwm/llama.py:425, :527-537, :572-592
  cache()              FlaxLLaMAAttention._concatenate_to_cache, creation + three prefill blocks     lwm/llama.py:441-492
  cache_decode()       the same method's one-token branch (:452-483) over an emulated 4-device "sp" axis: only the shard that
                       owns row cache_index writes
  vision_text()        the embedding choice of FlaxVideoLLaMAModule.__call__       lwm/vision_llama.py:308-311
                       the 'vision,text' objective of train_step                   lwm/train.py:185-202
  video()              VQGANModel.encode / decode over the oracle's networks       lwm/vqgan.py:117-141
  layer()              FlaxLLaMABlock.__call__, FlaxLLaMAAttention.__call__ (both branches), _split_heads / _merge_heads,
                       FlaxLLaMAMLP.__call__, RMSNorm, RoPE composed as the reference composes them
                                                                                   lwm/llama.py:434-438, :494-620, :658-661, :704-744
  network()            EVERY module class of lwm/vqgan.py (VQGANModel ... MidBlock) as class definitions under MiniFlax
                                                                                   lwm/vqgan.py:105-351
  model()              the model classes of lwm/llama.py (RMSNorm, Attention, MLP, Block, BlockCollection in its loop and its
                       nn.scan form, Module, ForCausalLMModule) and of lwm/vision_llama.py as class definitions under
                       MiniFlax: a 2-layer model produces logits from a train state in either on-disk layout
                                                                                   lwm/llama.py:320-1106, lwm/vision_llama.py:255-439
  cached_model()       the same model classes in cached inference: init_cache, a prefill, greedy one-token steps through
                       _concatenate_to_cache and the shifted mask                  lwm/llama.py:440-620, :806-824
  chat_prompt()        Sampler._process_frame / _read_process_vision / construct_input   lwm/vision_chat.py:59-145
  generation_inputs()  prepare_inputs_for_generation / update_inputs_for_generation of both model classes
                                                                                   lwm/vision_llama.py:447-474, lwm/llama.py:1113-1137
  flags()              the define_flags_with_default(...) statements of the three entry points (with a recorder), the keyword
                       defaults of LLaMAConfig / VideoLLaMAConfig / VQGANConfig, the size table LLAMA_STANDARD_CONFIGS

Stand-ins, by kind:
  * numpy for the jax.numpy / jax.lax names the code uses, one to one (shims()): asarray, reshape, stack, real, imag, square,
    sum, einsum, argmin, take, where, pad, clip, cumsum, ...; lax.complex(a, b) = a + 1j*b in complex64; lax.rsqrt = 1/sqrt in
    the argument's dtype; lax.select = where; lax.dynamic_update_slice = a copy with the block written at the start index,
    clamped so that it fits (XLA's documented semantics); stop_gradient / device_put / with_sharding_constraint = identity;
    shard_map(fn, ...) = fn (one device; cache_decode() emulates an n-device "sp" axis: per-device slices by the in_specs,
    lax.axis_index = the device, outputs concatenated), lax.cond(p, t, f) = t() if p else f(), x.at[i].set(v) = a copy with x[i] = v;
    jax.image.resize(method='nearest') = an integer repeat.
  * flax: MiniFlax (below, ~150 lines) emulates the part of flax.linen the classes use -- Module with dataclass-style fields
    from the annotations, setup(), @nn.compact, the NAMING RULE (a submodule made in setup() is named by its attribute, one
    made inside a compact __call__ is ClassName_<n> with n counting that class within the parent, name= overrides), param(),
    nn.scan as lwm/llama.py:927-941 uses it (`length` applications over leaves stacked on the scan axis), remat = identity;
    nn.Dense = x @ kernel (every Dense on the path has use_bias=False), nn.Embed = a lookup, nn.Dropout = identity
    (deterministic), nn.silu = x * sigmoid(x); combine_masks = logical AND of the masks that are not None cast to float32,
    make_causal_mask = (1, 1, L, L) with [q, k] = q >= k (flax 0.8.4's documented behaviour, gpu_requirements.txt).
  * what is NOT in the reference tree at all and is therefore NOT pinned by these vectors -- the arithmetic of
      - the `ringattention` package: `ringattention(...)` is the oracle's f32 blockwise restatement (after asserting the
        keyword arguments the call site passes), `ringattention_inference(...)` the oracle's dense-mask restatement,
        `blockwise_feedforward(module, x, chunk, pre_remat=True)` = module(x);
      - XLA's convolution / GroupNorm: nn.Conv / nn.GroupNorm in network() are the oracle's C primitives, fed the leaves the
        naming rule leads to;
      - tux.cross_entropy_loss_and_accuracy: a recorder that returns the oracle's restatement.
    numpy also performs the sums and products that XLA would: rounding and summation order are numpy's.  That is why the
    quantiser cases are WELL CONDITIONED (codes N(0, 1) or planted on the inputs: top-2 margins far above f32 rounding), where
    every summation order finds the same index; with the reference's random initialisation (codes U(-1/8192, 1/8192),
    lwm/vqgan.py:198-200) distances tie at f32 resolution and the index depends on the order of additions, under XLA as
    under anything else.

What the vectors pin, then, is everything a restatement can get wrong that is not rounding: formulas, operand order, casts,
pair interleaving, reshape / stack order, position and cache indexing, first-index argmin, mask semantics and bias constants,
which branch runs when and what it hands the op, residual structure, module wiring, parameter names and layouts, token
layouts, flag names and defaults.

Writes tests/golden/ref_run.npz.  Needs /root/reference (this container only); the tests read the .npz.
Re-run:  python tests/golden/gen_ref_run_golden.py [output.npz]   (deterministic: tests/test_golden.py regenerates and compares
when /root/reference is present)
"""
import ast
import os
import sys
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

    from typing import Optional
    jax.Array = np.ndarray
    return {"np": np, "jnp": jnp, "jax": jax, "lax": lax, "Tuple": Tuple, "Optional": Optional, "combine_masks": combine_masks,
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


def cache(out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    fn, a0, a1 = cut(f"{REF}/llama.py", "FlaxLLaMAAttention", "_concatenate_to_cache")
    out["cache_lines"] = np.array([[a0, a1]], np.int32)

    def dynamic_update_slice(operand, update, start):          # jax.lax.dynamic_update_slice: clamped start, copy
        res = np.array(operand, copy=True)
        st = [int(np.clip(int(s0), 0, d - u)) for s0, d, u in zip(start, operand.shape, update.shape)]
        res[tuple(slice(a, a + u) for a, u in zip(st, update.shape))] = update
        return res
    fn.__globals__["lax"].dynamic_update_slice = dynamic_update_slice
    fn.__globals__["jnp"].zeros = np.zeros
    fn.__globals__["jnp"].array = np.array
    fn.__globals__["jnp"].int32 = np.int32

    class Var:                                                 # flax's self.variable(collection, name, init_fn, *args)
        def __init__(self, value):
            self.value = value
    store = {}
    self = types.SimpleNamespace(config=None)
    self.has_variable = lambda col, name: (col, name) in store
    def variable(col, name, init, *args):
        if (col, name) not in store:
            store[(col, name)] = Var(init(*args))
        return store[(col, name)]
    self.variable = variable
    g = np.random.default_rng(440)
    B, L, H, D = 2, 16, 2, 4
    am = np.ones((B, L), np.int32)
    # init_cache: the module runs once on max_length rows (lwm/llama.py:876-895) -- creates zeros, returns its arguments
    k0, v0 = np.ones((B, L, H, D), np.float32), np.ones((B, L, H, D), np.float32)
    rk, rv, rm = fn(self, k0, v0, np.zeros((B, L, H, D), np.float32), am)
    assert rk is k0 and rv is v0 and rm is am and not store[("cache", "cached_key")].value.any()
    steps = []
    for Q in (5, 3, 8):                                        # prefill blocks at cache_index 0, 5, 8 (the last one ends at max_length)
        key, value = (g.standard_normal((B, Q, H, D)).astype(np.float32) for _ in range(2))
        idx = int(store[("cache", "cache_index")].value)
        rk, rv, rm = fn(self, key, value, np.zeros((B, Q, H, D), np.float32), am)
        assert rk.shape == (B, L, H, D) and rm is am                  # attention runs over the WHOLE cache: kv_len = max_length
        steps.append((idx, key, value, rk.copy(), rv.copy(), int(store[("cache", "cache_index")].value)))
    for i, (idx, key, value, ck, cv, nxt) in enumerate(steps):
        out.update({f"cache_{i}_index": np.int32(idx), f"cache_{i}_key": key, f"cache_{i}_value": value,
                    f"cache_{i}_k": ck, f"cache_{i}_v": cv, f"cache_{i}_next": np.int32(nxt)})
    out["cache_steps"] = np.int32(len(steps))


def cache_decode(out):
    """The one-token branch of _concatenate_to_cache (lwm/llama.py:452-483): the cache is sharded over the "sp" mesh axis and only
    the shard that owns row cache_index writes.  jax's shard_map is emulated for an n-device "sp" axis -- the function runs once
    per device on the slices its in_specs give it (dimension 1 split where the spec names 'sp'), jax.lax.axis_index('sp') is that
    device's index, the outputs are concatenated back along 'sp' -- lax.cond(p, t, f) = t() if p else f(), x.at[i].set(v) = a
    copy with x[i] = v."""
    fn, a0, a1 = cut(f"{REF}/llama.py", "FlaxLLaMAAttention", "_concatenate_to_cache")
    out["cache_decode_lines"] = np.array([[a0, a1]], np.int32)
    n = 4

    class At(np.ndarray):
        @property
        def at(self):
            arr = self

            class Ix:
                def __getitem__(self, idx):
                    class Set:
                        def set(self, v):
                            res = arr.copy()
                            res[idx] = v
                            return res
                    return Set()
            return Ix()
    state = {"rank": 0}

    def shard_map(f, mesh=None, in_specs=None, out_specs=None, check_rep=None):
        def run(*args):
            outs = []
            for r in range(n):
                state["rank"] = r
                loc = []
                for a, spec in zip(args, in_specs):
                    if len(spec) > 1 and spec[1] == "sp":
                        c = a.shape[1] // n
                        a = a[:, r * c:(r + 1) * c]
                    loc.append(a)
                outs.append(f(*loc))
            return tuple(np.concatenate([o[i] for o in outs], axis=1).view(At) for i in range(len(out_specs)))
        return run

    def dus(operand, update, start):
        res = operand.copy()
        st = [int(np.clip(int(s0), 0, d - u)) for s0, d, u in zip(start, operand.shape, update.shape)]
        res[tuple(slice(a, a + u) for a, u in zip(st, update.shape))] = update
        return res
    glob = fn.__globals__
    glob["lax"].dynamic_update_slice = dus
    glob["jnp"].zeros = lambda shape, dtype=None: np.zeros(shape, dtype).view(At)
    glob["jnp"].array, glob["jnp"].int32, glob["jnp"].logical_and = np.array, np.int32, np.logical_and
    glob["jax"].lax.axis_index = lambda axis: state["rank"]
    glob["jax"].lax.cond = lambda p, t, f: t() if bool(p) else f()
    glob.update(shard_map=shard_map, PS=lambda *a: a, LLaMAConfig=types.SimpleNamespace(
        get_jax_mesh=lambda mesh_dim: types.SimpleNamespace(shape={"sp": n})))

    class Var:
        def __init__(self, value):
            self.value = value
    store = {}
    self = types.SimpleNamespace(config=types.SimpleNamespace(mesh_dim="1,1,1,4"))
    self.has_variable = lambda col, name: (col, name) in store

    def variable(col, name, init, *args):
        if (col, name) not in store:
            store[(col, name)] = Var(init(*args))
        return store[(col, name)]
    self.variable = variable
    g = np.random.default_rng(452)
    B, L, H, D = 2, 16, 2, 4
    am = np.ones((B, L), np.int32)
    fn(self, np.ones((B, L, H, D), np.float32), np.ones((B, L, H, D), np.float32), np.zeros((B, L, H, D), np.float32), am)     # creation
    key, value = (g.standard_normal((B, 6, H, D)).astype(np.float32) for _ in range(2))
    fn(self, key, value, np.zeros((B, 6, H, D), np.float32), am)                                   # a prefill block: rows 0..5
    out.update({"cache_decode_prefill_key": key, "cache_decode_prefill_value": value})
    steps = []
    for _ in range(5):                                                                            # rows 6..10: shards 1 and 2 of 4
        key, value = (g.standard_normal((B, 1, H, D)).astype(np.float32) for _ in range(2))
        idx = int(store[("cache", "cache_index")].value)
        rk, rv, _ = fn(self, key, value, np.zeros((B, 1, H, D), np.float32), am)
        assert rk.shape == (B, L, H, D) and np.array_equal(rk[:, idx], key[:, 0])
        steps.append((idx, key, value, np.asarray(rk).copy(), np.asarray(rv).copy(), int(store[("cache", "cache_index")].value)))
    for i, (idx, key, value, ck, cv, nxt) in enumerate(steps):
        out.update({f"cache_decode_{i}_index": np.int32(idx), f"cache_decode_{i}_key": key, f"cache_decode_{i}_value": value,
                    f"cache_decode_{i}_k": ck, f"cache_decode_{i}_v": cv, f"cache_decode_{i}_next": np.int32(nxt)})
    out.update({"cache_decode_steps": np.int32(len(steps)), "cache_decode_ranks": np.int32(n)})


def vision_text(out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import llama_ops_ref as R
    g = np.random.default_rng(307)
    # (1) the embedding choice, lwm/vision_llama.py:308-311
    def emb_pick(f, src):
        iff = next(n for n in f.body if isinstance(n, ast.If) and "input_ids.shape[1] == 1" in ast.get_source_segment(src, n.test))
        return iff.orelse
    code, e0, e1 = statements(f"{REF}/vision_llama.py", "FlaxVideoLLaMAModule", "__call__", emb_pick)
    B, S, V, VV, d = 2, 12, 11, 7, 8
    wte, vte = g.standard_normal((V, d)).astype(np.float32), g.standard_normal((VV, d)).astype(np.float32)
    vm = g.random((B, S)) < 0.5
    ids = np.where(vm, g.integers(0, VV, (B, S)), g.integers(0, V, (B, S))).astype(np.int32)
    ns = shims()
    ns["jnp"].where = np.where
    loc = dict(ns, self=types.SimpleNamespace(wte=lambda i: wte[i], vte=lambda i: vte[i]), input_ids=ids, vision_masks=vm)
    exec(code, loc)
    out.update({"vt_embed_lines": np.array([[e0, e1]], np.int32), "vt_wte": wte, "vt_vte": vte, "vt_ids": ids, "vt_vm": vm,
                "vt_embeds": np.asarray(loc["input_embeds"], np.float32)})
    # (2) the objective, lwm/train.py:185-202
    src = open(f"{REF}/train.py").read()
    f = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "loss_and_accuracy")
    top = next(n for n in f.body if isinstance(n, ast.If))
    branch = top.orelse[0]
    assert "vision,text" in ast.get_source_segment(src, branch.test)
    last = max(i for i, n in enumerate(branch.body) if assigns(n, "loss"))
    stmts = branch.body[:last + 1]
    code = compile(ast.Module(body=stmts, type_ignores=[]), f"{REF}/train.py:{stmts[0].lineno}", "exec")
    out["vt_loss_lines"] = np.array([[stmts[0].lineno, stmts[-1].end_lineno]], np.int32)
    vl, tl = g.standard_normal((B, S, VV)).astype(np.float32), g.standard_normal((B, S, V)).astype(np.float32)
    tvm = g.random((B, S)) < 0.4
    tgt = np.where(tvm, g.integers(0, VV, (B, S)), g.integers(0, V, (B, S))).astype(np.int32)
    lm = (g.random((B, S)) < 0.8).astype(np.float32)
    calls, applied = [], []

    def ce(logits, tokens, valid=None):
        calls.append((np.asarray(logits), np.asarray(tokens), np.asarray(valid)))
        loss, acc, _ = R.cross_entropy_loss_and_accuracy(logits, tokens, valid)
        return loss, acc

    def apply(params, *args, **kw):
        applied.append(args)
        return types.SimpleNamespace(logits=(vl, tl))
    batch = {"input_tokens": ids, "input_vision_masks": vm, "target_tokens": tgt, "target_vision_masks": tvm, "loss_masks": lm}
    loc = dict(ns, params=None, batch=batch, model=types.SimpleNamespace(apply=apply), cross_entropy_loss_and_accuracy=ce,
               rng_generator=lambda keys: None, llama_config=types.SimpleNamespace(rng_keys=lambda: ()))
    exec(code, loc)
    assert len(calls) == 2 and calls[0][0] is vl and calls[1][0] is tl and applied[0][0] is ids and applied[0][1] is vm
    out.update({"vt_vision_logits": vl, "vt_text_logits": tl, "vt_targets": tgt, "vt_tvm": tvm, "vt_loss_masks": lm,
                "vt_vision_targets": calls[0][1], "vt_vision_valid": calls[0][2].astype(np.float32),
                "vt_text_targets": calls[1][1], "vt_text_valid": calls[1][2].astype(np.float32),
                "vt_loss": np.float64(loc["loss"]), "vt_vision_loss": np.float64(loc["vision_loss"]),
                "vt_text_loss": np.float64(loc["text_loss"]), "vt_vision_acc": np.float64(loc["vision_acc"]),
                "vt_text_acc": np.float64(loc["text_acc"])})


def video(out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import vqgan_ref as V
    from lwm_amd.vqgan import VQGANConfig, random_params
    enc, a0, a1 = cut(f"{REF}/vqgan.py", "VQGANModel", "encode")
    dec, b0, b1 = cut(f"{REF}/vqgan.py", "VQGANModel", "decode")
    dec.__globals__["jnp"].clip = np.clip
    out["video_lines"] = np.array([[a0, a1], [b0, b1]], np.int32)
    cfgo = VQGANConfig.get_default_config(dict(resolution=32, channel_mult=(1, 2, 4), num_embeddings=1024))
    params = random_params(cfgo, seed=11)
    cfg = cfgo.as_dict()
    # make the decoder overshoot [-1, 1] so that the clip has something to do
    params["decoder"]["Conv_1"]["kernel"] = params["decoder"]["Conv_1"]["kernel"] * 40.0
    cb = params["quantize"]["embeddings"]

    def quantize(z, encoding_indices=None):                    # the oracle's quantiser (the reference's own: vq(), above)
        if encoding_indices is not None:
            return V.vq_gather(cb, encoding_indices)
        idx = V.vq_argmin(z, cb)
        return V.vq_gather(cb, idx, z=z), idx
    self = types.SimpleNamespace(encoder=lambda x: V.encoder(params["encoder"], x, cfg),
                                 quant_conv=lambda h: V._conv(params["quant_conv"], h), quantize=quantize,
                                 post_quant_conv=lambda h: V._conv(params["post_quant_conv"], h),
                                 decoder=lambda h: V.decoder(params["decoder"], h, cfg, clip=False))
    g = np.random.default_rng(117)
    px = g.uniform(-1, 1, (2, 3, 32, 32, 3)).astype(np.float32)       # (B, T, H, W, C): a video
    zq, idx = enc(self, px)
    rec = dec(self, idx)
    assert idx.shape == (2, 3, 8, 8) and zq.shape == (2, 3, 8, 8, cfg["quantized_embed_dim"]) and rec.shape == px.shape
    frac = float(np.mean(np.abs(rec) == 1.0))
    assert 0.01 < frac < 0.99, frac
    zq4, idx4 = enc(self, px[:, 0])                                   # (B, H, W, C): images
    out.update({"video_px": px, "video_zq": zq, "video_idx": idx.astype(np.int32), "video_rec": rec, "video_seed": np.int32(11),
                "video_kernel_gain": np.float32(40.0), "video_zq_image": zq4, "video_idx_image": idx4.astype(np.int32),
                "video_clipped_fraction": np.float64(frac)})


def layer(out):
    import functools
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import attention_ref as A
    path = f"{REF}/llama.py"
    pre, _, _ = cut(path, None, "precompute_freqs_cis")
    rot, _, _ = cut(path, None, "apply_rotary_emb")
    norm, _, _ = cut(path, "RMSNorm", "_norm")
    norm_call, _, _ = cut(path, "RMSNorm", "__call__")
    split, h0, h1 = cut(path, "FlaxLLaMAAttention", "_split_heads")
    merge, m0, m1 = cut(path, "FlaxLLaMAAttention", "_merge_heads")
    attn, a0, a1 = cut(path, "FlaxLLaMAAttention", "__call__")
    mlp, f0, f1 = cut(path, "FlaxLLaMAMLP", "__call__")
    block, b0, b1 = cut(path, "FlaxLLaMABlock", "__call__")
    out["layer_lines"] = np.array([[h0, h1], [m0, m1], [a0, a1], [f0, f1], [b0, b1]], np.int32)
    setup_code, _, _ = statements(path, "FlaxLLaMAAttention", "setup", lambda f, src: [n for n in f.body if assigns(n, "causal_mask")])

    g = np.random.default_rng(664)
    B, S, H, D, F, L = 2, 24, 2, 16, 48, 32
    d = H * D
    W = {n: (g.standard_normal(shp) * 0.2).astype(np.float32) for n, shp in
         (("wq", (d, d)), ("wk", (d, d)), ("wv", (d, d)), ("wo", (d, d)), ("w1", (d, F)), ("w2", (F, d)), ("w3", (d, F)))}
    W["attention_norm"] = (1 + 0.1 * g.standard_normal(d)).astype(np.float32)
    W["ffn_norm"] = (1 + 0.1 * g.standard_normal(d)).astype(np.float32)
    x = g.standard_normal((B, S, d)).astype(np.float32)
    am = np.ones((B, S), np.int32)
    am[0, :3] = 0
    seg = np.zeros((B, S), np.int32)
    seg[1, 9:] = 1
    seg[1, 17:] = 2
    pos = np.tile(np.arange(S, dtype=np.int32), (B, 1))
    record, specs = [], []

    def ringattention(q, k, v, attn_bias, segment_ids, axis_name=None, float32_logits=None, cache_idx=None, blockwise_kwargs=None):
        kw = blockwise_kwargs
        assert axis_name == "sp" and float32_logits is True and cache_idx is None and kw["causal_block_size"] == 1
        assert attn_bias.shape == (B, 1, 1, S) and set(np.unique(attn_bias)) <= {0.0, float(np.finfo(np.float32).min)}
        record.append(("ringattention", kw["query_chunk_size"], kw["key_chunk_size"], segment_ids is not None))
        return A.blockwise_ring_attention(q, k, v, ring=1, q_chunk=kw["query_chunk_size"], k_chunk=kw["key_chunk_size"], causal=True,
                                          segment_ids=segment_ids, key_valid=(attn_bias[:, 0, 0] == 0).astype(np.uint8))

    def ringattention_inference(q, k, v, attn_mask, axis_name=None):
        assert axis_name == "sp" and attn_mask.shape == (B, 1, S, S)
        record.append(("ringattention_inference",))
        return A.ring_inference(q, k, v, attn_mask[:, 0])

    def blockwise_feedforward(module, xx, chunk_size, pre_remat=None):
        assert pre_remat is True
        record.append(("blockwise_feedforward", int(chunk_size)))
        return module(xx)

    def build(theta, q_chunk, k_chunk, scan_mlp, mlp_chunk):
        cfg = types.SimpleNamespace(max_sequence_length=L, scan_attention=True, scan_query_chunk_size=q_chunk, scan_key_chunk_size=k_chunk,
                                    attn_pdrop=0.0, mesh_dim="1,1,1,1", scan_layers=False, scan_mlp=scan_mlp, scan_mlp_chunk_size=mlp_chunk)
        ns = shims()
        ns["jnp"].take, ns["jnp"].where = np.take, np.where
        sa = types.SimpleNamespace(config=cfg, dtype=np.float32, precision=None, embed_dim=d, num_heads=H, head_dim=D, variables={},
                                   wq=lambda t: t @ W["wq"], wk=lambda t: t @ W["wk"], wv=lambda t: t @ W["wv"], wo=lambda t: t @ W["wo"],
                                   resid_dropout=lambda t, deterministic=True: t, has_variable=lambda c, n: False)
        sa._split_heads, sa._merge_heads = types.MethodType(split, sa), types.MethodType(merge, sa)
        sa.freqs_cis = pre(D, L, theta=theta, dtype=np.float32)
        exec(setup_code, dict(ns, self=sa, config=cfg))
        glob = attn.__globals__
        glob.update(ns)
        def shard_map(fn, mesh=None, in_specs=None, out_specs=None, check_rep=None):
            specs.append((getattr(fn, "func", fn).__name__, in_specs, out_specs))      # the partitioning contract of the call site
            return fn
        glob.update(apply_rotary_emb=rot, with_sharding_constraint=lambda t, spec: t, PS=lambda *a: a, partial=functools.partial,
                    shard_map=shard_map,
                    ringattention=ringattention, ringattention_inference=ringattention_inference,
                    LLaMAConfig=types.SimpleNamespace(get_jax_mesh=lambda mesh_dim: None))
        glob["jax"].checkpoint_policies = types.SimpleNamespace(nothing_saveable=None)

        def rms(weight):
            o = types.SimpleNamespace(eps=1e-6, dtype=np.float32, param_dtype=np.float32, weight=weight)
            o._norm = types.MethodType(norm, o)
            return lambda t: norm_call(o, t)
        sm = types.SimpleNamespace(w1=lambda t: t @ W["w1"], w2=lambda t: t @ W["w2"], w3=lambda t: t @ W["w3"],
                                   dropout=lambda t, deterministic=True: t)
        mlp.__globals__["nn"] = types.SimpleNamespace(silu=lambda t: t * (1.0 / (1.0 + np.exp(-t))).astype(t.dtype))
        block.__globals__.update(blockwise_feedforward=blockwise_feedforward, with_sharding_constraint=lambda t, spec: t, PS=lambda *a: a)
        sb = types.SimpleNamespace(config=cfg, attention=lambda *a: attn(sa, *a), attention_norm=rms(W["attention_norm"]),
                                   ffn_norm=rms(W["ffn_norm"]), feed_forward=lambda t, deterministic=True: mlp(sm, t, deterministic))
        return sb, rms

    for tag, theta, qc, kc, scan_mlp, mlp_chunk, use_seg in (("dense", 1e4, 1024, 1024, False, 8, True),      # S <= chunk: dense branch, plain FFN
                                                              ("blockwise", 1e7, 8, 8, True, 8, True),        # S > chunk: ringattention, blockwise FFN
                                                              ("blockwise_noseg", 1e4, 12, 8, True, 64, False)):   # scan_mlp but S < its chunk: plain FFN
        sb, rms = build(theta, qc, kc, scan_mlp, mlp_chunk)
        record.clear()
        y = block(sb, x, am, seg if use_seg else None, pos)
        assert y.shape == x.shape and y.dtype == np.float32
        out.update({f"layer_{tag}_out": y, f"layer_{tag}_final": rms(np.ones(d, np.float32))(y),    # ... and ln_f with a unit weight (lwm/llama.py:1034)
                    f"layer_{tag}_calls": np.array([r[0] for r in record]), f"layer_{tag}_theta": np.float64(theta),
                    f"layer_{tag}_chunks": np.array([qc, kc, int(scan_mlp), mlp_chunk, int(use_seg)], np.int32)})
    import json
    out["layer_specs_json"] = np.array(json.dumps(sorted({json.dumps(sp) for sp in specs})))      # (lwm/llama.py:557-566, :601-609)
    out.update({f"layer_W_{n}": w for n, w in W.items()})
    out.update({"layer_x": x, "layer_am": am, "layer_seg": seg, "layer_pos": pos, "layer_dims": np.array([B, S, H, D, F, L], np.int32)})


class MiniFlax:
    """The part of flax.linen the classes of lwm/vqgan.py use, over a parameter tree handed in."""

    def __init__(self, conv, groupnorm, silu):
        mf = self
        self.stack, self.read = [], set()          # scopes: [tree, path, per-class counters]; leaves that were read
        self.vars = {}                             # (module path, collection, name) -> holder with .value ('cache' collection)

        class Module:
            _fields = ()

            def __init_subclass__(cls):
                ann = {}
                for c in reversed(cls.__mro__):
                    ann.update({k: v for k, v in getattr(c, "__annotations__", {}).items() if not k.startswith("_")})
                cls._fields = tuple(ann)
                for meth in ("__call__", "encode", "decode"):
                    if meth in cls.__dict__:
                        setattr(cls, meth, mf._scoped(cls.__dict__[meth]))

            def __init__(self, *args, name=None, **kw):
                vals = dict(zip(self._fields, args), **kw)
                for f in self._fields:
                    object.__setattr__(self, f, vals[f] if f in vals else getattr(type(self), f))
                object.__setattr__(self, "_name", name)           # name=...: explicit (lwm/llama.py:953: name=str(i))
                object.__setattr__(self, "_set_up", False)
                if name is None and mf.stack and mf.stack[-1][3] == "compact":     # made inside a compact __call__: ClassName_<n>
                    cnt = mf.stack[-1][2]
                    n = cnt.get(type(self).__name__, 0)
                    cnt[type(self).__name__] = n + 1
                    object.__setattr__(self, "_name", f"{type(self).__name__}_{n}")

            def __setattr__(self, k, v):
                if isinstance(v, Module) and v._name is None:     # made in setup(): named by the attribute
                    object.__setattr__(v, "_name", k)
                object.__setattr__(self, k, v)

            def param(self, name, init, *args):
                tree, path = mf.stack[-1][0], mf.stack[-1][1]
                mf.read.add(path + (name,))
                return tree[name]

            def has_variable(self, collection, name):
                return (mf.stack[-1][1], collection, name) in mf.vars

            def variable(self, collection, name, init, *args):
                key = (mf.stack[-1][1], collection, name)
                if key not in mf.vars:
                    mf.vars[key] = types.SimpleNamespace(value=init(*args))
                return mf.vars[key]

            @property
            def variables(self):
                path, res = mf.stack[-1][1], {}
                for (p, col, name), var in mf.vars.items():
                    if p == path:
                        res.setdefault(col, {})[name] = var.value
                return res

            def is_mutable_collection(self, collection):
                return False

        class Conv(Module):
            features: int
            kernel_size: object
            strides: object = None
            padding: object = "SAME"

            def __call__(self, x):
                w, b = self.param("kernel", None), self.param("bias", None)
                assert w.shape[:2] == tuple(self.kernel_size) and w.shape[3] == self.features and w.shape[2] == x.shape[-1]
                if self.padding == "VALID":
                    st = self.strides[0]
                    return conv(x, w, b, stride=st, pad=0, out_hw=((x.shape[1] - w.shape[0]) // st + 1, (x.shape[2] - w.shape[1]) // st + 1))
                assert self.padding == "SAME" and self.strides is None
                return conv(x, w, b)

        class GroupNorm(Module):                       # flax defaults: num_groups 32, epsilon 1e-6, scale and bias
            def __call__(self, x):
                return groupnorm(x, self.param("scale", None), self.param("bias", None), groups=32, eps=1e-6, silu=False)

        class Dense(Module):                           # x @ kernel (every Dense of lwm/llama.py has use_bias=False)
            features: int
            use_bias: bool = True
            dtype: object = None
            param_dtype: object = None
            kernel_init: object = None
            precision: object = None

            def __call__(self, x):
                assert self.use_bias is False
                w = self.param("kernel", None)
                assert w.shape == (x.shape[-1], self.features)
                return x @ w

        class Embed(Module):
            num_embeddings: int
            features: int
            embedding_init: object = None
            dtype: object = None
            param_dtype: object = None

            def __call__(self, ids):
                e = self.param("embedding", None)
                assert e.shape == (self.num_embeddings, self.features)
                return e[ids]

        class Dropout(Module):                         # (redefined for lwm/llama.py: Dropout(rate=...)(x, deterministic=...))
            rate: float
            deterministic: bool = None

            def __call__(self, x, deterministic=None):
                assert deterministic is True or self.deterministic is True
                return x

        def scan(target, variable_axes=None, split_rngs=None, in_axes=None, length=None, metadata_params=None):
            """flax.linen.scan as lwm/llama.py:927-941 uses it: `length` applications of `target` whose parameters are ONE
            stacked leaf per parameter, layer i = index i of axis variable_axes['params']; the module returns (carry, out)."""
            axis = variable_axes["params"]

            def cut_layer(tree, i):
                return {k: cut_layer(v, i) if isinstance(v, dict) else np.take(v, i, axis=axis) for k, v in tree.items()}

            class Scanned(Module):
                def __call__(self_, carry, *bcast):
                    tree, path = mf.stack[-1][0], mf.stack[-1][1]
                    for i in range(length):
                        mf.stack.append((cut_layer(tree, i), path, {}, "scan"))
                        try:
                            carry, _ = target(*self_._args, **self_._kw)(carry, *bcast)
                        finally:
                            mf.stack.pop()
                    return carry, None

            def make(*args, name=None, **kw):
                m = Scanned(name=name)
                object.__setattr__(m, "_args", args)
                object.__setattr__(m, "_kw", kw)
                return m
            return make

        self.Module = Module
        self.nn = types.SimpleNamespace(Module=Module, compact=lambda f: f, Conv=Conv, GroupNorm=GroupNorm, Dropout=Dropout, silu=silu,
                                        Dense=Dense, Embed=Embed, scan=scan, broadcast=object(), PARTITION_NAME="partition_name",
                                        initializers=types.SimpleNamespace(ones=None))

    def _scoped(self, fn):
        mf = self

        def call(obj, *a, **kw):
            tree, path = mf.stack[-1][0], mf.stack[-1][1]
            if obj._name is not None:
                tree, path = tree.get(obj._name, {}), path + (obj._name,)     # (a module without parameters has no subtree)
            if not obj._set_up:
                object.__setattr__(obj, "_set_up", True)
                if hasattr(obj, "setup"):                                     # setup() runs in the module's own scope
                    mf.stack.append((tree, path, {}, "setup"))
                    obj.setup()
                    mf.stack.pop()
            mf.stack.append((tree, path, {}, "compact"))
            try:
                return fn(obj, *a, **kw)
            finally:
                mf.stack.pop()
        return call

    def run(self, params, thunk):
        self.stack.append((params, (), {}, "root"))
        try:
            return thunk()
        finally:
            self.stack.pop()


def leaves(tree, path=()):
    for k, v in tree.items():
        if isinstance(v, dict):
            yield from leaves(v, path + (k,))
        else:
            yield path + (k,)


def network(out):
    import sys
    from typing import Optional
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import vqgan_ref as V
    from lwm_amd.vqgan import VQGANConfig, random_params
    src = open(f"{REF}/vqgan.py").read()
    body = ast.parse(src).body
    first = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == "VQGANModel")
    classes = [n for n in body if isinstance(n, ast.ClassDef) and n.lineno >= first.lineno]
    assert [c.name for c in classes] == ["VQGANModel", "Encoder", "Decoder", "VectorQuantizer", "DownsamplingBlock", "ResnetBlock",
                                         "AttnBlock", "Downsample", "Upsample", "UpsamplingBlock", "MidBlock"]
    out["network_lines"] = np.array([[classes[0].lineno, classes[-1].end_lineno]], np.int32)
    mf = MiniFlax(V.conv2d, V.groupnorm, V.silu)
    ns = shims()
    ns["jnp"].pad, ns["jnp"].clip = np.pad, np.clip
    ns["jax"].image = types.SimpleNamespace(resize=None)

    def resize(x, shape, method=None):
        assert method == "nearest" and shape[0] == x.shape[0] and shape[3] == x.shape[3]
        fy, fx = shape[1] // x.shape[1], shape[2] // x.shape[2]
        assert (fy * x.shape[1], fx * x.shape[2]) == tuple(shape[1:3])
        return np.repeat(np.repeat(x, fy, axis=1), fx, axis=2)
    ns["jax"].image.resize = resize
    ns.update(nn=mf.nn, Optional=Optional, VQGANConfig=object)
    code = compile(ast.Module(body=classes, type_ignores=[]), f"{REF}/vqgan.py:{classes[0].lineno}", "exec")
    exec(code, ns)                                                  # the reference's class definitions, as they are

    # two sizes: 32 x 32 with three levels, and 8 x 8 with two -- small enough for the kernel SOURCES to run it on the host
    # (tests/test_golden.py::test_product_tokeniser_with_emulated_kernels_reproduces_the_reference_run)
    for tag, updates, seed in (("network", dict(resolution=32, channel_mult=(1, 2, 4), num_embeddings=1024), 105),
                               ("network8", dict(resolution=8, channel_mult=(1, 2), num_embeddings=256), 108)):
        _network_case(out, ns, mf, V, VQGANConfig, random_params, tag, updates, seed)


def _network_case(out, ns, mf, V, VQGANConfig, random_params, tag, updates, seed):
    cfgo = VQGANConfig.get_default_config(updates)
    cfg = cfgo.as_dict()
    params = random_params(cfgo, seed=seed)
    conf = types.SimpleNamespace(**cfg)
    conf.num_resolutions = len(cfg["channel_mult"])                # VQGANConfig.get_default_config, lwm/vqgan.py:97
    conf.dropout = 0.0
    g = np.random.default_rng(seed)
    res = cfg["resolution"]
    px = g.uniform(-1, 1, (2, res, res, 3)).astype(np.float32)
    # codes placed ON the encoder's outputs (plus far-away ones): every summation order finds the same index
    z = V._conv(params["quant_conv"], V.encoder(params["encoder"], px, cfg))
    zf = z.reshape(-1, z.shape[-1])
    cb = (zf.mean(0) + 50.0 * zf.std() * g.standard_normal(params["quantize"]["embeddings"].shape)).astype(np.float32)
    slots = g.permutation(cb.shape[0])[:zf.shape[0]]
    cb[slots] = zf
    params["quantize"]["embeddings"] = cb
    model = ns["VQGANModel"](conf)
    mf.read.clear()
    zq, idx = mf.run(params, lambda: model.encode(px))
    assert np.array_equal(idx.reshape(-1), slots)
    rec = mf.run(params, lambda: model.decode(idx))
    want = set(leaves(params))
    assert mf.read == want, (sorted(want - mf.read)[:5], sorted(mf.read - want)[:5])      # every leaf read, none missing
    out.update({f"{tag}_px": px, f"{tag}_idx": idx.astype(np.int32), f"{tag}_zq": np.asarray(zq, np.float32), f"{tag}_rec": rec,
                f"{tag}_seed": np.int32(seed), f"{tag}_codebook": cb, f"{tag}_leaves": np.int32(len(want))})


def model(out):
    """BASELINE configs[0] in miniature: the reference's own model classes produce logits on the CPU."""
    import functools
    import sys
    from typing import Any, Dict, List, Optional, Union
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import attention_ref as A
    path = f"{REF}/llama.py"
    src = open(path).read()
    want = ["RMSNorm", "precompute_freqs_cis", "apply_rotary_emb", "FlaxLLaMAAttention", "FlaxLLaMAMLP", "FlaxLLaMABlock",
            "FlaxLLaMABlockCollection", "FlaxLLaMAModule", "FlaxLLaMAForCausalLMModule"]
    nodes = [n for n in ast.parse(src).body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in want]
    assert [n.name for n in nodes] == want
    out["model_lines"] = np.array([[n.lineno, n.end_lineno] for n in nodes], np.int32)
    mf = MiniFlax(None, None, lambda t: t * (1.0 / (1.0 + np.exp(-t))).astype(t.dtype))
    ns = shims()
    jnp = ns["jnp"]
    jnp.take, jnp.where, jnp.zeros, jnp.array, jnp.int32, jnp.ones_like = np.take, np.where, np.zeros, np.array, np.int32, np.ones_like
    ns["jax"].checkpoint_policies = types.SimpleNamespace(nothing_saveable=None)
    ns["jax"].lax.Precision = object
    ns["jax"].nn.initializers = types.SimpleNamespace(normal=lambda stddev=None: None)
    record = []

    def ringattention(q, k, v, attn_bias, segment_ids, axis_name=None, float32_logits=None, cache_idx=None, blockwise_kwargs=None):
        kw = blockwise_kwargs
        assert axis_name == "sp" and float32_logits is True and cache_idx is None and kw["causal_block_size"] == 1
        record.append("ringattention")
        return A.blockwise_ring_attention(q, k, v, ring=1, q_chunk=kw["query_chunk_size"], k_chunk=kw["key_chunk_size"], causal=True,
                                          segment_ids=segment_ids, key_valid=(attn_bias[:, 0, 0] == 0).astype(np.uint8))

    def ringattention_inference(q, k, v, attn_mask, axis_name=None):
        record.append("ringattention_inference")
        return A.ring_inference(q, k, v, attn_mask[:, 0])

    def blockwise_feedforward(module, xx, chunk_size, pre_remat=None):
        record.append("blockwise_feedforward")
        return module(xx)

    class Output:                                      # transformers' ModelOutput: fields by name, non-None fields by position
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def __getitem__(self, i):
            return [v for v in self.__dict__.values() if v is not None][i]
    ns.update(nn=mf.nn, nn_partitioning=types.SimpleNamespace(ScanIn=lambda axis: axis), remat=lambda cls, **kw: cls,
              Optional=Optional, Union=Union, Any=Any, Dict=Dict, List=List, partial=functools.partial,
              LLaMAConfig=types.SimpleNamespace(get_jax_mesh=lambda mesh_dim: None), PS=lambda *a: a,
              with_sharding_constraint=lambda t, spec: t, shard_map=lambda fn, mesh=None, in_specs=None, out_specs=None, check_rep=None: fn,
              ringattention=ringattention, ringattention_inference=ringattention_inference, blockwise_feedforward=blockwise_feedforward,
              FlaxBaseModelOutput=Output, FlaxCausalLMOutput=Output)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), f"{path}:{nodes[0].lineno}", "exec"), ns)      # the reference's definitions

    g = np.random.default_rng(982)
    Vc, d, H, F, NL, L = 37, 32, 2, 48, 2, 32
    B, S = 2, 24
    std = 0.2
    per = {"attention/wq": (d, d), "attention/wk": (d, d), "attention/wv": (d, d), "attention/wo": (d, d),
           "feed_forward/w1": (d, F), "feed_forward/w2": (F, d), "feed_forward/w3": (d, F)}
    layers = [{**{k + "/kernel": (g.standard_normal(shp) * std).astype(np.float32) for k, shp in per.items()},
               "attention_norm/kernel": (1 + 0.1 * g.standard_normal(d)).astype(np.float32),
               "ffn_norm/kernel": (1 + 0.1 * g.standard_normal(d)).astype(np.float32)} for _ in range(NL)]
    top = {"transformer/wte/embedding": g.standard_normal((Vc, d)).astype(np.float32),
           "transformer/ln_f/kernel": (1 + 0.1 * g.standard_normal(d)).astype(np.float32),
           "lm_head/kernel": (g.standard_normal((d, Vc)) * std).astype(np.float32)}
    # the two on-disk layouts of a FlaxLLaMAForCausalLM train state, by the names lwm_amd/weights.py::flax_llama_to_lwm maps
    flat_layers = dict(top, **{f"transformer/h/{i}/{k}": v for i, lay in enumerate(layers) for k, v in lay.items()})
    flat_scan = dict(top, **{f"transformer/h/scan_decoder/{k}": np.stack([lay[k] for lay in layers], axis=0) for k in layers[0]})

    def nest(flat):
        tree = {}
        for k, v in flat.items():
            t = tree
            parts = k.split("/")
            for p in parts[:-1]:
                t = t.setdefault(p, {})
            t[parts[-1]] = v
        return tree
    tokens = g.integers(0, Vc, (B, S)).astype(np.int32)
    am = np.ones((B, S), np.int32)
    am[0, :3] = 0
    seg = np.zeros((B, S), np.int32)
    seg[1, 11:] = 1
    pos = np.tile(np.arange(S, dtype=np.int32), (B, 1))
    results = {}
    for tag, flat, scan_layers, chunk, use_seg in (("layers", flat_layers, False, 1024, True), ("scan", flat_scan, True, 1024, True),
                                                    ("scan_blockwise", flat_scan, True, 8, False)):
        cfg = types.SimpleNamespace(vocab_size=Vc, hidden_size=d, intermediate_size=F, num_hidden_layers=NL, num_attention_heads=H,
                                    max_sequence_length=L, rms_norm_eps=1e-6, initializer_range=0.02, resid_pdrop=0.0, embd_pdrop=0.0,
                                    attn_pdrop=0.0, tie_word_embeddings=False, scan_attention=True, scan_mlp=True,
                                    scan_query_chunk_size=chunk, scan_key_chunk_size=chunk, scan_mlp_chunk_size=chunk,
                                    scan_layers=scan_layers, param_scan_axis=0, mesh_dim="1,1,1,1", theta=10000)
        tree = nest(flat)
        mf.read.clear()
        record.clear()
        m = ns["FlaxLLaMAForCausalLMModule"](cfg, dtype=np.float32)
        res = mf.run(tree, lambda: m(tokens, am, seg if use_seg else None, pos))
        logits = res.logits
        assert logits.shape == (B, S, Vc) and logits.dtype == np.float32
        read = {"/".join(p) for p in mf.read}
        assert read == set(flat), (sorted(set(flat) - read)[:4], sorted(read - set(flat))[:4])       # every leaf read, none missing
        results[tag] = logits
        out.update({f"model_{tag}_logits": logits, f"model_{tag}_calls": np.array(sorted(set(record)))})
    assert np.array_equal(results["layers"], results["scan"])         # the stacked layout is the same model
    out.update({f"model_flat_layers::{k}": v for k, v in flat_layers.items()})
    out.update({f"model_flat_scan::{k}": v for k, v in flat_scan.items() if "scan_decoder" in k})
    out.update({"model_tokens": tokens, "model_am": am, "model_seg": seg, "model_dims": np.array([Vc, d, H, F, NL, L, B, S], np.int32)})

    # ---- the reference's own 'debug' size (LLAMA_STANDARD_CONFIGS['debug']: hidden 256, 2 heads -> head_dim 128, the one the HIP
    # kernels are written for), bf16-representable weights: what the product path, kernel sources included, is run against on the
    # host (tests/test_golden.py::test_product_path_with_emulated_kernels_reproduces_the_reference_model)
    gd = np.random.default_rng(2560)
    dV, dd, dH, dF, dNL, dL, dB, dS = 64, 256, 2, 256, 2, 128, 2, 64

    def bf(a):
        b = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
        return ((b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    dper = {"attention/wq": (dd, dd), "attention/wk": (dd, dd), "attention/wv": (dd, dd), "attention/wo": (dd, dd),
            "feed_forward/w1": (dd, dF), "feed_forward/w2": (dF, dd), "feed_forward/w3": (dd, dF)}
    dlayers = [{**{k + "/kernel": bf(gd.standard_normal(shp) * 0.06) for k, shp in dper.items()},
                "attention_norm/kernel": bf(1 + 0.1 * gd.standard_normal(dd)), "ffn_norm/kernel": bf(1 + 0.1 * gd.standard_normal(dd))}
               for _ in range(dNL)]
    dflat = {"transformer/wte/embedding": bf(gd.standard_normal((dV, dd))), "transformer/ln_f/kernel": bf(1 + 0.1 * gd.standard_normal(dd)),
             "lm_head/kernel": bf(gd.standard_normal((dd, dV)) * 0.06)}
    dflat.update({f"transformer/h/scan_decoder/{k}": np.stack([lay[k] for lay in dlayers], axis=0) for k in dlayers[0]})
    dtokens = gd.integers(0, dV, (dB, dS)).astype(np.int32)
    dam = np.ones((dB, dS), np.int32)
    dam[0, :5] = 0
    dseg = np.zeros((dB, dS), np.int32)
    dseg[1, 23:] = 1
    dcfg = types.SimpleNamespace(vocab_size=dV, hidden_size=dd, intermediate_size=dF, num_hidden_layers=dNL, num_attention_heads=dH,
                                 max_sequence_length=dL, rms_norm_eps=1e-6, initializer_range=0.02, resid_pdrop=0.0, embd_pdrop=0.0,
                                 attn_pdrop=0.0, tie_word_embeddings=False, scan_attention=True, scan_mlp=True,
                                 scan_query_chunk_size=1024, scan_key_chunk_size=1024, scan_mlp_chunk_size=1024,
                                 scan_layers=True, param_scan_axis=0, mesh_dim="1,1,1,1", theta=10000)
    mf.read.clear()
    md = ns["FlaxLLaMAForCausalLMModule"](dcfg, dtype=np.float32)
    dres = mf.run(nest(dflat), lambda: md(dtokens, dam, dseg, np.tile(np.arange(dS, dtype=np.int32), (dB, 1))))
    assert dres.logits.shape == (dB, dS, dV) and {"/".join(p_) for p_ in mf.read} == set(dflat)
    out.update({f"model_debug_flat_bf16::{k}": (v.view(np.uint32) >> 16).astype(np.uint16) for k, v in dflat.items()})    # (bf16 bits)
    out.update({"model_debug_logits": dres.logits, "model_debug_tokens": dtokens, "model_debug_am": dam, "model_debug_seg": dseg,
                "model_debug_dims": np.array([dV, dd, dH, dF, dNL, dL, dB, dS], np.int32)})

    # ---- the vision-language model of BASELINE configs[3] (lwm/vision_llama.py:255-443): the same transformer with a second
    # embedding table and a second head
    vpath = f"{REF}/vision_llama.py"
    vsrc = open(vpath).read()
    vwant = ["FlaxVideoLLaMAModule", "FlaxVideoLLaMAForCausalLMModule"]
    vnodes = [n for n in ast.parse(vsrc).body if isinstance(n, ast.ClassDef) and n.name in vwant]
    assert [n.name for n in vnodes] == vwant
    out["vmodel_lines"] = np.array([[n.lineno, n.end_lineno] for n in vnodes], np.int32)
    ns.update(VideoLLaMAConfig=object)
    jnp.zeros_like, jnp.cumsum = np.zeros_like, np.cumsum
    jnp.clip = lambda a, a_min=None, a_max=None: np.clip(a, a_min, a_max)
    exec(compile(ast.Module(body=vnodes, type_ignores=[]), f"{vpath}:{vnodes[0].lineno}", "exec"), ns)
    VV = 19
    vflat = dict(flat_scan, **{"transformer/vte/embedding": g.standard_normal((VV, d)).astype(np.float32),
                               "vision_head/kernel": (g.standard_normal((d, VV)) * std).astype(np.float32)})
    vm = g.random((B, S)) < 0.5
    vtokens = np.where(vm, g.integers(0, VV, (B, S)), tokens).astype(np.int32)
    cfg = types.SimpleNamespace(vocab_size=Vc, vision_vocab_size=VV, hidden_size=d, intermediate_size=F, num_hidden_layers=NL,
                                num_attention_heads=H, max_sequence_length=L, rms_norm_eps=1e-6, initializer_range=0.02, resid_pdrop=0.0,
                                embd_pdrop=0.0, attn_pdrop=0.0, tie_word_embeddings=False, tie_vision_embeddings=False, sample_mode="all",
                                scan_attention=True, scan_mlp=True, scan_query_chunk_size=1024, scan_key_chunk_size=1024,
                                scan_mlp_chunk_size=1024, scan_layers=True, param_scan_axis=0, mesh_dim="1,1,1,1", theta=10000)
    mf.read.clear()
    m = ns["FlaxVideoLLaMAForCausalLMModule"](cfg, dtype=np.float32)
    # attention_mask / segment_ids / position_ids left to the module's own defaults (lwm/vision_llama.py:385-394), as lwm/train.py:186-191 calls it
    res = mf.run(nest(vflat), lambda: m(vtokens, vm))
    vlogits, tlogits = res.logits
    assert vlogits.shape == (B, S, VV) and tlogits.shape == (B, S, Vc)
    read = {"/".join(p) for p in mf.read}
    assert read == set(vflat), (sorted(set(vflat) - read)[:4], sorted(read - set(vflat))[:4])
    out.update({"vmodel_tokens": vtokens, "vmodel_vm": vm, "vmodel_vision_logits": vlogits, "vmodel_text_logits": tlogits,
                "vmodel_vte": vflat["transformer/vte/embedding"], "vmodel_vision_head": vflat["vision_head/kernel"]})


def cached_model(out):
    """f.1: the reference's model classes in CACHED INFERENCE -- init_cache (lwm/llama.py:806-824: one pass over max_length rows with
    init_cache=True creates the cache variables), a prefill of the prompt, then one token at a time; flax's variable collections are
    MiniFlax's (`self.variable / has_variable / variables`), the one-token cache write runs through the shard_map emulation of
    cache_decode() with one device.  Every step's last-position logits are recorded; the test compares them with the oracle model's
    FULL forward over the sequence so far -- the cache machinery (mask shift, cache write, positions) computes the same function."""
    import functools
    from typing import Any, Dict, List, Optional, Union
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import attention_ref as A
    path = f"{REF}/llama.py"
    src = open(path).read()
    want = ["RMSNorm", "precompute_freqs_cis", "apply_rotary_emb", "FlaxLLaMAAttention", "FlaxLLaMAMLP", "FlaxLLaMABlock",
            "FlaxLLaMABlockCollection", "FlaxLLaMAModule", "FlaxLLaMAForCausalLMModule"]
    nodes = [n for n in ast.parse(src).body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in want]
    mf = MiniFlax(None, None, lambda t: t * (1.0 / (1.0 + np.exp(-t))).astype(t.dtype))
    ns = shims()
    jnp = ns["jnp"]

    class At(np.ndarray):
        @property
        def at(self):
            arr = self

            class Ix:
                def __getitem__(self, idx):
                    return types.SimpleNamespace(set=lambda v: (lambda r: (r.__setitem__(idx, v), r)[1])(arr.copy()))
            return Ix()

    def dus(operand, update, start):
        res = operand.copy()
        st = [int(np.clip(int(s0), 0, d - u)) for s0, d, u in zip(start, operand.shape, update.shape)]
        res[tuple(slice(a, a + u) for a, u in zip(st, update.shape))] = update
        return res
    jnp.take, jnp.where, jnp.array, jnp.int32, jnp.ones_like, jnp.logical_and = np.take, np.where, np.array, np.int32, np.ones_like, np.logical_and
    jnp.zeros = lambda shape, dtype=None: np.zeros(shape, dtype).view(At)
    ns["lax"].dynamic_update_slice = dus
    ns["jax"].lax.axis_index = lambda axis: 0
    ns["jax"].lax.cond = lambda p, t, f: t() if bool(p) else f()
    ns["jax"].checkpoint_policies = types.SimpleNamespace(nothing_saveable=None)
    ns["jax"].lax.Precision = object
    ns["jax"].nn.initializers = types.SimpleNamespace(normal=lambda stddev=None: None)
    calls = []

    def ringattention_inference(q, k, v, attn_mask, axis_name=None):
        calls.append((q.shape[1], k.shape[1]))
        return A.ring_inference(q, k, v, attn_mask[:, 0])

    def no_training_op(*a, **kw):
        raise AssertionError("the blockwise branch must not run in cached inference with S <= chunk")

    class Output:
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def __getitem__(self, i):
            return [v for v in self.__dict__.values() if v is not None][i]
    ns.update(nn=mf.nn, nn_partitioning=types.SimpleNamespace(ScanIn=lambda axis: axis), remat=lambda cls, **kw: cls,
              Optional=Optional, Union=Union, Any=Any, Dict=Dict, List=List, partial=functools.partial,
              LLaMAConfig=types.SimpleNamespace(get_jax_mesh=lambda mesh_dim: types.SimpleNamespace(shape={"sp": 1})), PS=lambda *a: a,
              with_sharding_constraint=lambda t, spec: t, shard_map=lambda fn, mesh=None, in_specs=None, out_specs=None, check_rep=None: fn,
              ringattention=no_training_op, ringattention_inference=ringattention_inference, blockwise_feedforward=no_training_op,
              FlaxBaseModelOutput=Output, FlaxCausalLMOutput=Output)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), f"{path}:{nodes[0].lineno}", "exec"), ns)

    g = np.random.default_rng(806)
    Vc, d, H, F, NL, L = 37, 32, 2, 48, 2, 20
    B, S, NEW = 2, 9, 5
    std = 0.2
    flat = {"transformer/wte/embedding": g.standard_normal((Vc, d)).astype(np.float32),
            "transformer/ln_f/kernel": (1 + 0.1 * g.standard_normal(d)).astype(np.float32),
            "lm_head/kernel": (g.standard_normal((d, Vc)) * std).astype(np.float32)}
    for i in range(NL):
        for k, shp in (("attention/wq", (d, d)), ("attention/wk", (d, d)), ("attention/wv", (d, d)), ("attention/wo", (d, d)),
                       ("feed_forward/w1", (d, F)), ("feed_forward/w2", (F, d)), ("feed_forward/w3", (d, F))):
            flat[f"transformer/h/{i}/{k}/kernel"] = (g.standard_normal(shp) * std).astype(np.float32)
        for k in ("attention_norm", "ffn_norm"):
            flat[f"transformer/h/{i}/{k}/kernel"] = (1 + 0.1 * g.standard_normal(d)).astype(np.float32)
    tree = {}
    for k, v in flat.items():
        t = tree
        parts = k.split("/")
        for p_ in parts[:-1]:
            t = t.setdefault(p_, {})
        t[parts[-1]] = v
    cfg = types.SimpleNamespace(vocab_size=Vc, hidden_size=d, intermediate_size=F, num_hidden_layers=NL, num_attention_heads=H,
                                max_sequence_length=L, rms_norm_eps=1e-6, initializer_range=0.02, resid_pdrop=0.0, embd_pdrop=0.0,
                                attn_pdrop=0.0, tie_word_embeddings=False, scan_attention=True, scan_mlp=False,
                                scan_query_chunk_size=1024, scan_key_chunk_size=1024, scan_mlp_chunk_size=1024,
                                scan_layers=False, param_scan_axis=0, mesh_dim="1,1,1,1", theta=10000)
    m = ns["FlaxLLaMAForCausalLMModule"](cfg, dtype=np.float32)
    ext = np.ones((B, L), np.int32)                                   # the key mask over max_length (prepare_inputs_for_generation)
    # init_cache (lwm/llama.py:806-824)
    mf.run(tree, lambda: m(np.ones((B, L), np.int32), np.ones((B, L), np.int32), None, np.tile(np.arange(L, dtype=np.int32), (B, 1)),
                           init_cache=True))
    cache_vars = sorted("/".join(p) + ":" + n for (p, col, n) in mf.vars if col == "cache")
    assert len(cache_vars) == 3 * NL and all(v.value.shape == (B, L, H, d // H) for (p, c_, n), v in mf.vars.items() if n != "cache_index")
    prompt = g.integers(0, Vc, (B, S)).astype(np.int32)
    step_logits, tokens = [], prompt
    pos = np.tile(np.arange(S, dtype=np.int32), (B, 1))
    res = mf.run(tree, lambda: m(prompt, ext, None, pos))              # prefill
    for t in range(NEW):
        logits = res.logits[:, -1]
        step_logits.append(logits)
        nxt = logits.argmax(-1).astype(np.int32)[:, None]            # greedy
        tokens = np.concatenate([tokens, nxt], axis=1)
        pos = pos[:, -1:] + 1                                        # update_inputs_for_generation
        if t + 1 < NEW:
            res = mf.run(tree, lambda: m(nxt, ext, None, pos))       # one token through the cache
    idx = {int(v.value) for (p, col, n), v in mf.vars.items() if n == "cache_index"}
    assert idx == {S + NEW - 1}, idx
    assert calls[-1] == (1, L) and (S, L) in calls                     # attention always runs over the WHOLE cache
    out.update({"cached_step_logits": np.stack(step_logits, 1), "cached_tokens": tokens, "cached_dims": np.array([Vc, d, H, F, NL, L, B, S, NEW], np.int32),
                "cached_cache_vars": np.array(cache_vars)})
    out.update({f"cached_flat::{k}": v for k, v in flat.items()})


def chat_prompt(out):
    import io
    import math
    from PIL import Image
    path = f"{REF}/vision_chat.py"
    frame, a0, a1 = cut(path, "Sampler", "_process_frame")
    read, b0, b1 = cut(path, "Sampler", "_read_process_vision")
    cons, c0, c1 = cut(path, "Sampler", "construct_input")
    out["chat_lines"] = np.array([[a0, a1], [b0, b1], [c0, c1]], np.int32)
    g = np.random.default_rng(57)
    # (1) frames: a wide and a tall picture
    for tag, (w, h) in (("wide", (97, 64)), ("tall", (50, 83))):
        img = g.integers(0, 256, (h, w, 3)).astype(np.uint8)
        y = frame(None, Image.fromarray(img), 32)
        assert y.shape == (32, 32, 3) and y.dtype == np.float32
        out.update({f"chat_frame_{tag}_in": img, f"chat_frame_{tag}_out": y})
    # (2) one picture -> tokens; (3) the whole prompt, two prompts of different length in one batch
    png = io.BytesIO()
    Image.fromarray(g.integers(0, 256, (300, 280, 3)).astype(np.uint8)).save(png, format="PNG")
    out["chat_png"] = np.frombuffer(png.getvalue(), np.uint8)
    codes = g.integers(0, 8192, (1, 16, 16)).astype(np.int64)

    class Tok:                                            # a tokenizer stub: one id per character, offset so that ids are not codes
        def encode(self, text):
            return [9000 + ord(c) for c in text]
    seen = []

    def encode(v):
        seen.append(np.array(v))
        return None, codes[:len(v)]
    self = types.SimpleNamespace(tokenizer=Tok(), vqgan=types.SimpleNamespace(encode=encode), n_tokens_per_frame=257, min_buffer_size=256,
                                 block_size=128)
    self._process_frame = types.MethodType(frame, self)
    self._read_process_vision = types.MethodType(read, self)
    glob = read.__globals__
    glob.update(open_file=lambda p, mode: io.BytesIO(png.getvalue()), Image=Image)
    glob["jax"].device_get = lambda x: x
    cons.__globals__.update(math=math, tqdm=lambda x: x)
    toks = read(self, "picture.png", 4)
    assert len(toks) == 257 and toks[-1] == 8193 and toks[:256] == codes[0].reshape(-1).tolist()
    assert seen[0].shape == (1, 256, 256, 3) and abs(float(seen[0].max())) <= 1.0
    out.update({"chat_codes": codes, "chat_tokens": np.array(toks, np.int64), "chat_pixels": seen[0]})
    batch = cons(self, [dict(input_path="picture.png", question="What is this?"), dict(input_path="picture.png", question="And what colour is the sky in it?")], 2)
    out.update({"chat_input_ids": batch["input_ids"].astype(np.int64), "chat_vision_masks": batch["vision_masks"],
                "chat_attention_mask": batch["attention_mask"].astype(np.int64), "chat_block_size": np.int32(128)})


def generation_inputs(out):
    """FlaxVideoLLaMAForCausalLM.prepare_inputs_for_generation / update_inputs_for_generation (lwm/vision_llama.py:447-474) and
    the text model's (lwm/llama.py:1113-1140): the extended key mask over max_length, the positions of a left-padded prompt
    (cumsum of the mask - 1) and of the following steps."""
    g = np.random.default_rng(447)
    B, S, L = 3, 9, 14
    am = np.ones((B, S), np.int32)
    am[0, :4] = 0
    am[2, :1] = 0
    ids = g.integers(0, 50, (B, S)).astype(np.int32)
    lines = []
    for tag, path, cls in (("vision", f"{REF}/vision_llama.py", "FlaxVideoLLaMAForCausalLM"), ("text", f"{REF}/llama.py", "FlaxLLaMAForCausalLM")):
        prep, a0, a1 = cut(path, cls, "prepare_inputs_for_generation")
        upd, b0, b1 = cut(path, cls, "update_inputs_for_generation")
        lines += [[a0, a1], [b0, b1]]
        prep.__globals__["jnp"].ones, prep.__globals__["jnp"].broadcast_to, prep.__globals__["jnp"].arange = np.ones, np.broadcast_to, np.arange

        def dus(operand, update, start):
            res = np.array(operand, copy=True)
            res[tuple(slice(int(s0), int(s0) + u) for s0, u in zip(start, update.shape))] = update
            return res
        prep.__globals__["lax"].dynamic_update_slice = dus
        self = types.SimpleNamespace(init_cache=lambda b, m: ("cache", b, m))
        for case, mask in (("padded", am), ("nomask", None)):
            kw = dict(vision_masks="vm") if tag == "vision" else {}
            r = prep(self, ids, L, attention_mask=mask, **kw)
            assert r["past_key_values"] == ("cache", B, L)
            nxt = upd(self, types.SimpleNamespace(past_key_values="pkv"), dict(r))
            out.update({f"gen_{tag}_{case}_mask": np.asarray(r["attention_mask"]), f"gen_{tag}_{case}_pos": np.asarray(r["position_ids"]),
                        f"gen_{tag}_{case}_next": np.asarray(nxt["position_ids"])})
    out.update({"gen_lines": np.array(lines, np.int32), "gen_ids": ids, "gen_am": am, "gen_max_length": np.int32(L)})


def flags(out):
    """The flag DEFINITIONS of the three entry points (b5): the `define_flags_with_default(...)` statement of lwm/train.py,
    lwm/vision_chat.py and lwm/vision_generation.py executed with a recorder for tux's function; the config groups
    (`X.get_default_config()`) become the marker "<group>"."""
    import json
    rec = {}
    for mod in ("train", "vision_chat", "vision_generation"):
        src = open(f"{REF}/{mod}.py").read()
        node = next(n for n in ast.parse(src).body if isinstance(n, ast.Assign) and "define_flags_with_default" in ast.get_source_segment(src, n))

        class Group:
            def __getattr__(self, name):
                return self

            def __call__(self, *a, **kw):
                return "<group>"
        ns = dict(define_flags_with_default=lambda **kw: (kw, kw))
        for name in ("DatasetFactory", "OptimizerFactory", "StreamingCheckpointer", "VideoLLaMAConfig", "JaxDistributedConfig", "tux"):
            ns[name] = Group()
        exec(compile(ast.Module(body=[node], type_ignores=[]), f"{REF}/{mod}.py:{node.lineno}", "exec"), ns)
        rec[mod] = dict(lines=[node.lineno, node.end_lineno], flags=ns["FLAGS"])
    out["flags_json"] = np.array(json.dumps(rec, sort_keys=True))
    # ... and the configuration objects those flags fill: the keyword defaults of LLaMAConfig / VideoLLaMAConfig / VQGANConfig
    # (read off the reference's __init__ signatures) and the size table LLAMA_STANDARD_CONFIGS (a literal)
    conf = {}
    for mod, cls in (("llama", "LLaMAConfig"), ("vision_llama", "VideoLLaMAConfig"), ("vqgan", "VQGANConfig")):
        src = open(f"{REF}/{mod}.py").read()
        c = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == cls)
        f = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
        names = [a.arg for a in f.args.args][1:]
        conf[cls] = dict(lines=[f.lineno, f.end_lineno],
                         defaults={n: ast.literal_eval(v) for n, v in zip(names[len(names) - len(f.args.defaults):], f.args.defaults)})
    src = open(f"{REF}/llama.py").read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "LLAMA_STANDARD_CONFIGS")
    conf["LLAMA_STANDARD_CONFIGS"] = dict(lines=[node.lineno, node.end_lineno], table=ast.literal_eval(node.value))
    out["configs_json"] = np.array(json.dumps(conf, sort_keys=True))


def main():
    out = {}
    rope(out)
    rmsnorm(out)
    vq(out)
    masks(out)
    cache(out)
    cache_decode(out)
    vision_text(out)
    video(out)
    layer(out)
    network(out)
    model(out)
    cached_model(out)
    chat_prompt(out)
    generation_inputs(out)
    flags(out)
    target = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_run.npz")
    np.savez_compressed(target, **out)
    print("wrote", os.path.basename(target) + ";", "lines", {k: out[k].tolist() for k in out if k.endswith("_lines")}, "vq margin", float(out["vq_min_margin"]))


if __name__ == "__main__":
    main()
