"""Weight I/O for the harness and the VQGAN.

Three on-disk formats reach the reference's hot path:

  * HF-PyTorch LLaMA checkpoints -- how the released LWM text models are consumed outside JAX
    (README.md:74, scripts/sample_pyt.py:8: `LlamaForCausalLM.from_pretrained`).  torch Linear
    weights are (out, in) and wq/wk follow the rotate_half RoPE convention; the reference's
    flax Dense kernels are (in, out) and its RoPE rotates interleaved pairs
    (lwm/llama.py:353-375), so q/k projections are re-ordered on load
    (`lwm_amd.llama.hf_rotary_to_interleaved`).
  * the flax-msgpack stream that `tux.StreamingCheckpointer` writes (lwm/train.py:104-107,
    lwm/vision_chat.py:182-185 `load_trainstate_checkpoint`): a concatenation of msgpack
    records `(key_tuple, flax.serialization.to_bytes(leaf))`.  [The parameter NAMES and the stacked
    `scan_decoder` layout are verified against a run of the reference's own model classes
    (tests/golden/gen_ref_run_golden.py::model, tests/test_golden.py); the byte format is upstream and unverifiable here:
    `tux` and `flax` are not in this image; the record layout below restates
    flax/serialization.py (`_ndarray_to_bytes`, ext type 1 = (shape, dtype name, raw bytes);
    `__msgpack_chunks__` for leaves > 2**30 bytes) and tux/checkpoint.py.]
  * the VQGAN parameter pickle (lwm/vqgan.py:19).  If it was written from jax arrays it names
    `jax._src.array._reconstruct_array`; `load_pickle_tree` resolves that (and flax's
    FrozenDict) to plain numpy / dict without importing jax.

Everything here is host-side plumbing: bytes -> torch tensors.  No arithmetic.
"""
from __future__ import annotations

import glob
import io
import json
import os
import pickle
import re

import numpy as np
import torch

from .llama import LLaMAConfig, hf_rotary_to_interleaved

# ------------------------------------------------------------------ HF LLaMA
_HF_LAYER = re.compile(r"^model\.layers\.(\d+)\.(.+)$")
# HF name -> (harness name, kind).  MLP: flax w1 = gate, w3 = up, w2 = down (lwm/llama.py:631-661).
_HF_BLOCK = {
    "self_attn.q_proj.weight": ("attention.wq", "rotary"),
    "self_attn.k_proj.weight": ("attention.wk", "rotary"),
    "self_attn.v_proj.weight": ("attention.wv", "linear"),
    "self_attn.o_proj.weight": ("attention.wo", "linear"),
    "mlp.gate_proj.weight": ("feed_forward.w1", "linear"),
    "mlp.down_proj.weight": ("feed_forward.w2", "linear"),
    "mlp.up_proj.weight": ("feed_forward.w3", "linear"),
    "input_layernorm.weight": ("attention_norm.kernel", "vector"),
    "post_attention_layernorm.weight": ("ffn_norm.kernel", "vector"),
}
_HF_TOP = {
    "model.embed_tokens.weight": ("wte", "vector"),
    "model.norm.weight": ("ln_f.kernel", "vector"),
    "lm_head.weight": ("lm_head", "linear"),
}


def config_from_hf(cfg: dict) -> LLaMAConfig:
    """HF `config.json` (LlamaConfig) -> LLaMAConfig with the reference's field names
    (lwm/llama.py:133-199).  Grouped-query checkpoints are refused: the reference has
    num_key_value_heads == num_attention_heads everywhere."""
    nh = cfg["num_attention_heads"]
    if cfg.get("num_key_value_heads", nh) != nh:
        raise ValueError("grouped-query attention is not part of the reference model (lwm/llama.py:390-421)")
    if cfg.get("rope_scaling"):
        raise ValueError("rope_scaling is not used by LWM checkpoints (theta is scaled instead, README.md:112-117)")
    return LLaMAConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
                       intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
                       num_attention_heads=nh, max_sequence_length=cfg.get("max_position_embeddings", 4096),
                       rms_norm_eps=cfg.get("rms_norm_eps", 1e-6), initializer_range=cfg.get("initializer_range", 0.02),
                       theta=cfg.get("rope_theta", 10000.0))


def read_hf_checkpoint(path):
    """-> (state_dict, config dict or None).  `path`: a directory with config.json and
    *.safetensors or pytorch_model*.bin shards, or a single weights file."""
    cfg = None
    if os.path.isdir(path):
        cj = os.path.join(path, "config.json")
        if os.path.exists(cj):
            with open(cj) as f:
                cfg = json.load(f)
        files = sorted(glob.glob(os.path.join(path, "*.safetensors"))) or \
            sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
        if not files:
            raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {path}")
    else:
        files = [path]
    sd = {}
    for fn in files:
        if fn.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd.update(load_file(fn))
        else:
            sd.update(torch.load(fn, map_location="cpu", weights_only=True))
    return sd, cfg


def hf_to_lwm(state_dict, num_heads):
    """HF LlamaForCausalLM state dict -> {harness parameter name: tensor} (flax layouts)."""
    out = {}
    for name, w in state_dict.items():
        if name.endswith("rotary_emb.inv_freq"):
            continue
        m = _HF_LAYER.match(name)
        if m:
            if m.group(2) not in _HF_BLOCK:
                raise KeyError(f"unexpected tensor {name!r} in an HF LLaMA checkpoint")
            tgt, kind = _HF_BLOCK[m.group(2)]
            tgt = f"h.{m.group(1)}.{tgt}"
        elif name in _HF_TOP:
            tgt, kind = _HF_TOP[name]
        else:
            raise KeyError(f"unexpected tensor {name!r} in an HF LLaMA checkpoint")
        if kind == "rotary":
            out[tgt] = hf_rotary_to_interleaved(w, num_heads)
        elif kind == "linear":
            out[tgt] = w.t().contiguous()
        else:
            out[tgt] = w
    if "lm_head" not in out and "wte" in out:          # tie_word_embeddings
        out["lm_head"] = out["wte"].t().contiguous()
    return out


def load_params(model, params, strict=True, report=None):
    """Copy {name: tensor} into a harness LLaMAForCausalLM (dtype/device of the model).  strict=False loads what
    matches; `report` (a callable taking one string) is then told which parameters kept their initial values and
    which checkpoint entries found no home -- a silent partial load looks like a trained model and is not."""
    own = dict(model.named_parameters())
    missing = [n for n in own if n not in params]
    extra = [n for n in params if n not in own]
    if strict and (missing or extra):
        raise KeyError(f"checkpoint/model mismatch: missing {missing[:4]} extra {extra[:4]}")
    if report is not None and (missing or extra):
        report(f"checkpoint/model mismatch: {len(missing)} model parameter(s) NOT in the checkpoint (left at their "
               f"initial values): {missing[:6]}{' ...' if len(missing) > 6 else ''}; {len(extra)} checkpoint entr"
               f"{'y' if len(extra) == 1 else 'ies'} unused: {extra[:6]}{' ...' if len(extra) > 6 else ''}")
    with torch.no_grad():
        for n, p in own.items():
            if n in params:
                src = params[n]
                if tuple(src.shape) != tuple(p.shape):
                    raise ValueError(f"{n}: checkpoint {tuple(src.shape)} vs model {tuple(p.shape)}")
                p.copy_(src.to(device=p.device, dtype=p.dtype))
    return model


def load_hf_llama(path, dtype=torch.bfloat16, device="cuda", **config_updates):
    """One call: HF checkpoint directory -> harness model on the device."""
    from .llama import LLaMAForCausalLM
    sd, cfg = read_hf_checkpoint(path)
    if cfg is None:
        raise FileNotFoundError(f"{path}: config.json needed to size the model")
    lc = config_from_hf(cfg)
    for k, v in config_updates.items():
        setattr(lc, k, v)
    with torch.device(device):
        model = LLaMAForCausalLM(lc, dtype)
    return load_params(model, hf_to_lwm(sd, lc.num_attention_heads))


# ------------------------------------------------------------------ flax msgpack stream
_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
# flax.serialization._chunk writes {'__msgpack_chunked_array__': True, 'shape': {'0': d0, ...},
# 'chunks': {'0': flat piece, ...}} for leaves above MAX_CHUNK_SIZE = 2**30 bytes (every stacked
# f32 kernel of a scan_layers 7B checkpoint).  The second spelling is accepted for older writers.
_CHUNK_KEYS = ("__msgpack_chunked_array__", "__msgpack_chunks__")
_MAX_CHUNK_BYTES = 2 ** 30


def _np_from(shape, dtype_name, buf):
    if dtype_name == "bfloat16":                       # numpy has no bf16: keep the bits
        return torch.frombuffer(bytearray(buf), dtype=torch.bfloat16).reshape(tuple(shape))
    return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(tuple(shape))


def _ext_hook(code, data):
    import msgpack
    if code == _EXT_NDARRAY:
        shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
        return _np_from(shape, dtype_name, buf)
    if code == _EXT_NPSCALAR:
        dtype_name, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, dtype=np.dtype(dtype_name))[0]
    if code == _EXT_COMPLEX:
        re_, im_ = msgpack.unpackb(data, raw=False)
        return complex(re_, im_)
    return msgpack.ExtType(code, data)


def _index_dict_to_list(d):
    """flax `_dict_to_tuple`: {'0': a, '1': b, ...} (or a list / tuple) -> [a, b, ...] in index order."""
    if isinstance(d, dict):
        get = lambda i: d[str(i)] if str(i) in d else d[i]
        return [get(i) for i in range(len(d))]
    return list(d)


def _unchunk(x):
    if isinstance(x, dict):
        if any(k in x for k in _CHUNK_KEYS):           # flax.serialization._unchunk
            shape = tuple(int(v) for v in _index_dict_to_list(x["shape"]))
            parts = _index_dict_to_list(x["chunks"])
            if isinstance(parts[0], torch.Tensor):
                return torch.cat([p.reshape(-1) for p in parts]).reshape(shape)
            return np.concatenate([np.asarray(p).reshape(-1) for p in parts]).reshape(shape)
        return {k: _unchunk(v) for k, v in x.items()}
    return x


def flax_from_bytes(blob):
    """flax.serialization.msgpack_restore."""
    import msgpack
    return _unchunk(msgpack.unpackb(blob, ext_hook=_ext_hook, raw=False, strict_map_key=False))


def read_flax_stream(path_or_file):
    """tux StreamingCheckpointer file -> {'a/b/c': ndarray | bf16 tensor}.  Streams: one leaf
    in memory at a time while decoding."""
    import msgpack
    f = open(path_or_file, "rb") if isinstance(path_or_file, (str, os.PathLike)) else path_or_file
    try:
        out = {}
        unpacker = msgpack.Unpacker(f, read_size=1 << 26, max_buffer_size=0, raw=False)
        for key, blob in unpacker:
            out["/".join(str(k) for k in key)] = flax_from_bytes(blob)
        return out
    finally:
        if f is not path_or_file:
            f.close()


def write_flax_stream(path_or_file, flat, max_chunk_bytes=_MAX_CHUNK_BYTES):
    """Inverse of read_flax_stream for numpy / torch leaves (used by the tests and for exporting
    harness weights in the reference's format).  Leaves above `max_chunk_bytes` are split the way
    flax.serialization._chunk splits them."""
    import msgpack
    f = open(path_or_file, "wb") if isinstance(path_or_file, (str, os.PathLike)) else path_or_file
    try:
        packer = msgpack.Packer(use_bin_type=True)
        for key, val in flat.items():
            if isinstance(val, torch.Tensor):
                name = {torch.bfloat16: "bfloat16", torch.float32: "float32", torch.float16: "float16",
                        torch.int32: "int32", torch.int64: "int64"}[val.dtype]
                raw = val.detach().cpu().contiguous().flatten().view(torch.uint8).numpy().tobytes()
                shape = tuple(val.shape)
            else:
                val = np.ascontiguousarray(val)
                name, raw, shape = val.dtype.name, val.tobytes(), val.shape
            ext = lambda shp, buf: msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(shp), name, buf), use_bin_type=True))
            if len(raw) > max_chunk_bytes:             # flax.serialization._chunk
                item = len(raw) // max(1, int(np.prod(shape)))
                per = max(1, max_chunk_bytes // item) * item
                pieces = [raw[i:i + per] for i in range(0, len(raw), per)]
                tree = {"__msgpack_chunked_array__": True,
                        "shape": {str(i): int(d) for i, d in enumerate(shape)},
                        "chunks": {str(i): ext((len(pc) // item,), pc) for i, pc in enumerate(pieces)}}
                leaf = msgpack.packb(tree, use_bin_type=True)
            else:
                leaf = msgpack.packb(ext(shape, raw), use_bin_type=True)
            f.write(packer.pack((tuple(key.split("/")), leaf)))
    finally:
        if f is not path_or_file:
            f.close()


def flax_llama_to_lwm(flat, prefix="params/", param_scan_axis=0):
    """Flat flax names of FlaxLLaMAForCausalLM (lwm/llama.py:982-1106: `transformer/wte/embedding`,
    `transformer/h/<i>/attention/wq/kernel`, ..., `transformer/ln_f/kernel`, `lm_head/kernel`)
    -> harness names.  Layouts already match (flax Dense kernels are (in, out)).

    scan_layers=True -- the reference's default (lwm/llama.py:158) and what every launcher sets --
    stores ONE stacked leaf per parameter under `transformer/h/scan_decoder/...` (nn.scan named
    'scan_decoder', lwm/llama.py:927-941) with the layer index on axis `param_scan_axis`
    (lwm/llama.py:159); those are unstacked into h.<i>.* here."""
    out = {}
    for k, v in flat.items():
        if not k.startswith(prefix):
            continue                                   # optimizer state, step
        k = k[len(prefix):]
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.array(v))
        m = re.match(r"^transformer/h/scan_decoder/(attention|feed_forward)/(w[qkvo123])/kernel$", k)
        if m:
            for i, layer in enumerate(t.unbind(param_scan_axis)):
                out[f"h.{i}.{m.group(1)}.{m.group(2)}"] = layer
            continue
        m = re.match(r"^transformer/h/scan_decoder/(attention_norm|ffn_norm)/kernel$", k)
        if m:
            for i, layer in enumerate(t.unbind(param_scan_axis)):
                out[f"h.{i}.{m.group(1)}.kernel"] = layer
            continue
        if k == "transformer/wte/embedding":
            out["wte"] = t
        elif k == "transformer/vte/embedding":          # FlaxVideoLLaMA: VQGAN code embedding (lwm/vision_llama.py:264-270)
            out["vte"] = t
        elif k == "vision_head/kernel":                 # ... and its output head (lwm/vision_llama.py:354-360)
            out["vision_head"] = t
        elif k == "transformer/ln_f/kernel":
            out["ln_f.kernel"] = t
        elif k == "lm_head/kernel":
            out["lm_head"] = t
        else:
            m = re.match(r"^transformer/h/(\d+)/(attention|feed_forward)/(w[qkvo123])/kernel$", k)
            if m:
                out[f"h.{m.group(1)}.{m.group(2)}.{m.group(3)}"] = t
                continue
            m = re.match(r"^transformer/h/(\d+)/(attention_norm|ffn_norm)/kernel$", k)
            if not m:
                raise KeyError(f"unexpected flax parameter {k!r}")
            out[f"h.{m.group(1)}.{m.group(2)}.kernel"] = t
    return out


# ------------------------------------------------------------------ pickles written from jax
def _reconstruct_array(fun, args, arr_state, aval_state):
    """jax/_src/array.py `_reconstruct_array` minus the device_put: the numpy value."""
    value = fun(*args)
    value.__setstate__(arr_state)
    return value


class _FrozenDictShim(dict):
    def __setstate__(self, state):
        self.update(state.get("_dict", state) if isinstance(state, dict) else state)


# Everything a pickled parameter tree / config dict legitimately references.  Anything else --
# os.system, builtins.eval, a reduce to subprocess.Popen ... -- is refused: loading a checkpoint
# must not be able to run code.
_PICKLE_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"),
    ("builtins", "frozenset"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"),
    ("builtins", "str"), ("builtins", "bytes"), ("builtins", "complex"), ("builtins", "slice"),
}


class _ConfigDictShim(dict):
    """ml_collections.ConfigDict pickles as an object whose state holds `_fields`."""

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state.get("_fields", state))


class _TreeUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("jax") and name == "_reconstruct_array":
            return _reconstruct_array
        if module.startswith("flax") and name == "FrozenDict":
            return _FrozenDictShim
        if module.startswith("ml_collections") and name in ("ConfigDict", "FrozenConfigDict"):
            return _ConfigDictShim
        if module.startswith("numpy") and name in ("dtype", "ndarray") or \
                (module.startswith("numpy.dtypes") and name.endswith("DType")):
            return super().find_class(module, name)
        if (module, name) in _PICKLE_ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"{module}.{name}: not allowed in a parameter / config pickle")


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    return x


def load_pickle_tree(path_or_bytes):
    """The VQGAN checkpoint (lwm/vqgan.py:19) -> nested dict of numpy arrays, whether it was
    pickled from numpy or from jax arrays / FrozenDict.  jax is neither needed nor imported."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        f = io.BytesIO(path_or_bytes)
        return _plain(_TreeUnpickler(f).load())
    with open(path_or_bytes, "rb") as f:
        return _plain(_TreeUnpickler(f).load())
