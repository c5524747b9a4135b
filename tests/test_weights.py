"""Weight I/O (lwm_amd/weights.py) and the HF-transformers anchor.

tests/golden/hf_llama_tiny.npz holds logits / loss / gradients of HF `LlamaForCausalLM` (the
PyTorch implementation the reference points at, scripts/sample_pyt.py:8) for a tiny model whose
weights are a function of a seed (tests/golden/hf_fixture.py).  Here, on CPU: the checkpoint
converter (layout transposes + the rotate_half -> interleaved q/k re-ordering) feeding the fp32
oracle model must reproduce them, which pins oracle/llama_model_ref.py -- and through it RoPE,
RMSNorm, causal attention, SwiGLU and the loss -- against code that is not ours."""
import io
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import hf_fixture as F  # noqa: E402

from lwm_amd import weights as W  # noqa: E402
from lwm_amd.llama import hf_rotary_to_interleaved  # noqa: E402
from oracle import llama_model_ref as M  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "hf_llama_tiny.npz"))


def _oracle_state(requires_grad=False):
    cfg = W.config_from_hf(F.HF_CONFIG)
    st = W.hf_to_lwm(F.state_dict(), cfg.num_attention_heads)
    st = {k: v.clone().requires_grad_(requires_grad) for k, v in st.items()}
    return cfg, st


def test_oracle_model_reproduces_hf_transformers_logits_and_loss():
    cfg, st = _oracle_state()
    ids = F.token_ids()
    logits = M.forward_logits(st, cfg, ids[:, :-1])
    ref = torch.from_numpy(GOLD["logits"])
    assert (logits - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    loss, acc = M.forward_loss(st, cfg, ids[:, :-1], ids[:, 1:])
    assert abs(loss.item() - float(GOLD["loss"])) <= 1e-5 * float(GOLD["loss"])
    assert abs(acc.item() - float(GOLD["accuracy"])) < 1e-6


def test_oracle_model_reproduces_hf_transformers_gradients():
    cfg, st = _oracle_state(requires_grad=True)
    ids = F.token_ids()
    loss, _ = M.forward_loss(st, cfg, ids[:, :-1], ids[:, 1:])
    loss.backward()
    nh = cfg.num_attention_heads
    for hf_name, ours, kind in (("grad_q_proj_0", "h.0.attention.wq", "rotary"),
                                ("grad_k_proj_1", "h.1.attention.wk", "rotary"),
                                ("grad_v_proj_0", "h.0.attention.wv", "linear")):
        g = torch.from_numpy(GOLD[hf_name])
        g = hf_rotary_to_interleaved(g, nh) if kind == "rotary" else g.t()
        got = st[ours].grad
        assert (got - g).abs().max().item() <= 2e-4 * g.abs().max().item(), hf_name


def test_attention_oracle_equals_the_attention_inside_the_hf_anchored_model():
    """Closes the chain HF transformers == oracle model == oracle/attention_ref.dense_attention
    (== blockwise / ring oracle == HIP kernels, tests/test_oracle.py and the -m gpu tests): the
    softmax-attention expression of oracle/llama_model_ref.py, which the HF vectors pin, against
    the fp64 attention oracle on the same q, k, v, packed segments and padding included."""
    import math
    from oracle import attention_ref as R
    g = torch.Generator().manual_seed(3)
    B, S, H, D = 1, 80, 2, 128
    q, k, v = (torch.randn(B, S, H, D, generator=g) for _ in range(3))
    seg = torch.zeros(B, S, dtype=torch.int32)
    seg[:, 30:] = 1
    am = torch.ones(B, S, dtype=torch.int32)
    am[:, 4:7] = 0
    vis = torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None]
    vis = vis & (seg[:, None, :, None] == seg[:, None, None, :]) & (am[:, None, None, :] > 0)
    s_ = (torch.einsum("bqhd,bkhd->bhqk", q, k) / math.sqrt(D)).masked_fill(~vis, float("-inf"))
    a = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s_, dim=-1), v)
    out, _ = R.dense_attention(q.numpy(), k.numpy(), v.numpy(), causal=True, seg_q=seg.numpy(),
                               seg_k=seg.numpy(), key_valid=am.numpy().astype(np.uint8))
    rows = (vis.any(-1)[0, 0]).numpy()              # rows with no visible key are defined as 0 by the oracle
    assert np.abs(a.numpy()[:, rows] - out[:, rows]).max() <= 2e-6


def test_hf_checkpoint_directory_round_trip(tmp_path):
    import json
    from safetensors.torch import save_file
    sd = {k: v.to(torch.bfloat16) for k, v in F.state_dict().items()}
    save_file(sd, str(tmp_path / "model.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps(F.HF_CONFIG))
    got, cfg = W.read_hf_checkpoint(str(tmp_path))
    assert cfg["rope_theta"] == 10000.0 and set(got) == set(sd)
    lw = W.hf_to_lwm(got, cfg["num_attention_heads"])
    assert lw["h.1.feed_forward.w2"].shape == (512, 256) and lw["lm_head"].shape == (256, 384)
    assert torch.equal(lw["h.0.attention.wv"], sd["model.layers.0.self_attn.v_proj.weight"].t())
    with pytest.raises(KeyError):
        W.hf_to_lwm({"model.layers.0.self_attn.qkv.weight": torch.zeros(2, 2)}, 2)
    with pytest.raises(ValueError):
        W.config_from_hf(dict(F.HF_CONFIG, num_key_value_heads=1))


def test_flax_msgpack_stream_round_trip():
    """tux StreamingCheckpointer layout: records (key tuple, flax to_bytes(leaf)); bf16 leaves keep
    their bits; the harness names come out of the flax names."""
    g = torch.Generator().manual_seed(0)
    flat = {
        "params/transformer/wte/embedding": torch.randn(16, 8, generator=g).to(torch.bfloat16),
        "params/transformer/h/0/attention/wq/kernel": torch.randn(8, 8, generator=g),
        "params/transformer/h/0/attention_norm/kernel": np.ones(8, np.float32),
        "params/transformer/ln_f/kernel": np.arange(8, dtype=np.float32),
        "params/lm_head/kernel": torch.randn(8, 16, generator=g),
        "step": np.int32(7),
    }
    flat["step"] = np.asarray(flat["step"])
    buf = io.BytesIO()
    W.write_flax_stream(buf, flat)
    buf.seek(0)
    back = W.read_flax_stream(buf)
    assert set(back) == set(flat)
    assert back["params/transformer/wte/embedding"].dtype == torch.bfloat16
    assert torch.equal(back["params/transformer/wte/embedding"], flat["params/transformer/wte/embedding"])
    assert np.array_equal(back["params/transformer/ln_f/kernel"], flat["params/transformer/ln_f/kernel"])
    names = W.flax_llama_to_lwm(back)
    assert set(names) == {"wte", "h.0.attention.wq", "h.0.attention_norm.kernel", "ln_f.kernel", "lm_head"}
    # a leaf flax would have split into chunks
    import msgpack
    part = lambda a: msgpack.ExtType(1, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes()), use_bin_type=True))
    a = np.arange(12, dtype=np.float32)
    # flax.serialization._chunk: a True flag, shape and chunks as index-keyed dicts
    blob = msgpack.packb({"__msgpack_chunked_array__": True, "shape": {"0": 3, "1": 4},
                          "chunks": {"0": part(a[:5]), "1": part(a[5:10]), "2": part(a[10:])}}, use_bin_type=True)
    assert np.array_equal(W.flax_from_bytes(blob), a.reshape(3, 4))


def test_flax_stream_scan_layers_layout_with_chunked_leaves():
    """What the reference actually writes (scan_layers=True, lwm/llama.py:158, every scripts/run_*.sh):
    one stacked leaf per parameter under transformer/h/scan_decoder with the layer on axis 0, large
    leaves split by flax into index-keyed chunks.  The writer chunks like flax (threshold lowered so a
    test-sized leaf is split); the reader must re-join and the converter must unstack."""
    g = torch.Generator().manual_seed(1)
    L, d, f = 3, 8, 16
    flat = {
        "params/transformer/wte/embedding": torch.randn(32, d, generator=g),
        "params/transformer/ln_f/kernel": torch.ones(d),
        "params/lm_head/kernel": torch.randn(d, 32, generator=g),
    }
    for w in ("wq", "wk", "wv", "wo"):
        flat[f"params/transformer/h/scan_decoder/attention/{w}/kernel"] = torch.randn(L, d, d, generator=g)
    for w, shp in (("w1", (d, f)), ("w2", (f, d)), ("w3", (d, f))):
        flat[f"params/transformer/h/scan_decoder/feed_forward/{w}/kernel"] = torch.randn(L, *shp, generator=g)
    for nm in ("attention_norm", "ffn_norm"):
        flat[f"params/transformer/h/scan_decoder/{nm}/kernel"] = torch.randn(L, d, generator=g)
    buf = io.BytesIO()
    W.write_flax_stream(buf, flat, max_chunk_bytes=400)     # the (3,8,16) f32 leaves become 4 chunks each
    raw = buf.getvalue()
    assert b"__msgpack_chunked_array__" in raw
    back = W.read_flax_stream(io.BytesIO(raw))
    for k, v in flat.items():
        assert torch.equal(torch.as_tensor(np.asarray(back[k])), v), k
    names = W.flax_llama_to_lwm(back)
    assert "h.2.feed_forward.w3" in names and "h.0.attention_norm.kernel" in names
    assert len([n for n in names if n.startswith("h.")]) == L * 9
    assert torch.equal(names["h.1.attention.wk"], flat["params/transformer/h/scan_decoder/attention/wk/kernel"][1])
    assert torch.equal(names["h.2.ffn_norm.kernel"], flat["params/transformer/h/scan_decoder/ffn_norm/kernel"][2])


def test_pickle_loader_refuses_callables():
    """A checkpoint / config pickle must not be able to run code: only array and
    container reconstructors are resolvable."""
    class Evil:
        def __reduce__(self):
            import os
            return os.system, ("true",)
    for payload in (Evil(), {"llama_config": Evil()}):
        with pytest.raises(pickle.UnpicklingError):
            W.load_pickle_tree(pickle.dumps(payload))
    import collections
    ok = W.load_pickle_tree(pickle.dumps({"a": np.arange(3), "b": collections.OrderedDict(x=np.float32(2.0))}))
    assert np.array_equal(ok["a"], np.arange(3)) and float(ok["b"]["x"]) == 2.0


def test_vqgan_pickle_written_from_jax_arrays_loads_without_jax():
    """A pickle that names jax._src.array._reconstruct_array and flax's FrozenDict (what
    pickle.dump of a flax param tree of jax arrays produces) must load as numpy / dict."""
    assert "jax" not in sys.modules
    arr = np.arange(6, dtype=np.float32).reshape(2, 3)
    mods = {}
    for name in ("jax", "jax._src", "jax._src.array", "flax", "flax.core", "flax.core.frozen_dict"):
        mods[name] = types.ModuleType(name)

    def _reconstruct_array(fun, args, arr_state, aval_state):   # stand-in so that pickling can name it
        raise AssertionError("must not be called")
    _reconstruct_array.__module__, _reconstruct_array.__qualname__ = "jax._src.array", "_reconstruct_array"
    mods["jax._src.array"]._reconstruct_array = _reconstruct_array

    class FakeJaxArray:
        def __init__(self, a):
            self.a = a

        def __reduce__(self):
            fun, args, state = self.a.__reduce__()
            return _reconstruct_array, (fun, args, state, {"weak_type": False})

    class FrozenDict(dict):
        def __reduce__(self):
            return FrozenDict, (), {"_dict": dict(self)}
    FrozenDict.__module__, FrozenDict.__qualname__ = "flax.core.frozen_dict", "FrozenDict"
    mods["flax.core.frozen_dict"].FrozenDict = FrozenDict

    sys.modules.update(mods)
    try:
        blob = pickle.dumps(FrozenDict(encoder=FrozenDict(conv_in={"kernel": FakeJaxArray(arr)}),
                                       quantize={"embeddings": arr * 2}))
    finally:
        for name in mods:
            sys.modules.pop(name, None)
    tree = W.load_pickle_tree(blob)
    assert type(tree) is dict and type(tree["encoder"]) is dict
    assert np.array_equal(tree["encoder"]["conv_in"]["kernel"], arr)
    assert np.array_equal(tree["quantize"]["embeddings"], arr * 2)


def test_config_tables_and_update_string(tmp_path):
    """LLaMAConfig.load_config / --update_llama_config (lwm/llama.py:300-312, lwm/train.py:120-121)."""
    import json
    from lwm_amd.llama import LLaMAConfig, parse_config_updates
    c = LLaMAConfig.load_config("7b")
    assert (c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads) == (4096, 11008, 32, 32)
    assert LLaMAConfig.load_config("65b").rms_norm_eps == 1e-5
    # the string of scripts/run_eval_needle.sh:19
    c.update("dict(theta=10000000,max_sequence_length=131072,scan_attention=True,scan_query_chunk_size=1024,"
             "scan_key_chunk_size=1024,scan_mlp=True,scan_mlp_chunk_size=1024,scan_layers=True)")
    assert c.theta == 10000000 and c.max_sequence_length == 131072 and c.scan_layers is True
    assert parse_config_updates("dict(sample_mode='vision',theta=50000000)") == dict(sample_mode="vision", theta=50000000)
    assert parse_config_updates("{'theta': 1e7}") == {"theta": 1e7}
    for bad in ("__import__('os').system('true')", "dict(a=open('x'))", "dict(**{'a': 1})", "[1, 2]"):
        with pytest.raises((ValueError, SyntaxError)):
            parse_config_updates(bad)
    (tmp_path / "c.json").write_text(json.dumps(dict(LLaMAConfig.load_config("debug").to_dict(), theta=5e7)))
    assert LLaMAConfig.load_config(f"json::{tmp_path / 'c.json'}").theta == 5e7
    with pytest.raises(ValueError):
        LLaMAConfig.load_config("yaml::x")
    # head_dim 100: refused when the config is made, not at the first launch (the kernels are head_dim 128)
    with pytest.raises(NotImplementedError):
        LLaMAConfig.load_config("3b")
    for ok in ("13b", "30b", "65b", "debug"):
        LLaMAConfig.load_config(ok)
