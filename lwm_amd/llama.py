"""Model HARNESS: the LLaMA transformer of lwm/llama.py assembled from this package's
operators, so that the hot path can be validated and timed in its real position
(BASELINE config #1: 2-layer slice of LWM-7B, S = 4096).  It is not a product subsystem of
its own: every projection is a plain library GEMM (hipBLASLt through torch.matmul), the
embedding a gather; what is hand-written HIP is what sits between them -- RMSNorm, RoPE,
RingAttention, the SwiGLU gate, the chunked lm_head loss.

Names, parameter layouts and config knobs follow the reference: flax Dense kernels are
(in, out) (lwm/llama.py:390-421, :631-655); `wte`, `h.<i>.attention.{wq,wk,wv,wo}`,
`h.<i>.feed_forward.{w1,w2,w3}`, `h.<i>.{attention_norm,ffn_norm}.kernel`, `ln_f.kernel`,
`lm_head.kernel` (lwm/llama.py:664-744, :982-1106).
"""
from __future__ import annotations

import os

import torch

from . import ops as _ops
from .llama_ops import (LLaMAMLP, RMSNorm, apply_rotary_emb, chunked_lm_head_loss, dense, dense_fused, dense_multi,
                        fused_dense_ok, precompute_freqs_cis, qkv_rope, rmsnorm_residual, swiglu)
from .ringattention import (blockwise_feedforward, concatenate_to_cache, ringattention,
                            ringattention_inference, sp_layout_is_explicit, sp_positions, sp_size_rank)

# The model sizes of lwm/llama.py:33-130:
# name: (hidden, intermediate, layers, heads, max_sequence_length, rms_norm_eps)
_SIZES = {"200m": (1024, 2048, 14, 8, 2048, 1e-6), "1b": (2048, 5504, 22, 16, 2048, 1e-6),
          "3b": (3200, 8640, 26, 32, 2048, 1e-6), "7b": (4096, 11008, 32, 32, 4096, 1e-6),
          "13b": (5120, 13824, 40, 40, 2048, 1e-6), "30b": (6656, 17920, 60, 52, 2048, 1e-6),
          "65b": (8192, 22016, 80, 64, 2048, 1e-5), "debug": (256, 256, 2, 2, 2048, 1e-6)}
LLAMA_STANDARD_CONFIGS = {
    name: dict(vocab_size=32000, hidden_size=d, intermediate_size=f, num_hidden_layers=L, num_attention_heads=h,
               max_sequence_length=s, initializer_range=0.02, rms_norm_eps=eps, use_cache=True,
               tie_word_embeddings=False)
    for name, (d, f, L, h, s, eps) in _SIZES.items()}


def parse_config_updates(text):
    """`--update_llama_config` (lwm/train.py:120-121, scripts/run_*.sh): a string such as
    "dict(theta=10000000,max_sequence_length=131072,scan_attention=True)" or a dict literal.
    The reference eval()s it; here only `dict(name=literal, ...)` / `{...}` of Python literals is
    accepted."""
    import ast
    text = text.strip()
    if not text:
        return {}
    node = ast.parse(text, mode="eval").body
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "dict" and not node.args:
        out = {}
        for kw in node.keywords:
            if kw.arg is None:
                raise ValueError("update_llama_config: **expansion is not accepted")
            out[kw.arg] = ast.literal_eval(kw.value)
        return out
    val = ast.literal_eval(node)
    if not isinstance(val, dict):
        raise ValueError("update_llama_config must be dict(...) or a dict literal")
    return val


class LLaMAConfig:
    """lwm/llama.py:133-199: same field names and defaults (dropout fields are 0 and unused)."""

    def __init__(self, vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, max_sequence_length=4096, rms_norm_eps=1e-6, initializer_range=0.02,
                 use_cache=True, bos_token_id=0, eos_token_id=1, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                 tie_word_embeddings=False, scan_attention=True, scan_mlp=True, scan_query_chunk_size=1024,
                 scan_key_chunk_size=1024, scan_mlp_chunk_size=1024, scan_layers=True, param_scan_axis=0, mesh_dim=None,
                 theta=10000, **kwargs):
        # (every keyword and default of the reference's signature -- tests/test_golden.py holds them to the reference's own
        # __init__; scan_layers / param_scan_axis describe the checkpoint layout, the *_pdrop fields are 0 and unused)
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.max_sequence_length, self.rms_norm_eps = max_sequence_length, rms_norm_eps
        self.initializer_range = initializer_range
        self.use_cache, self.bos_token_id, self.eos_token_id = use_cache, bos_token_id, eos_token_id
        self.resid_pdrop, self.embd_pdrop, self.attn_pdrop = resid_pdrop, embd_pdrop, attn_pdrop
        self.tie_word_embeddings = tie_word_embeddings
        self.scan_attention, self.scan_mlp = scan_attention, scan_mlp
        self.scan_query_chunk_size, self.scan_key_chunk_size = scan_query_chunk_size, scan_key_chunk_size
        self.scan_mlp_chunk_size, self.theta = scan_mlp_chunk_size, theta
        self.scan_layers, self.param_scan_axis, self.mesh_dim = scan_layers, param_scan_axis, mesh_dim
        for k, v in kwargs.items():
            setattr(self, k, v)
        self._validate()

    def _validate(self):
        """The attention kernels are written for head_dim 128 (LWM-7B, lwm/llama.py:70-81; also the 13b /
        30b / 65b sizes).  A size they cannot run -- the 3b table entry has head_dim 100 -- is refused
        here, when the config is made, not at the first kernel launch."""
        h, d = self.num_attention_heads, self.hidden_size
        if h <= 0 or d % h or d // h != 128:
            raise NotImplementedError(
                f"hidden_size {d} / num_attention_heads {h}: head_dim must be 128 for the MI355X attention "
                f"kernels (lwm_amd/csrc/attn_common.h); this is the '3b' entry of the reference's size table")

    def update(self, updates):
        """ConfigDict-style in-place update; a string is parsed like --update_llama_config."""
        if isinstance(updates, str):
            updates = parse_config_updates(updates)
        for k, v in dict(updates).items():
            setattr(self, k, v)
        self._validate()
        return self

    def to_dict(self):
        return {k: v for k, v in vars(self).items() if not k.startswith("_")}

    @classmethod
    def from_dict(cls, d):
        return cls(**dict(d))

    @classmethod
    def load_config(cls, path, **updates):
        """lwm/llama.py:300-312: a standard size name, 'json::<file>' or 'pickle::<file>' (the latter
        holds {'llama_config': {...}})."""
        if path in LLAMA_STANDARD_CONFIGS:
            cfg = dict(LLAMA_STANDARD_CONFIGS[path])
        else:
            if "::" not in path:
                raise ValueError(f"unknown model size {path!r}; expected one of {sorted(LLAMA_STANDARD_CONFIGS)} "
                                 "or json::<file> / pickle::<file>")
            load_type, load_path = path.split("::", 1)
            if load_type == "json":
                import json
                with open(load_path) as f:
                    cfg = json.load(f)
            elif load_type == "pickle":
                from .weights import load_pickle_tree
                cfg = dict(load_pickle_tree(load_path)["llama_config"])
            else:
                raise ValueError(f"Unsupported load config type: {load_type}")
        cfg.update(updates)
        return cls(**cfg)


def _multi_rank():
    """More than one rank along the bound "sp" axis (never the WORLD size: see set_sp_group)."""
    return sp_size_rank("sp")[0] > 1


def _dense(i, o, std, dtype):
    return torch.nn.Parameter(torch.randn(i, o, dtype=torch.float32).mul_(std).to(dtype))


def key_padding_bias(attention_mask):
    """lwm/llama.py:527-537: the (B, S_global) key mask as the additive bias the blockwise branch hands to ringattention,
    (B,1,1,S_global) with 0 where attention_mask > 0 and finfo.min elsewhere."""
    B, S = attention_mask.shape
    return torch.where(attention_mask.reshape(B, 1, 1, S) > 0, 0.0, torch.finfo(torch.float32).min)


def cached_visibility(B, Q, max_len, q0, attention_mask, device):
    """lwm/llama.py:574-592 with a cache present: key j of the max_len cache rows is visible to query i of the block iff
    j <= q0 + i (q0 = cache_index, plus the block's offset when the update is sharded) and attention_mask[b, j] > 0
    (attention_mask: (B, >= max_len) or None).  -> (B,1,Q,max_len) bool."""
    ar = torch.arange(max_len, device=device)
    mask = (ar[None, :] <= (torch.arange(Q, device=device) + q0)[:, None])[None, None].expand(B, 1, Q, max_len)
    if attention_mask is not None:
        mask = mask & (attention_mask[:, None, None, :max_len] > 0)
    return mask


class LLaMAAttention(torch.nn.Module):
    """FlaxLLaMAAttention.__call__, training branch (lwm/llama.py:494-570, :616-617)."""

    def __init__(self, cfg: LLaMAConfig, dtype=torch.bfloat16):
        super().__init__()
        self.cfg, self.dtype = cfg, dtype
        d = cfg.hidden_size
        self.num_heads, self.head_dim = cfg.num_attention_heads, d // cfg.num_attention_heads
        self.wq, self.wk, self.wv, self.wo = (_dense(d, d, cfg.initializer_range, dtype) for _ in range(4))

    def forward(self, x, freqs_cis, attention_mask=None, segment_ids=None, position_ids=None, cache=None, layout=None,
                residual=None):
        """layout (extension): which global positions this rank's rows are -- None = the rule bound to the "sp" axis
        (lwm_amd.ringattention.set_sp_group), "contiguous" | "zigzag", or a lwm_amd.ring.SeqLayout (a packed batch's
        balanced ownership table, lwm_amd.ring.balanced_layout).
        residual (extension): the block's `x`, added to the result (lwm/llama.py:726) -- in the wo GEMM's epilogue."""
        B, S, d = x.shape
        split = lambda t: t.reshape(B, S, self.num_heads, self.head_dim)      # reshape, no transpose (:434-438)
        fused = cache is None and freqs_cis.is_cuda and fused_dense_ok(x, (self.wq, self.wk, self.wv, self.wo))
        if fused:
            # wq | wk | wv as ONE library GEMM into a (B,S,3,H,D) buffer, RoPE on its q | k part in one launch; the
            # attention kernels take the three strided views as they are
            xq, xk, xv = qkv_rope(x, self.wq, self.wk, self.wv, freqs_cis, position_ids, self.num_heads)
            if _multi_rank():
                xq, xk, xv = xq.contiguous(), xk.contiguous(), xv.contiguous()      # (the exchange sends whole blocks)
        else:
            xq, xk, xv = (split(t) for t in dense_multi(x, (self.wq, self.wk, self.wv)))
            xq, xk = apply_rotary_emb(xq, xk, freqs_cis, position_ids)
            xv = xv.contiguous()
        if cache is not None:
            out = dense(self._cached(xq, xk, xv, attention_mask, cache).reshape(B, S, d), self.wo)
            return out if residual is None else residual + out
        bias = None
        if attention_mask is not None:                                        # (:527-537)
            # the bias is NOT sharded over "sp" (lwm/llama.py:563): it covers the global sequence
            n_sp = sp_size_rank("sp")[0]
            if attention_mask.shape[-1] != S * n_sp:
                raise ValueError(f"attention_mask covers {attention_mask.shape[-1]} positions; the sequence ring "
                                 f"needs the global length {S * n_sp} (= local {S} x sp {n_sp}) on every rank")
            bias = key_padding_bias(attention_mask.reshape(B, S * n_sp))
        out = ringattention(xq, xk, xv, bias, segment_ids, axis_name="sp", float32_logits=True,
                            cache_idx=None,
                            blockwise_kwargs=dict(causal_block_size=1, deterministic=True, attn_pdrop=0.0,
                                                  query_chunk_size=self.cfg.scan_query_chunk_size,
                                                  key_chunk_size=self.cfg.scan_key_chunk_size),
                            layout=layout)
        if fused:
            return dense_fused(out.reshape(B, S, d), (self.wo,), residual)
        out = out.reshape(B, S, d) @ self.wo
        return out if residual is None else residual + out

    def _cached(self, xq, xk, xv, attention_mask, cache):
        """The inference branch (lwm/llama.py:571-614): mask = key j visible to query i iff
        j <= cache_index + i and attention_mask[j] (:577-592); the new keys/values are written into
        the cache at cache_index (:440-492); attention runs over the WHOLE cache (kv_len =
        max_length != q_len).  `cache`: dict(cached_key, cached_value (B, max_length, H, D),
        cache_index int); `attention_mask`: (B, max_length), ones beyond the prompt (:1121-1124)."""
        B, Q = xq.shape[:2]
        ck, cv = cache["cached_key"], cache["cached_value"]
        if "index_dev" in cache:
            # hipGraph-capturable decode step: cache_index lives on the device (one int32 shared by all
            # layers), the mask was built from it once for this step, and the new row is written by
            # lwm_kv_cache_write_at -- no host value changes between replays.
            _ops.kv_cache_write_at(ck, xk.contiguous(), cache["index_dev"])
            _ops.kv_cache_write_at(cv, xv, cache["index_dev"])
            return ringattention_inference(xq.contiguous(), ck, cv, cache["mask_dev"], axis_name="sp")
        n_sp, r_sp = sp_size_rank("sp")
        # the cache is sharded over "sp" (lwm/llama.py:454-467): every rank holds max_length/sp rows
        max_len, idx = ck.shape[1] * n_sp, int(cache["cache_index"])
        if xq.dtype == torch.float32 and n_sp > 1:
            raise NotImplementedError("cached inference in float32 runs on one rank (the sharded decode / dense-mask kernels "
                                      "take bf16 operands): --dtype=bf16, or mesh_dim without an sp axis")
        # (float32 -- the reference's --dtype=fp32 -- takes the structured form for decode steps too: the f32 flavour of the
        #  op evaluates `key <= cache_index + query AND attention_mask[key]` itself, there is no f32 dense-mask kernel)
        if (Q > 1 or xq.dtype == torch.float32) and n_sp == 1:
            # prefill into the cache: the same mask, handed over as its structure (see
            # ringattention_inference) -- key tiles past cache_index + Q are never read
            cache["cache_index"] = concatenate_to_cache(ck, cv, xk.contiguous(), xv, idx, axis_name="sp")
            kvld = None if attention_mask is None else attention_mask[:, :max_len]
            return ringattention_inference(xq.contiguous(), ck, cv, None, axis_name="sp", causal_offset=idx,
                                           key_valid=kvld)
        # a sharded prefill block (Q > 1) holds the queries [r*Q, (r+1)*Q) of the update; a decode query
        # is replicated (lwm/llama.py:599)
        q0 = idx + (r_sp * Q if Q > 1 else 0)
        mask = cached_visibility(B, Q, max_len, q0, attention_mask, xq.device)
        cache["cache_index"] = concatenate_to_cache(ck, cv, xk.contiguous(), xv, idx, axis_name="sp")
        return ringattention_inference(xq.contiguous(), ck, cv, mask, axis_name="sp")


class LLaMABlock(torch.nn.Module):
    """FlaxLLaMABlock (lwm/llama.py:664-744): pre-norm residual block, blockwise FFN."""

    def __init__(self, cfg: LLaMAConfig, dtype=torch.bfloat16):
        super().__init__()
        self.cfg = cfg
        self.attention_norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype)
        self.attention = LLaMAAttention(cfg, dtype)
        self.ffn_norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype)
        self.feed_forward = LLaMAMLP(cfg.hidden_size, cfg.intermediate_size, dtype, cfg.initializer_range)

    def forward(self, x, freqs_cis, attention_mask=None, segment_ids=None, position_ids=None, cache=None, layout=None):
        if cache is None and x.is_cuda and x.dtype == torch.bfloat16:
            # training / uncached branch: `x` goes through each norm as (norm(x), x) so that the residual branch's gradient
            # is added inside the RMSNorm backward kernel, and each residual add rides in the epilogue of the GEMM before it
            h, x = rmsnorm_residual(self.attention_norm, x)
            x = self.attention(h, freqs_cis, attention_mask, segment_ids, position_ids, None, layout, residual=x)
            h, x = rmsnorm_residual(self.ffn_norm, x)
            if self.cfg.scan_mlp and h.shape[1] >= self.cfg.scan_mlp_chunk_size:      # (:728-734)
                return x + blockwise_feedforward(self.feed_forward, h, self.cfg.scan_mlp_chunk_size)
            return self.feed_forward(h, residual=x)
        x = x + self.attention(self.attention_norm(x), freqs_cis, attention_mask, segment_ids, position_ids, cache, layout)
        h = self.ffn_norm(x)
        if self.cfg.scan_mlp and h.shape[1] >= self.cfg.scan_mlp_chunk_size:      # (:728-734)
            ff = blockwise_feedforward(self.feed_forward, h, self.cfg.scan_mlp_chunk_size)
        else:
            ff = self.feed_forward(h)
        return x + ff


class LLaMAForCausalLM(torch.nn.Module):
    """FlaxLLaMAModule + lm_head (lwm/llama.py:982-1106), loss of lwm/train.py:171-181."""

    def __init__(self, cfg: LLaMAConfig, dtype=torch.bfloat16):
        super().__init__()
        self.cfg, self.dtype = cfg, dtype
        self.wte = torch.nn.Parameter(torch.randn(cfg.vocab_size, cfg.hidden_size).mul_(cfg.initializer_range).to(dtype))
        self.h = torch.nn.ModuleList(LLaMABlock(cfg, dtype) for _ in range(cfg.num_hidden_layers))
        self.ln_f = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, dtype)
        self.lm_head = _dense(cfg.hidden_size, cfg.vocab_size, cfg.initializer_range, dtype)
        self._freqs = None

    def _table(self, device):
        if self._freqs is None or self._freqs.device != device:
            head_dim = self.cfg.hidden_size // self.cfg.num_attention_heads
            self._freqs = precompute_freqs_cis(head_dim, self.cfg.max_sequence_length, self.cfg.theta, device=device)
        return self._freqs

    @staticmethod
    def _ring_position_ids(input_ids, position_ids, cache, layout=None):
        """position_ids of a sequence-sharded training batch: the GLOBAL positions of this rank's rows under the
        ownership rule bound to the "sp" axis (lwm_amd.ringattention.set_sp_group: zigzag by default, or the
        reference's contiguous blocks, lwm/llama.py:560-562) or under `layout` -- the rows were cut with `sp_shard`,
        so RoPE and the causal mask both see where each token really sits.

        A caller that brings its OWN position_ids to a sharded training batch cut the rows itself; the masks are
        evaluated at the positions the ownership rule gives those rows, so the rule has to be NAMED then -- `layout=`
        here, set_sp_group(group, layout=...) or LWM_SP_LAYOUT.  With the defaulted rule (zigzag) a batch sharded the
        reference's way would get RoPE at contiguous positions and causal masks at zigzag ones, without an error
        anywhere: refused instead."""
        n_sp = sp_size_rank("sp")[0]
        if n_sp > 1 and position_ids is None and cache is None:
            B, S = input_ids.shape
            position_ids = sp_positions(S, "sp", input_ids.device, layout).to(torch.int32)[None].expand(B, S).contiguous()
        elif n_sp > 1 and cache is None and layout is None and not sp_layout_is_explicit("sp"):
            raise ValueError(
                "position_ids were given for a sequence-sharded batch, but nobody said which global positions this "
                "rank's rows ARE: pass layout= ('contiguous' = the reference's blocks, lwm/llama.py:560-562; 'zigzag' = "
                "what sp_shard cuts by default; or a SeqLayout), or name the rule once with "
                "set_sp_group(group, layout=...) / LWM_SP_LAYOUT")
        if n_sp > 1 and position_ids is None:
            raise ValueError("cached inference over a sequence ring needs explicit global position_ids")
        return n_sp, position_ids

    def hidden_states(self, input_ids, attention_mask=None, segment_ids=None, position_ids=None, cache=None, layout=None):
        n_sp, position_ids = self._ring_position_ids(input_ids, position_ids, cache, layout)
        x = torch.nn.functional.embedding(input_ids.long(), self.wte)
        fc = self._table(x.device)
        if cache is not None and self._fused_decode_ok(x, n_sp):
            return self.ln_f(self._decode_layers_fused(x, fc, attention_mask, position_ids, cache))
        for i, blk in enumerate(self.h):
            x = blk(x, fc, attention_mask, segment_ids, position_ids, None if cache is None else cache[i], layout)
        return self.ln_f(x)

    def _fused_decode_ok(self, x, n_sp):
        """One token per batch row through the KV cache, bf16, no autograd, one rank: the layers can run as GEMV launch
        pairs that carry their neighbours (lwm_gemv_fused_bf16; LWM_DECODE_FUSED=0 issues every launch on its own)."""
        d = self.cfg.hidden_size
        # (d / 128 partial sums of squares per row ride along: lwm_gemv_fused_bf16 takes at most 64 of them; the weights
        # must be what the GEMV streams -- contiguous bf16 -- or the per-block path runs)
        return (os.environ.get("LWM_DECODE_FUSED", "1") == "1" and x.shape[1] == 1 and x.shape[0] <= 4 and n_sp == 1
                and x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and d % 128 == 0
                and d <= 8192 and self.cfg.intermediate_size % 32 == 0 and self.cfg.intermediate_size <= 12288
                and all(p.dtype == torch.bfloat16 and p.is_contiguous() for p in self.h[0].parameters() if p.dim() == 2))

    @staticmethod
    def _norm_weight_bf16(norm):
        """the RMSNorm weight as the kernels read it (bf16), cached on the module while the parameter is unchanged"""
        w = norm.kernel
        hit = getattr(norm, "_lwm_bf16", None)
        if hit is None or hit[0] != (w._version, w.data_ptr()):
            hit = ((w._version, w.data_ptr()), w.detach().to(torch.bfloat16).contiguous())
            norm._lwm_bf16 = hit
        return hit[1]

    def _decode_layers_fused(self, x, fc, attention_mask, position_ids, cache):
        """The blocks of a cached one-token step (lwm/llama.py:704-744 with q_len = 1) with each RMSNorm folded into the
        x load of the projections that follow it and each residual add into the reduction of the projection before it:
        per layer 4 fewer launches of the ~19; same roundings as the separate kernels (rstd sums in another order)."""
        from .llama_ops import gemv_fused
        B, _, d = x.shape
        H, D = self.cfg.num_attention_heads, d // self.cfg.num_attention_heads
        x2 = x.reshape(B, d)
        ss = torch.zeros(B, 32, dtype=torch.float32, device=x.device)
        ss[:, 0] = x2.float().pow(2).sum(-1)
        eps = self.cfg.rms_norm_eps
        for i, blk in enumerate(self.h):
            att, mlp = blk.attention, blk.feed_forward
            q, k, v = gemv_fused(x2, (att.wq, att.wk, att.wv), norm=(ss, self._norm_weight_bf16(blk.attention_norm), eps))
            split = lambda t: t.reshape(B, 1, H, D)
            xq, xk = apply_rotary_emb(split(q), split(k), fc, position_ids)
            a = att._cached(xq, xk, split(v).contiguous(), attention_mask, cache[i]).reshape(B, d)
            (x2,), ss = gemv_fused(a, (att.wo,), residual=x2, want_ss=True)
            gate, up = gemv_fused(x2, (mlp.w1, mlp.w3), norm=(ss, self._norm_weight_bf16(blk.ffn_norm), eps))
            (x2,), ss = gemv_fused(swiglu(gate, up), (mlp.w2,), residual=x2, want_ss=True)
        return x2.reshape(B, 1, d)

    def init_cache(self, batch_size, max_length, device=None):
        """FlaxLLaMAPreTrainedModel.init_cache (lwm/llama.py:810-825): per layer, zeroed
        (B, max_length, H, D) key/value caches and cache_index = 0."""
        device = device or self.wte.device
        H = self.cfg.num_attention_heads
        D = self.cfg.hidden_size // H
        n_sp = sp_size_rank("sp")[0]
        if max_length % n_sp:
            raise ValueError(f"max_length {max_length} is not divisible by the sp ring size {n_sp}")
        # sharded over "sp": each rank holds its contiguous max_length/sp rows (lwm/llama.py:454-467)
        z = lambda: torch.zeros(batch_size, max_length // n_sp, H, D, dtype=self.dtype, device=device)
        return [dict(cached_key=z(), cached_value=z(), cache_index=0) for _ in self.h]

    @torch.no_grad()
    def generate(self, input_ids, attention_mask=None, max_new_tokens=16, max_length=None, return_logits=False,
                 graph=False):
        """Greedy decoding through the KV cache: prepare_inputs_for_generation /
        update_inputs_for_generation of the reference (lwm/llama.py:1113-1137) + argmax.
        Prefill writes the prompt's keys/values at cache_index 0 and attends over the whole
        (B, max_length) cache under the dense mask; every further step feeds one token.

        graph=True: the one-token step (every layer's RMSNorm, projections, RoPE, cache write, decode
        attention, MLP, and the head) is captured ONCE in a hipGraph and replayed per token; the
        cache index, position and current token advance on the device inside the graph.  A decode
        step is some 15 launches per layer of a few microseconds of work each -- launch-bound when
        issued one by one.  Single-rank only (the cross-rank combine is not captured)."""
        B, S = input_ids.shape
        max_length = max_length or (S + max_new_tokens)
        dev = input_ids.device
        if graph and self.dtype == torch.float32:
            raise NotImplementedError("generate(graph=True) captures the bf16 decode kernels; a float32 model decodes eagerly "
                                      "(graph=False) through the f32 flavour of the attention op")
        cache = self.init_cache(B, max_length, dev)
        ext = torch.ones(B, max_length, dtype=torch.int32, device=dev)
        if attention_mask is not None:
            pos = attention_mask.to(torch.int32).cumsum(-1) - 1
            ext[:, :S] = attention_mask.to(torch.int32)
        else:
            pos = torch.arange(S, dtype=torch.int32, device=dev)[None].expand(B, S)
        pos = pos.clamp_min(0).to(torch.int32).contiguous()
        head = self.lm_head        # f32 logits: bf16 operands are exact in f32, so `dense(.., torch.float32)` on the
        #                            bf16 kernel forms the same products as h.float() @ kernel.float() and reads half the bytes
        tokens, logits_out = [input_ids], []

        def emit(logits):
            if return_logits:
                logits_out.append(logits.clone())
            nxt = logits.argmax(-1, keepdim=True).to(input_ids.dtype)
            tokens.append(nxt.clone())
            return nxt

        # prefill (and, without graph, every later step): host-side cache_index
        step_in = input_ids
        n_eager = max_new_tokens if not graph else min(1, max_new_tokens)
        for _ in range(n_eager):
            h = self.hidden_states(step_in, ext, None, pos, cache)
            step_in = emit(dense(h[:, -1], head, torch.float32))
            pos = (pos[:, -1:] + 1).contiguous()
        if graph and max_new_tokens > 1:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                raise NotImplementedError("generate(graph=True) is single-rank")
            idx = torch.tensor([int(cache[0]["cache_index"])], dtype=torch.int32, device=dev)
            ar = torch.arange(max_length, device=dev, dtype=torch.int32)
            tok, posd = step_in.clone(), pos.clone()
            dcache = [dict(cached_key=c["cached_key"], cached_value=c["cached_value"], index_dev=idx) for c in cache]

            def step():
                mask = ((ar[None, :] <= idx) & (ext > 0))[:, None, None, :]
                for c in dcache:
                    c["mask_dev"] = mask
                logits = dense(self.hidden_states(tok, ext, None, posd, dcache)[:, -1], head, torch.float32)
                tok.copy_(logits.argmax(-1, keepdim=True).to(tok.dtype))
                posd.add_(1)
                idx.add_(1)
                return logits

            # one eager step on a side stream (library handles, first-call state), then capture
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                emit(step())
            torch.cuda.current_stream(dev).wait_stream(side)
            if max_new_tokens > 2:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    static_logits = step()
                for _ in range(max_new_tokens - 2):
                    g.replay()
                    emit(static_logits)
        out = torch.cat(tokens, dim=1)
        return (out, torch.stack(logits_out, 1)) if return_logits else out

    def loss(self, input_tokens, target_tokens, loss_masks=None, attention_mask=None, segment_ids=None,
             position_ids=None, chunk=8192, layout=None, sp_sharded=True):
        """lwm/train.py:171-181.  Over a sequence ring (more than one rank on the "sp" axis) the arguments are this
        rank's ROWS of the batch (sp_shard) and the result is this rank's SHARE of the reference's per-sequence mean:
        the count of valid targets is all-reduced over "sp" (every sp rank must enter the call), and the shares add up
        to the reference's loss.  sp_sharded=False says the tokens are REPLICATED on the sp ranks (evaluation /
        scoring of a short batch on every rank): no collective, each rank gets the whole loss."""
        h = self.hidden_states(input_tokens, attention_mask, segment_ids, position_ids, layout=layout)
        return chunked_lm_head_loss(h, self.lm_head, target_tokens, loss_masks, chunk, sp_sharded=sp_sharded)

def hf_rotary_to_interleaved(w_out_in, num_heads):
    """HF-PyTorch LLaMA checkpoints (README.md:74, scripts/sample_pyt.py:8) store wq/wk for the
    rotate_half RoPE convention; the reference rotates interleaved (even, odd) pairs
    (lwm/llama.py:360-364).  w_out_in: torch Linear weight (out, in).  Returns the flax-layout
    kernel (in, out) whose head dims are re-ordered [0, d/2, 1, d/2+1, ...] so that interleaved
    RoPE on it equals rotate_half RoPE on the original."""
    out_f, in_f = w_out_in.shape
    hd = out_f // num_heads
    w = w_out_in.reshape(num_heads, 2, hd // 2, in_f).transpose(1, 2).reshape(out_f, in_f)
    return w.t().contiguous()
