"""Drop-in operator surface of the reference's `ringattention` package for the
hot path, with torch.Tensor in place of jax.Array and a torch.distributed
process group in place of the mesh axis name.

Reference call sites:  lwm/llama.py:30 (import), :539-569 (training op),
:729-734 (blockwise_feedforward).  Same argument names, meaning and error
behaviour; tile sizes are the kernels' own (the reference's
query/key_chunk_size only trade memory for speed and do not change results).
"""
import os

import torch

from .ring import (HipBlockOps, SeqLayout, SingleComm, TorchRingComm, cache_update, ring_attention,
                   ring_inference)

_SP_GROUP = {"group": None, "bound": False, "layout": None, "explicit": False}


def set_sp_group(group, layout=None):
    """Bind mesh axis name "sp" (lwm/llama.py:201-203) to a process group (`dist.group.WORLD` for a
    job that is one sequence ring; None unbinds) and choose WHICH positions a rank of it owns:

      layout="zigzag"      rank r holds the half-chunks r and 2n-1-r of the sequence (local rows [0, c/2) and
                           [c/2, c)): causal work is the same on every rank.  The DEFAULT for a group of more than one
                           rank (LWM_SP_LAYOUT overrides): the harness, the CLI entry points and `ringattention` all
                           follow it -- shard token / target / mask tensors with `sp_shard`, take positions from
                           `sp_positions`.
      layout="contiguous"  the reference's ownership, rank r holds [r*c, (r+1)*c) (lwm/llama.py:560-562,
                           lwm/data.py:494-500): under a causal mask rank n-1 then computes ~1.9x the mean at n = 8
                           and every step of the job waits for it.

    Results do not depend on the layout (attention sees global positions, tests/test_ring_gloo.py); only which rank
    computes what does."""
    _SP_GROUP["group"] = group
    _SP_GROUP["bound"] = group is not None
    if layout is None:
        layout = os.environ.get("LWM_SP_LAYOUT") or None
    if layout not in (None, "zigzag", "contiguous"):
        raise ValueError(f"unknown sp layout {layout!r} (zigzag | contiguous)")
    _SP_GROUP["layout"] = layout if group is not None else None
    # (the caller -- or LWM_SP_LAYOUT -- NAMED the rule: what sp_layout_is_explicit() reports, see
    # LLaMAForCausalLM._ring_position_ids)
    _SP_GROUP["explicit"] = layout is not None and group is not None


def sp_layout_is_explicit(axis_name="sp"):
    """True when the ownership rule along the axis was NAMED (set_sp_group(layout=...) / LWM_SP_LAYOUT) rather than
    defaulted.  A caller that brings its own position_ids to a sequence-sharded batch has sharded the rows itself: with
    a defaulted rule nobody has said which positions those rows ARE (the reference's contiguous blocks,
    lwm/llama.py:560-562, or sp_shard's zigzag half-chunks), and a wrong guess is a silently wrong causal mask."""
    if axis_name is None or isinstance(axis_name, str) or axis_name is _SP_GROUP["group"]:
        return bool(_SP_GROUP["explicit"])
    return False


def _resolve_axis(axis_name):
    """axis name -> process group.  An UNBOUND "sp" axis in a multi-process job is an error, not the
    WORLD group: a data-parallel job that never called set_sp_group must not silently run a sequence
    ring across its replicas (the reference's mesh always names the axis, lwm/llama.py:201-203)."""
    if axis_name is None or isinstance(axis_name, str):
        if not _SP_GROUP["bound"]:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                raise RuntimeError(
                    'mesh axis "sp" is not bound to a process group: call '
                    "lwm_amd.ringattention.set_sp_group(group) (lwm_amd.mesh.sp_group builds it from "
                    "--mesh_dim; pass torch.distributed.group.WORLD for a pure sequence ring)")
        return _SP_GROUP["group"]
    return axis_name  # already a ProcessGroup


def sp_size_rank(axis_name="sp"):
    """(size, rank) of this process along the "sp" axis; (1, 0) outside torch.distributed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    g = _resolve_axis(axis_name)
    if g is None:
        return 1, 0
    return dist.get_world_size(g), dist.get_rank(g)


def sp_layout(axis_name="sp", local_len=None):
    """The ownership rule in force along the "sp" axis: "contiguous" for one rank; else what set_sp_group was given
    (also when the axis is handed over as the bound ProcessGroup object itself), else "zigzag" (when `local_len` is given
    and odd -- zigzag needs two half-chunks -- "contiguous").  A ProcessGroup that was never bound with set_sp_group
    has no rule of its own: it gets the reference's contiguous blocks (lwm/llama.py:560-562), as the low-level
    lwm_amd.ring.ring_attention does -- pass `layout=` to ringattention / sp_shard / sp_positions for anything else."""
    n, _ = sp_size_rank(axis_name)
    if n == 1:
        return "contiguous"
    named = axis_name is None or isinstance(axis_name, str)
    if not named and axis_name is not _SP_GROUP["group"]:
        return "contiguous"
    kind = _SP_GROUP["layout"]
    if kind is None:
        kind = "zigzag" if (local_len is None or local_len % 2 == 0) else "contiguous"
    return kind


def sp_positions(local_len, axis_name="sp", device=None, layout=None):
    """Global token positions of this rank's `local_len` rows, int64 (c,): what the loader slices with, what RoPE
    rotates by (the reference's contiguous ownership makes this arange(c) + r*c, lwm/llama.py:1081-1082 + :560-562).
    layout: a lwm_amd.ring.SeqLayout to use instead of the rule bound to the axis (a packed batch's own balanced
    ownership, lwm_amd.ring.balanced_layout -- pass the same object as position_ids' source and to ringattention)."""
    n, r = sp_size_rank(axis_name)
    lay = layout if isinstance(layout, SeqLayout) else SeqLayout(layout or sp_layout(axis_name, local_len), n, local_len * n)
    idx = lay.global_index(r)
    return idx if device is None else idx.to(device)


def sp_shard(t, dim=1, axis_name="sp", layout=None):
    """This rank's rows of a FULL-length tensor along `dim` (tokens, targets, loss masks, vision masks): the
    boundary permutation of the ownership rule in force (or of `layout`, see sp_positions).  The attention masks
    (attention_mask / segment_ids) are NOT sharded: every rank keeps them full length (lwm/llama.py:563-564)."""
    n, _ = sp_size_rank(axis_name)
    if n == 1:
        return t
    S = t.shape[dim]
    if S % n:
        raise ValueError(f"length {S} is not divisible by the sp axis ({n})")
    return t.index_select(dim, sp_positions(S // n, axis_name, t.device, layout))


def sp_all_reduce_sum(t, axis_name="sp"):
    """Sum of `t` over the ranks of the "sp" axis, in place (no-op for one rank): what makes a per-sequence statistic
    of a sequence-sharded batch -- the count of valid targets of a row, a loss term -- the whole sequence's.  f32 / f64
    / integer tensors; a gloo group moves host memory, so a device tensor is staged through the host there."""
    import torch.distributed as dist
    n, _ = sp_size_rank(axis_name)
    if n == 1:
        return t
    g = _resolve_axis(axis_name)
    if t.is_cuda and dist.get_backend(g) != "nccl":
        h = t.detach().cpu()
        dist.all_reduce(h, group=g)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=g)
    return t


def key_valid_from_bias(attn_bias):
    """The additive key-padding bias of lwm/llama.py:527-537 -- (B,1,1,S_global), 0 where attention_mask > 0 and
    finfo(dtype).min elsewhere (-3.4028e38 at fp32, -3.3895e38 at bf16) -- back to the key mask it was made from, which is
    what the kernels evaluate: (B, S_global) uint8."""
    if attn_bias.dim() != 4 or attn_bias.shape[1] != 1 or attn_bias.shape[2] != 1:
        raise ValueError("attn_bias must be (B,1,1,S_global) as built at lwm/llama.py:527-537")
    return (attn_bias[:, 0, 0, :].float() > -1e30).to(torch.uint8).contiguous()


def ringattention(q, k, v, attn_bias, segment_ids, axis_name="sp", float32_logits=True,
                  cache_idx=None, blockwise_kwargs=None, layout=None):
    """q,k,v: local (B, S/sp, H, D) bf16 shards.  attn_bias: (B,1,1,S_global)
    additive key-padding bias {0, finfo.min} or None (lwm/llama.py:533-537);
    segment_ids: (B, S_global) int or None -- both replicated on every rank
    (lwm/llama.py:563-564).  Returns out with q's shape/dtype.
    `layout` (extension): which positions the local rows are -- None = the rule bound by set_sp_group
    (sp_layout(): zigzag for more than one rank)."""
    kw = dict(blockwise_kwargs or {})
    if cache_idx is not None:
        raise NotImplementedError("cache_idx is None at every reference call site (lwm/llama.py:544)")
    if float(kw.get("attn_pdrop", 0.0)) != 0.0 and not kw.get("deterministic", True):
        raise NotImplementedError("attention dropout: attn_pdrop=0.0 in the reference config "
                                  "(lwm/llama.py:151)")
    cbs = kw.get("causal_block_size", 1)
    if cbs not in (None, 1):
        raise NotImplementedError("causal_block_size must be 1 (lwm/llama.py:546) or None")
    if not float32_logits:
        # logits are always f32 here; float32_logits=False would only lower precision
        pass
    key_valid = None if attn_bias is None else key_valid_from_bias(attn_bias)
    if layout is None:
        layout = sp_layout(axis_name, q.shape[1])
    return ring_attention(q, k, v, group=_resolve_axis(axis_name), causal=cbs == 1,
                          segment_ids=segment_ids, key_valid=key_valid, layout=layout)


def _comm(group):
    import torch.distributed as dist
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return SingleComm()
    if group is None and dist.get_world_size() == 1:
        return SingleComm()
    if group is None:   # unbound "sp" axis: _resolve_axis has raised already for axis names
        raise RuntimeError("no sequence-parallel group: call set_sp_group first")
    from .ring import _torch_comm
    return _torch_comm(group)       # one communicator (and exchange-buffer pool) per process group


def ringattention_inference(q, k, v, attn_mask, axis_name="sp", q_sharded=None, block_ops=None, comm=None, *,
                            causal_offset=None, key_valid=None):
    """Dense-mask ring attention for S <= chunk size and for cached decoding
    (call site lwm/llama.py:599-614).  q: (B,Q,H,D); k, v: this rank's (B,K/sp,H,D)
    shard (the KV cache when decoding); attn_mask: boolean (B,1,Q,K_global) built at
    lwm/llama.py:577-592.  As in the reference, q is replicated over "sp" when
    Q == 1 and sharded otherwise (`q_sp_dim`, :599) unless `q_sharded` says so.

    Extension (keyword-only, single rank): with a KV cache present the mask of :577-592 is always
    `key <= cache_index + query  AND  attention_mask[key]` (segment_mask is None, :582).  Passing that
    structure -- causal_offset = cache_index, key_valid = attention_mask -- instead of the dense
    (B,1,Q,K) tensor runs the causal kernel with q_start = cache_index: key tiles beyond the diagonal are
    never read and no Q x K mask is materialised (prefill of a long prompt into a longer cache)."""
    cm = comm if comm is not None else _comm(_resolve_axis(axis_name))
    if causal_offset is not None and cm.size == 1:
        kv = None if key_valid is None else (key_valid != 0).to(torch.uint8).contiguous()
        out, _ = (block_ops or HipBlockOps).fwd(q, k, v, q_start=int(causal_offset), k_start=0, causal=True,
                                               key_valid=kv)
        return out
    if attn_mask is None:
        raise ValueError("attn_mask is required (the structured form is single-rank only)")
    if attn_mask.dim() != 4 or attn_mask.shape[1] != 1:
        raise ValueError("attn_mask must be (B,1,Q,K) as built at lwm/llama.py:577-592")
    if q_sharded is None:
        q_sharded = q.shape[1] != 1
    mask = (attn_mask[:, 0] != 0).to(torch.uint8).contiguous()
    return ring_inference(block_ops or HipBlockOps, cm, q, k, v, mask, q_sharded=q_sharded)


def concatenate_to_cache(cached_key, cached_value, key, value, cache_index, axis_name="sp",
                         new_sharded=None, block_ops=None, comm=None):
    """FlaxLLaMAAttention._concatenate_to_cache (lwm/llama.py:440-492) on this rank's
    (B, max_length/sp, H, D) cache shards, in place.  Returns the new cache_index."""
    if new_sharded is None:
        new_sharded = key.shape[1] != 1            # decode keys are replicated (:468-471)
    cm = comm if comm is not None else _comm(_resolve_axis(axis_name))
    if cm.size == 1:
        new_sharded = False
    return cache_update(block_ops or HipBlockOps, cm, cached_key, cached_value, key, value, int(cache_index),
                        new_sharded=new_sharded)


def blockwise_feedforward(module, x, chunk_size, pre_remat=True):
    """lwm/llama.py:729-734: apply a position-wise module over sequence chunks (identical result to
    module(x)).  `pre_remat` (the reference wraps the MLP in remat with policy nothing_saveable,
    lwm/llama.py:673-678): each chunk's activations inside `module` are recomputed in the backward
    instead of stored -- at 1M tokens the two (S, 11008) SwiGLU intermediates are 44 GB per layer."""
    S = x.shape[1]

    def run(c):
        if pre_remat and torch.is_grad_enabled() and c.requires_grad:
            from torch.utils.checkpoint import checkpoint
            return checkpoint(module, c, use_reentrant=False)
        return module(c)

    if chunk_size is None or S <= chunk_size:
        return run(x)
    if S % chunk_size:
        raise ValueError(f"sequence length {S} is not a multiple of chunk_size {chunk_size}")
    return torch.cat([run(c) for c in x.split(chunk_size, dim=1)], dim=1)
