"""CPU float32 reference of the LLaMA harness (lwm/llama.py restated with plain PyTorch-CPU
ops: dense masked attention, interleaved RoPE, RMSNorm, SwiGLU, tux cross entropy) --
TEST INFRASTRUCTURE, NOT PRODUCT.  BASELINE config #1 (2-layer slice, S = 4096, fp32 on CPU).
The softmax-attention arithmetic is UNPINNED (see oracle/attention_ref.py: it lives in the absent `ringattention`
package).  Everything around it is PINNED (round 5) to a run of the reference's own lines: one whole layer --
FlaxLLaMABlock.__call__, FlaxLLaMAAttention.__call__ (both branches), FlaxLLaMAMLP.__call__, RMSNorm, RoPE -- executed out
of /root/reference with numpy standing in for jax.numpy (tests/golden/gen_ref_run_golden.py, ref_run.npz); forward_hidden()
below reproduces it to 4e-7 (tests/test_golden.py::test_oracle_model_layer_reproduces_the_reference_run), as it
reproduces HF transformers' LlamaForCausalLM (tests/test_weights.py)."""
import math

import numpy as np
import torch


def _rmsnorm(x, w, eps):
    x32 = x.float()
    return (x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)) * w.float()


def _rope(x, fc, pos):
    B, S, H, D = x.shape
    xr = x.float().reshape(B, S, H, D // 2, 2)
    c, s = fc[pos][..., 0][:, :, None, :], fc[pos][..., 1][:, :, None, :]
    return torch.stack((xr[..., 0] * c - xr[..., 1] * s, xr[..., 0] * s + xr[..., 1] * c), dim=-1).reshape(B, S, H, D)


def forward_loss(state, cfg, input_tokens, target_tokens, loss_masks=None, attention_mask=None,
                 segment_ids=None):
    """state: dict name -> float32 CPU tensor (requires_grad as wanted), names as lwm_amd.llama."""
    return _loss(forward_logits(state, cfg, input_tokens, attention_mask, segment_ids), target_tokens, loss_masks)


def forward_logits(state, cfg, input_tokens, attention_mask=None, segment_ids=None):
    """(B, S, vocab) float32 logits (lwm/llama.py:982-1106).  Pinned against HF transformers'
    LlamaForCausalLM by tests/test_weights.py (tests/golden/hf_llama_tiny.npz)."""
    return forward_hidden(state, cfg, input_tokens, attention_mask, segment_ids) @ state["lm_head"]


def vision_text_loss(state, cfg, input_tokens, input_vision_masks, target_tokens, target_vision_masks,
                     loss_masks=None, attention_mask=None, segment_ids=None):
    """lwm/vision_llama.py:307-311 (mixed embedding), :411-415 (two heads), lwm/train.py:183-202."""
    vm = input_vision_masks.bool()
    ids = input_tokens.long()
    emb = torch.where(vm[..., None], state["vte"][torch.where(vm, ids, 0)], state["wte"][torch.where(vm, 0, ids)])
    h = forward_hidden(state, cfg, input_tokens, attention_mask, segment_ids, input_embeds=emb)
    tvm = target_vision_masks.bool()
    lm = torch.ones(input_tokens.shape) if loss_masks is None else loss_masks.float()
    v_loss, v_acc = _loss(h @ state["vision_head"], torch.where(tvm, target_tokens, 0), lm * tvm.float())
    t_loss, t_acc = _loss(h @ state["lm_head"], torch.where(tvm, 0, target_tokens), lm * (~tvm).float())
    return 0.5 * (v_loss + t_loss), dict(vision_loss=v_loss, vision_acc=v_acc, text_loss=t_loss, text_acc=t_acc)


def forward_hidden(state, cfg, input_tokens, attention_mask=None, segment_ids=None, input_embeds=None):
    """(B, S, d) float32 final hidden states (after ln_f)."""
    B, S = input_tokens.shape
    H = cfg.num_attention_heads
    D = cfg.hidden_size // H
    freqs = 1.0 / (cfg.theta ** (np.arange(0, D, 2)[: D // 2].astype(np.float32) / D))
    ang = np.outer(np.arange(cfg.max_sequence_length), freqs).astype(np.float32)
    fc = torch.from_numpy(np.stack((np.cos(ang), np.sin(ang)), -1))
    pos = torch.arange(S)[None].expand(B, S)
    x = state["wte"][input_tokens.long()] if input_embeds is None else input_embeds
    vis = torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None]
    if segment_ids is not None:
        vis = vis & (segment_ids[:, None, :, None] == segment_ids[:, None, None, :])
    if attention_mask is not None:
        vis = vis & (attention_mask[:, None, None, :] > 0)
    for i in range(cfg.num_hidden_layers):
        p = f"h.{i}."
        hn = _rmsnorm(x, state[p + "attention_norm.kernel"], cfg.rms_norm_eps)
        q = (hn @ state[p + "attention.wq"]).reshape(B, S, H, D)
        k = (hn @ state[p + "attention.wk"]).reshape(B, S, H, D)
        v = (hn @ state[p + "attention.wv"]).reshape(B, S, H, D)
        q, k = _rope(q, fc, pos), _rope(k, fc, pos)
        s = torch.einsum("bqhd,bkhd->bhqk", q, k) / math.sqrt(D)
        s = s.masked_fill(~vis, float("-inf"))
        # a query with no visible key (left padding) gives 0, as in oracle/attention_ref.py and in the kernels -- not NaN,
        # which a second layer would spread over the whole batch row through 0 * NaN (the reference returns an average of
        # masked keys there and never uses it, lwm/vision_chat.py:138-140)
        pr = torch.where(vis.any(-1, keepdim=True), torch.softmax(s, dim=-1), torch.zeros(()))
        a = torch.einsum("bhqk,bkhd->bqhd", pr, v).reshape(B, S, H * D)
        x = x + a @ state[p + "attention.wo"]
        hn = _rmsnorm(x, state[p + "ffn_norm.kernel"], cfg.rms_norm_eps)
        ff = (torch.nn.functional.silu(hn @ state[p + "feed_forward.w1"]) * (hn @ state[p + "feed_forward.w3"])) \
            @ state[p + "feed_forward.w2"]
        x = x + ff
    return _rmsnorm(x, state["ln_f.kernel"], cfg.rms_norm_eps)


def _loss(logits, target_tokens, loss_masks=None):
    """tux.cross_entropy_loss_and_accuracy as called at lwm/train.py:177-181."""
    B, S = target_tokens.shape
    valid = torch.ones(B, S) if loss_masks is None else loss_masks.float()
    logp = torch.log_softmax(logits.float(), dim=-1)
    tok_lp = torch.gather(logp, -1, target_tokens.long()[..., None])[..., 0]
    tok_lp = torch.where(valid > 0, tok_lp, torch.zeros_like(tok_lp))
    length = valid.sum(-1).clamp_min(1e-10)
    loss = -(tok_lp.sum(-1) / length).mean()
    correct = torch.where(valid > 0, logits.argmax(-1) == target_tokens.long(), torch.zeros_like(valid, dtype=torch.bool))
    acc = (correct.float().sum(-1) / length).mean()
    return loss, acc
