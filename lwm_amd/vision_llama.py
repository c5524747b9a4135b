"""Vision-language variant of the model HARNESS (lwm/vision_llama.py): the transformer is the one of
lwm_amd/llama.py -- same blocks, same RingAttention hot path -- plus a second embedding table for the
VQGAN code vocabulary (`vte`, 8192 codes + 256 special tokens, lwm/vision_llama.py:30-32, :264-270), a
second output head (`vision_head`, :354-360), the per-position choice between the two embeddings
(:307-311) and the 0.5 * (vision CE + text CE) objective of lwm/train.py:183-202.  BASELINE config #4
(VQGAN-tokenised video frames in a 256K context) runs through this module."""
from __future__ import annotations

import torch

from .llama import LLaMAConfig, LLaMAForCausalLM, _dense
from .llama_ops import chunked_lm_head_loss


class VideoLLaMAConfig(LLaMAConfig):
    """lwm/vision_llama.py:29-36."""

    def __init__(self, vision_vocab_size=8448, tie_vision_embeddings=False, sample_mode="all", **kwargs):
        super().__init__(**kwargs)
        self.vision_vocab_size = vision_vocab_size      # 8192 + 256
        self.tie_vision_embeddings = tie_vision_embeddings
        self.sample_mode = sample_mode


class VideoLLaMAForCausalLM(LLaMAForCausalLM):
    def __init__(self, cfg: VideoLLaMAConfig, dtype=torch.bfloat16):
        super().__init__(cfg, dtype)
        std = cfg.initializer_range
        self.vte = torch.nn.Parameter(torch.randn(cfg.vision_vocab_size, cfg.hidden_size).mul_(std).to(dtype))
        if not cfg.tie_vision_embeddings:
            self.vision_head = _dense(cfg.hidden_size, cfg.vision_vocab_size, std, dtype)

    def _embed(self, input_ids, vision_masks):
        """lwm/vision_llama.py:298-311: one-token steps pick the table by sample_mode; otherwise text
        positions read wte, vision positions read vte (ids are zeroed for the table they do not use)."""
        ids = input_ids.long()
        if ids.shape[1] == 1 and self.cfg.sample_mode in ("text", "vision"):
            return torch.nn.functional.embedding(ids, self.wte if self.cfg.sample_mode == "text" else self.vte)
        if ids.shape[1] == 1:
            raise NotImplementedError("sample_mode='all' cannot decode one token at a time (lwm/vision_llama.py:303)")
        vm = vision_masks.to(torch.bool)
        zero = torch.zeros_like(ids)
        text = torch.nn.functional.embedding(torch.where(vm, zero, ids), self.wte)
        vis = torch.nn.functional.embedding(torch.where(vm, ids, zero), self.vte)
        return torch.where(vm[..., None], vis, text)

    def hidden_states(self, input_ids, vision_masks, attention_mask=None, segment_ids=None, position_ids=None,
                      cache=None):
        x = self._embed(input_ids, vision_masks)
        fc = self._table(x.device)
        for i, blk in enumerate(self.h):
            x = blk(x, fc, attention_mask, segment_ids, position_ids, None if cache is None else cache[i])
        return self.ln_f(x)

    def _vision_kernel(self):
        return self.vte.t() if self.cfg.tie_vision_embeddings else self.vision_head

    def loss(self, input_tokens, input_vision_masks, target_tokens, target_vision_masks, loss_masks=None,
             attention_mask=None, segment_ids=None, position_ids=None, chunk=8192):
        """modality 'vision,text' of lwm/train.py:183-209 -> (loss, metrics); both heads go through the
        chunked head+loss operator so that no (S, vocab) logits tensor exists at 256K-1M tokens."""
        h = self.hidden_states(input_tokens, input_vision_masks, attention_mask, segment_ids, position_ids)
        tvm = target_vision_masks.to(torch.bool)
        lm = torch.ones_like(target_tokens, dtype=torch.float32) if loss_masks is None else loss_masks.float()
        zero = torch.zeros_like(target_tokens)
        v_loss, v_acc = chunked_lm_head_loss(h, self._vision_kernel().contiguous(), torch.where(tvm, target_tokens, zero),
                                             lm * tvm.float(), chunk)
        t_loss, t_acc = chunked_lm_head_loss(h, self.lm_head, torch.where(tvm, zero, target_tokens),
                                             lm * (~tvm).float(), chunk)
        return 0.5 * (v_loss + t_loss), dict(vision_loss=v_loss, vision_acc=v_acc, text_loss=t_loss, text_acc=t_acc)
