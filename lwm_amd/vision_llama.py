"""Vision-language variant of the model HARNESS (lwm/vision_llama.py): the transformer is the one of
lwm_amd/llama.py -- same blocks, same RingAttention hot path -- plus a second embedding table for the
VQGAN code vocabulary (`vte`, 8192 codes + 256 special tokens, lwm/vision_llama.py:30-32, :264-270), a
second output head (`vision_head`, :354-360), the per-position choice between the two embeddings
(:307-311) and the 0.5 * (vision CE + text CE) objective of lwm/train.py:183-202.  BASELINE config #4
(VQGAN-tokenised video frames in a 256K context) runs through this module."""
from __future__ import annotations

import torch

from .llama import LLaMAConfig, LLaMAForCausalLM, _dense
from .llama_ops import chunked_lm_head_loss, dense


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
                      cache=None, layout=None):
        _, position_ids = self._ring_position_ids(input_ids, position_ids, cache, layout)
        x = self._embed(input_ids, vision_masks)
        fc = self._table(x.device)
        for i, blk in enumerate(self.h):
            x = blk(x, fc, attention_mask, segment_ids, position_ids, None if cache is None else cache[i], layout)
        return self.ln_f(x)

    def _vision_kernel(self):
        return self.vte.t() if self.cfg.tie_vision_embeddings else self.vision_head

    def loss(self, input_tokens, input_vision_masks, target_tokens, target_vision_masks, loss_masks=None,
             attention_mask=None, segment_ids=None, position_ids=None, chunk=8192, layout=None, sp_sharded=True):
        """modality 'vision,text' of lwm/train.py:183-209 -> (loss, metrics); both heads go through the
        chunked head+loss operator so that no (S, vocab) logits tensor exists at 256K-1M tokens."""
        h = self.hidden_states(input_tokens, input_vision_masks, attention_mask, segment_ids, position_ids, layout=layout)
        tvm = target_vision_masks.to(torch.bool)
        lm = torch.ones_like(target_tokens, dtype=torch.float32) if loss_masks is None else loss_masks.float()
        zero = torch.zeros_like(target_tokens)
        v_loss, v_acc = chunked_lm_head_loss(h, self._vision_kernel().contiguous(), torch.where(tvm, target_tokens, zero),
                                             lm * tvm.float(), chunk, sp_sharded=sp_sharded)
        t_loss, t_acc = chunked_lm_head_loss(h, self.lm_head, torch.where(tvm, zero, target_tokens),
                                             lm * (~tvm).float(), chunk, sp_sharded=sp_sharded)
        return 0.5 * (v_loss + t_loss), dict(vision_loss=v_loss, vision_acc=v_acc, text_loss=t_loss, text_acc=t_acc)

    # ---- generation (lwm/vision_llama.py:447-745).  Eager, through the KV cache of lwm_amd/llama.py.
    def _prefill(self, input_ids, vision_masks, attention_mask, max_length):
        B, S = input_ids.shape
        dev = input_ids.device
        cache = self.init_cache(B, max_length, dev)
        ext = torch.ones(B, max_length, dtype=torch.int32, device=dev)
        if attention_mask is not None:
            pos = attention_mask.to(torch.int32).cumsum(-1) - 1          # prepare_inputs_for_generation (:447-466)
            ext[:, :S] = attention_mask.to(torch.int32)
        else:
            pos = torch.arange(S, dtype=torch.int32, device=dev)[None].expand(B, S)
        pos = pos.clamp_min(0).to(torch.int32).contiguous()
        vm = torch.zeros_like(input_ids, dtype=torch.bool) if vision_masks is None else vision_masks.to(torch.bool)
        h = self.hidden_states(input_ids, vm, ext, None, pos, cache)
        return h[:, -1], cache, ext, (pos[:, -1:] + 1).contiguous()

    def _step(self, tok, cache, ext, pos):
        h = self.hidden_states(tok, None, ext, None, pos, cache)
        return h[:, -1], (pos + 1).contiguous()

    @staticmethod
    def _pick(logits, temperature, top_k, do_sample, gen):
        """FlaxTemperatureLogitsWarper / FlaxTopKLogitsWarper + categorical sampling (greedy when
        do_sample is False or temperature == 0)."""
        logits = logits.float()
        if not do_sample or temperature == 0:
            return logits.argmax(-1, keepdim=True)
        logits = logits / float(temperature)
        if top_k and 0 < top_k < logits.shape[-1]:
            kth = logits.topk(int(top_k), dim=-1).values[..., -1:]
            logits = logits.masked_fill(logits < kth, float("-inf"))
        return torch.multinomial(torch.softmax(logits, -1), 1, generator=gen)

    @torch.no_grad()
    def generate(self, input_ids, vision_masks=None, attention_mask=None, max_new_tokens=16, max_length=None,
                 temperature=1.0, top_k=None, do_sample=False, eos_token_id=None, pad_token_id=0, generator=None):
        """Text continuation of a (left-padded) vision-language prompt -- what lwm/vision_chat.py:205-227
        runs with sample_mode='text': prefill over both embedding tables, then one token at a time
        through the text head.  Returns the NEW tokens (B, max_new_tokens), pad after eos."""
        if self.cfg.sample_mode != "text":
            raise ValueError("generate() decodes text: set sample_mode='text' (scripts/run_vision_chat.sh)")
        B, S = input_ids.shape
        max_length = max_length or (S + max_new_tokens)
        h, cache, ext, pos = self._prefill(input_ids, vision_masks, attention_mask, max_length)
        head = self.lm_head        # (f32 logits from the bf16 kernel: llama_ops.dense)
        out = torch.full((B, max_new_tokens), int(pad_token_id), dtype=input_ids.dtype, device=input_ids.device)
        done = torch.zeros(B, dtype=torch.bool, device=input_ids.device)
        for i in range(max_new_tokens):
            tok = self._pick(dense(h, head, torch.float32), temperature, top_k, do_sample, generator).to(input_ids.dtype)
            out[:, i] = torch.where(done, out[:, i], tok[:, 0])
            if eos_token_id is not None:
                done |= tok[:, 0] == eos_token_id
                if bool(done.all()):
                    break
            if i + 1 < max_new_tokens:
                h, pos = self._step(tok, cache, ext, pos)
        return out

    @torch.no_grad()
    def generate_vision(self, input_ids, cfg_scales, attention_mask=None, vision_masks=None, max_new_tokens=257,
                        temperature=1.0, top_k=None, generator=None, max_length=None):
        """FlaxVideoLLaMAForCausalLM.generate_vision / _sample_vision (lwm/vision_llama.py:476-745):
        the batch holds the conditional prompts followed by the same number of unconditional ones;
        logits = uncond + cfg * (cond - uncond) over the VISION head (sample_mode='vision'), top-k /
        temperature sampling, every 257th new token forced to the end-of-frame code 8192 (:549-552),
        the chosen token fed to both halves.  Returns the new tokens of the conditional half."""
        if self.cfg.sample_mode != "vision":
            raise ValueError("generate_vision() needs sample_mode='vision' (scripts/run_sample_image.sh)")
        B2, S = input_ids.shape
        if B2 % 2:
            raise ValueError("generate_vision: batch = conditional prompts + as many unconditional ones")
        B = B2 // 2
        cfg = torch.as_tensor(cfg_scales, dtype=torch.float32, device=input_ids.device).reshape(-1, 1).expand(B, 1)
        max_length = max_length or (S + max_new_tokens)
        h, cache, ext, pos = self._prefill(input_ids, vision_masks, attention_mask, max_length)
        head = self._vision_kernel().contiguous()
        out = torch.empty((B, max_new_tokens), dtype=input_ids.dtype, device=input_ids.device)
        for i in range(max_new_tokens):
            logits = dense(h, head, torch.float32)
            cond, uncond = logits[:B], logits[B:]
            tok = self._pick(uncond + cfg * (cond - uncond), temperature, top_k, True, generator).to(input_ids.dtype)
            if (i + 1) % 257 == 0:
                tok = torch.full_like(tok, 8192)
            out[:, i] = tok[:, 0]
            if i + 1 < max_new_tokens:
                h, pos = self._step(torch.cat([tok, tok], 0), cache, ext, pos)
        return out
