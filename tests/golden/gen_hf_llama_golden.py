"""Golden vectors from HF transformers' LlamaForCausalLM -- the implementation the reference itself
points PyTorch users at (README.md:74, scripts/sample_pyt.py:8) -- run HERE on CPU in float32 with
eager attention.  The reference's JAX path cannot be imported in this image; this is the one
executable implementation of the same model that the reference names, so these vectors anchor
RoPE convention + q/k re-ordering, RMSNorm, causal attention, SwiGLU, the loss and the backward
against code that is not ours.

    python tests/golden/gen_hf_llama_golden.py      # needs `transformers`; writes hf_llama_tiny.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import hf_fixture as F  # noqa: E402


def main():
    import transformers
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(**F.HF_CONFIG)
    cfg._attn_implementation = "eager"
    model = LlamaForCausalLM(cfg).float()
    missing, unexpected = model.load_state_dict(F.state_dict(), strict=False)
    assert not [m for m in missing if "rotary" not in m] and not unexpected, (missing, unexpected)
    ids = F.token_ids()
    logits = model(input_ids=ids[:, :-1]).logits.float()
    logp = torch.log_softmax(logits, -1)
    loss = -torch.gather(logp, -1, ids[:, 1:, None])[..., 0].mean()          # lwm/train.py:171-181
    loss.backward()
    acc = (logits.argmax(-1) == ids[:, 1:]).float().mean()
    # greedy generation through HF's KV cache from a left-padded prompt (positions = cumsum(mask) - 1)
    PL, NEW = 43, 12       # the prompt length whose greedy top-2 margins are all > 0.1
    gmask = torch.ones(1, PL, dtype=torch.long)
    gmask[:, :5] = 0
    with torch.no_grad():
        gen = model.generate(input_ids=ids[:, :PL], attention_mask=gmask, max_new_tokens=NEW, do_sample=False,
                             output_scores=True, return_dict_in_generate=True, pad_token_id=0)
    gen_scores = torch.stack(gen.scores, 1).float()          # (1, NEW, vocab): raw logits under greedy
    p = dict(model.named_parameters())
    np.savez_compressed(
        os.path.join(HERE, "hf_llama_tiny.npz"),
        logits=logits.detach().numpy(), loss=np.float32(loss.item()), accuracy=np.float32(acc.item()),
        grad_q_proj_0=p["model.layers.0.self_attn.q_proj.weight"].grad.numpy(),
        grad_k_proj_1=p["model.layers.1.self_attn.k_proj.weight"].grad.numpy(),
        grad_v_proj_0=p["model.layers.0.self_attn.v_proj.weight"].grad.numpy(),
        gen_mask=gmask.numpy().astype(np.int32), gen_tokens=gen.sequences.numpy().astype(np.int64),
        gen_scores=gen_scores.numpy(),
        transformers_version=np.array(transformers.__version__), torch_version=np.array(torch.__version__))
    print("loss", loss.item(), "acc", acc.item(), "logits |max|", logits.abs().max().item())


if __name__ == "__main__":
    main()
