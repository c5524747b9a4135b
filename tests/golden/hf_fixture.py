"""Shared by gen_hf_llama_golden.py (runs HF transformers) and the parity tests (which do not
need transformers): the tiny LLaMA whose weights are a pure function of a seed."""
import torch

HF_CONFIG = dict(vocab_size=384, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                 num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=512,
                 rms_norm_eps=1e-6, rope_theta=10000.0, initializer_range=0.02, hidden_act="silu",
                 tie_word_embeddings=False, attention_bias=False, mlp_bias=False)
SEQ = 193          # 192 predicted positions: not a multiple of the kernels' 64/256-row tiles


def state_dict(seed=20240917):
    """HF-named float32 tensors whose values are exactly representable in bf16."""
    g = torch.Generator().manual_seed(seed)
    c = HF_CONFIG
    d, f, v = c["hidden_size"], c["intermediate_size"], c["vocab_size"]
    rnd = lambda *s, std: (torch.randn(*s, generator=g) * std).to(torch.bfloat16).float()
    sd = {"model.embed_tokens.weight": rnd(v, d, std=1.0)}
    for i in range(c["num_hidden_layers"]):
        p = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"self_attn.{n}.weight"] = rnd(d, d, std=0.09)      # wide enough for peaked softmaxes
        sd[p + "mlp.gate_proj.weight"] = rnd(f, d, std=0.06)
        sd[p + "mlp.up_proj.weight"] = rnd(f, d, std=0.06)
        sd[p + "mlp.down_proj.weight"] = rnd(d, f, std=0.06)
        sd[p + "input_layernorm.weight"] = (1 + rnd(d, std=0.1))
        sd[p + "post_attention_layernorm.weight"] = (1 + rnd(d, std=0.1))
    sd["model.norm.weight"] = (1 + rnd(d, std=0.1))
    sd["lm_head.weight"] = rnd(v, d, std=0.06)
    return {k: t.to(torch.bfloat16).float() for k, t in sd.items()}


def token_ids(seed=7):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, HF_CONFIG["vocab_size"], (1, SEQ), generator=g)
