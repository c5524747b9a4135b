"""The HIP-kernel harness against HF transformers' LlamaForCausalLM (golden vectors generated on
CPU in float32 by tests/golden/gen_hf_llama_golden.py; the implementation the reference points
PyTorch users at, scripts/sample_pyt.py:8).  Weights go through lwm_amd.weights (layout
transposes + rotate_half -> interleaved q/k re-ordering), activations are bf16 here and f32
there.  Tolerance: logits max|d| <= 3e-2 * max|ref| and cosine >= 0.9995; loss 5e-3 relative;
gradient cosine >= 0.995."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _model():
    import torch
    import hf_fixture as F
    from lwm_amd import weights as W
    from lwm_amd.llama import LLaMAForCausalLM
    cfg = W.config_from_hf(F.HF_CONFIG)
    model = LLaMAForCausalLM(cfg).cuda()
    W.load_params(model, W.hf_to_lwm(F.state_dict(), cfg.num_attention_heads))
    return F, cfg, model


def test_harness_logits_match_hf_transformers():
    import torch
    F, cfg, model = _model()
    gold = np.load(os.path.join(HERE, "golden", "hf_llama_tiny.npz"))
    ids = F.token_ids().cuda()
    with torch.no_grad():
        h = model.hidden_states(ids[:, :-1])
        logits = (h.float() @ model.lm_head.float()).cpu()
    ref = torch.from_numpy(gold["logits"])
    assert (logits - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(logits.flatten().double(), ref.flatten().double(), dim=0).item()
    assert cos >= 0.9995, cos


def test_harness_loss_and_gradients_match_hf_transformers():
    import torch
    from lwm_amd.llama import hf_rotary_to_interleaved
    F, cfg, model = _model()
    gold = np.load(os.path.join(HERE, "golden", "hf_llama_tiny.npz"))
    ids = F.token_ids().cuda()
    loss, acc = model.loss(ids[:, :-1].contiguous(), ids[:, 1:].contiguous(), chunk=64)
    loss.backward()
    assert abs(loss.item() - float(gold["loss"])) <= 5e-3 * float(gold["loss"])
    p = dict(model.named_parameters())
    for hf_name, ours, kind in (("grad_q_proj_0", "h.0.attention.wq", "rotary"),
                                ("grad_k_proj_1", "h.1.attention.wk", "rotary"),
                                ("grad_v_proj_0", "h.0.attention.wv", "linear")):
        g = torch.from_numpy(gold[hf_name])
        g = (hf_rotary_to_interleaved(g, cfg.num_attention_heads) if kind == "rotary" else g.t()).flatten().double()
        got = p[ours].grad.float().cpu().flatten().double()
        cos = float(got @ g / (got.norm() * g.norm()))
        assert cos >= 0.995, (hf_name, cos)
        assert abs(float(got.norm() / g.norm()) - 1) <= 3e-2, hf_name
