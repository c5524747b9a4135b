"""Needle retrieval THROUGH THE MODEL at 32K / 128K / 1M tokens: the 2-layer
harness (RMSNorm, RoPE with the reference's long-context theta, RingAttention forward, lm_head) with
the hand-set induction circuit of tests/_induction.py -- verified on the CPU oracle model in
tests/test_induction_oracle.py -- must name the token that followed the single earlier occurrence of
the final token, at several depths.  A previous-token head that must resolve j = i-1 among up to 2^20
positions and a content match that must survive RoPE over the full distance exercise exactly what a
long-context kernel can get wrong: position offsets, tile boundaries, online-softmax rescaling."""
import pytest

pytestmark = pytest.mark.gpu


def _model(theta, S):
    import torch
    from lwm_amd.llama import LLaMAConfig, LLaMAForCausalLM
    from lwm_amd.weights import load_params
    from tests import _induction as I
    cfg_kw, st = I.build(theta, S)
    cfg = LLaMAConfig(**cfg_kw, scan_mlp_chunk_size=65536)
    with torch.device("cuda"):
        model = LLaMAForCausalLM(cfg)
    return I, load_params(model, st)


@pytest.mark.parametrize("theta,S,depths", [(1e7, 32768, (0.0, 0.5, 0.999)), (1e7, 131072, (0.02, 0.71)),
                                            (5e7, 1 << 20, (0.35,))])
def test_harness_retrieves_the_needle(theta, S, depths):
    import torch
    I, model = _model(theta, S)
    for depth in depths:
        toks, pos = I.haystack(S, depth)
        with torch.no_grad():
            h = model.hidden_states(toks.cuda())
            logits = (h[0, -1].float() @ model.lm_head.float()).cpu()
        top = logits.topk(2)
        assert top.indices[0].item() == I.VALUE_TOKEN, (S, depth, pos, top)
        assert (top.values[0] - top.values[1]).item() > 10, (S, depth, top)
        del h
    # the needle removed: the answer must not appear
    toks, pos = I.haystack(S, depths[0])
    toks[0, pos], toks[0, pos + 1] = 5, 6
    with torch.no_grad():
        h = model.hidden_states(toks.cuda())
    assert (h[0, -1].float() @ model.lm_head.float()).argmax().item() != I.VALUE_TOKEN
