"""The hand-set induction circuit (tests/_induction.py) on the CPU float32 oracle model: the
construction itself is checked here (retrieval at every depth, sharp heads) so that the GPU test
-- the same weights through the HIP harness at 32K..1M tokens -- tests the kernels, not the
construction."""
import pytest
import torch

from lwm_amd.llama import LLaMAConfig
from oracle import llama_model_ref as M
from tests import _induction as I


@pytest.mark.parametrize("theta,S,design", [(1e7, 768, 131072), (5e7, 1024, 1 << 20)])
def test_oracle_model_retrieves_the_needle(theta, S, design):
    cfg_kw, st = I.build(theta, design)             # gains sized for the context the GPU test runs
    cfg = LLaMAConfig(**dict(cfg_kw, max_sequence_length=S))
    for depth in (0.0, 0.3, 0.97):
        toks, pos = I.haystack(S, depth)
        logits = M.forward_logits(st, cfg, toks)[0, -1]
        assert logits.argmax().item() == I.VALUE_TOKEN, (depth, logits.topk(3))
        top2 = logits.topk(2).values
        assert (top2[0] - top2[1]).item() > 20         # 64 on a clean copy vs <= ~30 for any other code
    # without the needle pair the answer must not appear by accident
    toks, pos = I.haystack(S, 0.5)
    toks[0, pos], toks[0, pos + 1] = 5, 6
    assert M.forward_logits(st, cfg, toks)[0, -1].argmax().item() != I.VALUE_TOKEN
