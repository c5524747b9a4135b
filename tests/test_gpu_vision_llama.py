"""BASELINE config #4 in miniature: synthetic video frames -> VQGAN.encode (HIP) -> code indices packed the
way lwm/data.py:205-219 packs them (<vision> frame codes eof ... eov </vision>, vision_mask over codes and
eof/eov) -> the vision-language harness (lwm/vision_llama.py: vte / wte choice, two heads,
0.5*(vision CE + text CE), lwm/train.py:183-202) fwd+bwd on the HIP operators, against the float32 CPU
oracle model on the same tokens.  Tolerance: bf16 vs fp32 -- loss 1e-2 relative, gradient cosine >= 0.99."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EOF_TOKEN, EOV_TOKEN = 8192, 8193          # lwm/data.py DatasetFactory defaults (eof_token / eov_token)


def _tokens_from_frames(n_frames, text_vocab, seed):
    import torch
    from lwm_amd.vqgan import VQGAN, VQGANConfig, random_params
    vcfg = VQGANConfig.get_default_config(dict(resolution=32, channel_mult=(1, 2), num_embeddings=8192))
    vq = VQGAN(params=random_params(vcfg, seed=seed), config=vcfg)
    g = torch.Generator().manual_seed(seed)
    frames = torch.rand(n_frames, 32, 32, 3, generator=g) * 2 - 1
    _, idx = vq.encode(frames.cuda())                     # (T, 16, 16) int32 codes
    per = idx.reshape(n_frames, -1).cpu().long()
    g2 = torch.Generator().manual_seed(seed + 1)
    text = lambda n: torch.randint(3, text_vocab, (n,), generator=g2)
    toks, vmask = [text(37)], [torch.zeros(37, dtype=torch.bool)]
    for j in range(n_frames):
        toks += [per[j], torch.tensor([EOV_TOKEN if j == n_frames - 1 else EOF_TOKEN])]
        vmask += [torch.ones(per.shape[1] + 1, dtype=torch.bool)]
    toks += [text(50)]
    vmask += [torch.zeros(50, dtype=torch.bool)]
    return torch.cat(toks)[None], torch.cat(vmask)[None], per


def test_vision_text_slice_matches_cpu_oracle():
    import torch
    from lwm_amd.vision_llama import VideoLLaMAConfig, VideoLLaMAForCausalLM
    from oracle import llama_model_ref as M
    cfg = VideoLLaMAConfig(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                           num_attention_heads=4, max_sequence_length=4096, theta=1e7, scan_mlp=False)
    toks, vmask, per = _tokens_from_frames(5, cfg.vocab_size, 3)
    assert per.shape[1] == 256 and len(torch.unique(per)) > 20        # the tokeniser produced real codes
    S = toks.shape[1] - 1
    inp, tgt, ivm, tvm = toks[:, :-1], toks[:, 1:], vmask[:, :-1], vmask[:, 1:]
    g = torch.Generator().manual_seed(9)
    lm = (torch.rand(1, S, generator=g) > 0.1).float()
    torch.manual_seed(0)
    model = VideoLLaMAForCausalLM(cfg).cuda()
    loss, met = model.loss(inp.cuda(), ivm.cuda(), tgt.cuda(), tvm.cuda(), lm.cuda(), chunk=512)
    loss.backward()
    st = {n: p.detach().float().cpu().clone().requires_grad_(True) for n, p in model.named_parameters()}
    rl, rmet = M.vision_text_loss(st, cfg, inp, ivm, tgt, tvm, lm)
    rl.backward()
    assert abs(loss.item() - rl.item()) <= 1e-2 * abs(rl.item()), (loss.item(), rl.item())
    for k in ("vision_loss", "text_loss"):
        assert abs(met[k].item() - rmet[k].item()) <= 1e-2 * abs(rmet[k].item()), k
    for n, p in model.named_parameters():
        a, b = p.grad.float().cpu().flatten().double(), st[n].grad.flatten().double()
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))
        assert cos >= 0.99, (n, cos)
    # rows of the table a position does not use receive no gradient: text rows of vte / code rows of wte
    assert model.vte.grad[EOV_TOKEN + 1:].abs().max().item() == 0
