"""The `--dtype=fp32` flavour on the GPU (the reference's default dtype, lwm/train.py:36; BASELINE configs[0] is the fp32
model): the attention op on the exact-f32 matrix instruction (csrc/attn_f32.h) and the f32 elementwise kernels
(csrc/elem_f32.h) through the C ABI against the fp64 oracle -- at the bound SURVEY.md section 8c states for an fp32
kernel path, max|err| <= 1e-5 * max|ref| (the bf16 path's bound is 8e-3) -- and the model harness in float32 against the
CPU reference model: BASELINE configs[0] (LWM-7B 2-layer slice, S = 4096, fp32) run on the device."""
import numpy as np
import pytest

from oracle import attention_ref as R

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _rand(shape, seed):
    import torch
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _np(t):
    return t.detach().float().cpu().numpy()


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _masks(B, S, Sk, seg, kv, seed):
    rng = np.random.default_rng(seed)
    seg_q = seg_k = key_valid = None
    if seg:
        cuts = np.sort(rng.choice(np.arange(1, S), size=min(4, S - 1), replace=False))
        s = np.zeros((B, S), np.int32)
        for c in cuts:
            s[:, c:] += 1
        seg_q = seg_k = s
    if kv:
        key_valid = (rng.random((B, Sk)) > 0.15).astype(np.uint8)
    return seg_q, seg_k, key_valid


@pytest.mark.parametrize("B,Sq,Sk,H,causal,seg,kv", [
    (1, 256, 256, 1, True, False, False),
    (1, 1024, 1024, 2, True, False, False),
    (2, 512, 512, 3, True, True, True),
    (1, 777, 777, 2, True, False, False),      # ragged
    (1, 300, 1000, 2, False, False, True),     # q_len != kv_len, padded keys
    (1, 1, 513, 2, False, False, False),       # single query row
    (1, 2048, 2048, 2, True, True, False),     # packed segments
])
def test_f32_fwd_bwd_vs_oracle(B, Sq, Sk, H, causal, seg, kv):
    import torch
    from lwm_amd import ops
    q, k, v, do = _rand((B, Sq, H, 128), 1), _rand((B, Sk, H, 128), 2), _rand((B, Sk, H, 128), 3), _rand((B, Sq, H, 128), 4)
    seg_q, seg_k, key_valid = _masks(B, Sq, Sk, seg, kv, 5)
    t = lambda a, dt: None if a is None else torch.from_numpy(a).to(dt).cuda()
    kw = dict(causal=causal, seg_q=t(seg_q, torch.int32), seg_k=t(seg_k, torch.int32), key_valid=t(key_valid, torch.uint8))
    qd, kd, vd, dod = q.cuda(), k.cuda(), v.cuda(), do.cuda()
    out, lse = ops.attn_fwd_block(qd, kd, vd, **kw)
    assert out.dtype == torch.float32
    delta = ops.attn_bwd_delta(out, dod, lse)
    dk, dv = ops.attn_bwd_dkdv_block(qd, kd, vd, dod, lse, delta, **kw)
    dq = ops.attn_bwd_dq_block(qd, kd, vd, dod, lse, delta, **kw)
    torch.cuda.synchronize()
    okw = dict(causal=causal, seg_q=seg_q, seg_k=seg_k, key_valid=key_valid)
    ro, rl = R.dense_attention(_np(q), _np(k), _np(v), **okw)
    assert _rel(_np(out), ro) <= TOL
    fin = np.isfinite(rl)
    assert np.array_equal(np.isfinite(_np(lse)), fin), "fully-masked rows must give lse = -inf"
    assert not fin.any() or np.abs(_np(lse)[fin] - rl[fin]).max() <= 1e-5
    rq, rk, rv = R.dense_attention_bwd(_np(q), _np(k), _np(v), _np(do), **okw)[:3]
    assert _rel(_np(dq), rq) <= TOL and _rel(_np(dk), rk) <= TOL and _rel(_np(dv), rv) <= TOL
    assert dq.dtype == dk.dtype == dv.dtype == torch.float32


def test_f32_is_deterministic_and_chains_through_the_carries():
    """No atomics, fixed summation orders: repeated launches give identical bits.  Two K/V blocks chained through the
    f32 carries at global offsets (a 2-step ring on one query block) equal the one-shot launch to rounding."""
    import torch
    from lwm_amd import ops
    B, S, H = 1, 1024, 3
    q, k, v, do = (_rand((B, S, H, 128), s).cuda() for s in (61, 62, 63, 64))
    out, lse = ops.attn_fwd_block(q, k, v, causal=True)
    delta = ops.attn_bwd_delta(out, do, lse)
    run = lambda: [ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True),
                   *ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True)]
    ref = [t.clone() for t in run()]
    for _ in range(3):
        assert all(torch.equal(a, b) for a, b in zip(run(), ref))
    assert torch.equal(ops.attn_fwd_block(q, k, v, causal=True)[0], out)
    # keys in two halves: own half first, then the earlier one, through the carries
    h = S // 2
    ql, dol = q[:, h:], do[:, h:]
    acc = ops.attn_fwd_block(ql, k[:, h:], v[:, h:], q_start=h, k_start=h, causal=True, final=False)
    o2, l2 = ops.attn_fwd_block(ql, k[:, :h], v[:, :h], q_start=h, k_start=0, causal=True, out_acc=acc[0], lse_acc=acc[1],
                                carry_in=True, final=True)
    assert _rel(_np(o2), _np(out[:, h:])) <= TOL and (l2 - lse[:, :, h:]).abs().max().item() <= 1e-5
    d2 = ops.attn_bwd_delta(o2, dol, l2)
    a1 = ops.attn_bwd_dq_block(ql, k[:, h:], v[:, h:], dol, l2, d2, q_start=h, k_start=h, causal=True, final=False)
    dq2 = ops.attn_bwd_dq_block(ql, k[:, :h], v[:, :h], dol, l2, d2, q_start=h, k_start=0, causal=True, dq_acc=a1,
                                carry_in=True, final=True)
    assert _rel(_np(dq2), _np(ref[0][:, h:])) <= 5 * TOL


def test_f32_operator_surface_and_what_it_refuses():
    """ring_attention / ringattention with float32 tensors: out and gradients in float32, equal to the oracle; the
    inference kernels (dense masks, split-K) say that they take bf16."""
    import torch
    from lwm_amd import ops
    from lwm_amd.ring import ring_attention
    B, S, H = 2, 640, 2
    q, k, v, do = (_rand((B, S, H, 128), s) for s in (7, 8, 9, 10))
    seg = np.zeros((B, S), np.int32)
    seg[:, 200:] = 1
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = ring_attention(qd, kd, vd, causal=True, segment_ids=torch.from_numpy(seg).cuda())
    out.backward(do.cuda())
    torch.cuda.synchronize()
    assert out.dtype == torch.float32 and qd.grad.dtype == torch.float32
    ro, _ = R.dense_attention(_np(q), _np(k), _np(v), causal=True, seg_q=seg, seg_k=seg)
    rq, rk, rv = R.dense_attention_bwd(_np(q), _np(k), _np(v), _np(do), causal=True, seg_q=seg, seg_k=seg)[:3]
    assert _rel(_np(out), ro) <= TOL
    assert _rel(_np(qd.grad), rq) <= TOL and _rel(_np(kd.grad), rk) <= TOL and _rel(_np(vd.grad), rv) <= TOL
    with pytest.raises(ValueError, match="bf16"):
        ops.attn_fwd_splitk(q.cuda(), k.cuda(), v.cuda(), k_splits=2)
    from lwm_amd._capi import LwmError
    with pytest.raises(LwmError, match="one run of positions"):
        ops.attn_fwd_block(q[:1, :512].cuda(), k[:1, :512].cuda(), v[:1, :512].cuda(), k_piece2=(256, 4096))
    with pytest.raises(ValueError):      # operands of one call share a dtype
        ops.attn_fwd_block(q.cuda(), k.cuda().to(torch.bfloat16), v.cuda())


@pytest.mark.parametrize("n,layout_kind,schedule", [(2, "contiguous", "ring"), (4, "zigzag", "ring"), (4, "zigzag", "mesh")])
def test_f32_ring_of_n_equals_ring_of_1(n, layout_kind, schedule):
    """The sequence ring with float32 shards (n threads play the ranks on one GPU, tests/test_gpu_ring_sim.py): the pair
    form of lwm_amd/ring.py's driver over lwm_attn_*_f32, carries and the ordered f32 sum (lwm_sum_f32) included."""
    import queue
    import threading
    import torch
    from lwm_amd.ring import HipBlockOps, SeqLayout, ring_backward, ring_forward
    from tests.test_gpu_ring_sim import ThreadComm
    S, H = 256 * n, 2
    q, k, v, do = (_rand((1, S, H, 128), s).cuda() for s in (21, 22, 23, 24))
    seg = torch.zeros(1, S, dtype=torch.int32)
    seg[:, S // 3:] = 1
    seg = seg.cuda()
    lay = SeqLayout(layout_kind, n, S)
    inboxes = [queue.Queue() for _ in range(n)]
    links = {(a, b): queue.Queue() for a in range(n) for b in range(n)}
    res, errs = [None] * n, []

    def worker(r):
        try:
            idx = lay.global_index(r).cuda()
            ql, kl, vl, dol = (t[:, idx].clone() for t in (q, k, v, do))
            comm = ThreadComm(r, n, inboxes, schedule, links)
            out, lses = ring_forward(HipBlockOps, comm, ql, kl, vl, layout=lay, causal=True, segment_ids=seg)
            dq, dk, dv = ring_backward(HipBlockOps, comm, ql, kl, vl, out, lses, dol, layout=lay, causal=True, segment_ids=seg)
            torch.cuda.synchronize()
            res[r] = (idx, out, dq, dk, dv)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    assert not errs, errs
    full = [torch.zeros_like(q) for _ in range(4)]
    for idx, *parts in res:
        for dst, src in zip(full, parts):
            assert src.dtype == torch.float32
            dst[:, idx] = src
    sg = seg.cpu().numpy()
    ro, _ = R.dense_attention(_np(q), _np(k), _np(v), causal=True, seg_q=sg, seg_k=sg)
    rq, rk, rv = R.dense_attention_bwd(_np(q), _np(k), _np(v), _np(do), causal=True, seg_q=sg, seg_k=sg)[:3]
    for name, a, b in zip(("out", "dq", "dk", "dv"), full, (ro, rq, rk, rv)):
        assert _rel(_np(a), b) <= TOL, name


def _model_f32_vs_reference(cfg, S, packed, seed, chunk):
    import torch
    from lwm_amd.llama import LLaMAForCausalLM
    from oracle import llama_model_ref as M
    torch.manual_seed(seed)
    model = LLaMAForCausalLM(cfg, torch.float32).cuda()
    g = torch.Generator().manual_seed(seed + 1)
    tokens = torch.randint(0, cfg.vocab_size, (1, S + 1), generator=g)
    inp, tgt = tokens[:, :-1].contiguous(), tokens[:, 1:].contiguous()
    lm = (torch.rand(1, S, generator=g) > 0.1).float()
    seg = am = None
    if packed:
        seg = torch.zeros(1, S, dtype=torch.int32)
        seg[:, S // 3:] = 1
        seg[:, (3 * S) // 4:] = 2
        am = torch.ones(1, S, dtype=torch.int32)
        am[:, 5:9] = 0
    loss, acc = model.loss(inp.cuda(), tgt.cuda(), lm.cuda(), None if am is None else am.cuda(),
                           None if seg is None else seg.cuda(), chunk=chunk)
    loss.backward()
    assert loss.dtype == torch.float32 and all(p.grad.dtype == torch.float32 for p in model.parameters())
    st = {n_: p.detach().float().cpu().clone().requires_grad_(True) for n_, p in model.named_parameters()}
    rl, ra = M.forward_loss(st, cfg, inp, tgt, lm, am, seg)
    rl.backward()
    # float32 on both sides (the oracle model is torch-CPU f32): what separates them is summation order
    assert abs(loss.item() - rl.item()) <= 2e-5 * abs(rl.item()), (loss.item(), rl.item())
    assert abs(acc.item() - ra.item()) <= 1e-6 + 2.0 / S
    worst = (None, 1.0)
    for n_, p in model.named_parameters():
        a, b = p.grad.cpu().flatten().double(), st[n_].grad.flatten().double()
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))
        worst = (n_, cos) if cos < worst[1] else worst
        assert cos >= 1 - 1e-5, (n_, cos)
        assert abs(float(a.norm() / b.norm().clamp_min(1e-30)) - 1) <= 1e-3, n_
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()), n_
    return loss.item(), rl.item(), worst


def test_f32_model_small_slice_packed():
    from lwm_amd.llama import LLaMAConfig
    cfg = LLaMAConfig(vocab_size=4096, hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=4,
                      max_sequence_length=2048, scan_mlp_chunk_size=256)
    _model_f32_vs_reference(cfg, 1536, True, 0, 1024)


def test_f32_config0_7b_two_layer_slice_4k():
    """BASELINE configs[0] -- LWM-7B, 2-layer slice, seq = 4096, bs = 1, fp32 -- on the MI355X in float32 against the CPU
    reference model in float32: loss to 2e-5, every parameter gradient to cosine 1 - 1e-5."""
    from lwm_amd.llama import LLaMAConfig
    cfg = LLaMAConfig.load_config("7b", num_hidden_layers=2)
    _model_f32_vs_reference(cfg, 4096, False, 1, 1024)


def test_f32_cli_train_takes_the_reference_launchers_dtype():
    """python -m lwm_amd.cli.train --dtype=fp32 (scripts/run_train_text.sh:21): a few optimizer steps of the debug model in
    float32 run and give finite losses of the size a random model gives."""
    import torch
    from lwm_amd.cli import train
    hist = train.main(["--load_llama_config=debug", "--mesh_dim=1,-1,1,1", "--dtype=fp32", "--tokenizer=synthetic", "--modality=text",
                       "--total_steps=3", "--log_freq=0",
                       "--update_llama_config=dict(theta=10000,max_sequence_length=2048,scan_query_chunk_size=256)",
                       "--train_dataset.json_dataset.seq_length=1024", "--train_dataset.json_dataset.batch_size=2",
                       "--optimizer.adamw_optimizer.lr=1e-3", "--optimizer.adamw_optimizer.lr_warmup_steps=1"])
    assert len(hist) == 3 and all(np.isfinite(h["loss"]) for h in hist) and 8.0 < hist[0]["loss"] < 13.0


def test_f32_cached_generation_equals_the_uncached_model_and_the_cpu_reference():
    """Cached inference in float32 on one rank (the inference launchers' --dtype='fp32', scripts/run_eval_needle.sh:17):
    prefill into the KV cache and the one-token steps both run through the f32 training-op kernel with the mask handed over
    as its structure (key <= cache_index + query AND attention_mask[key], lwm/llama.py:577-592), the cache is written by
    lwm_kv_cache_write as bytes.  Every step's logits equal the CPU f32 reference model's logits at that position of the
    final sequence (teacher forcing) to 1e-4 of their maximum; the greedy tokens are the reference's argmax; a left-padded
    prompt works; generate(graph=True) says that it captures bf16 kernels."""
    import torch
    from lwm_amd.llama import LLaMAConfig, LLaMAForCausalLM
    from oracle import llama_model_ref as M
    cfg = LLaMAConfig(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                      max_sequence_length=512)
    torch.manual_seed(11)
    model = LLaMAForCausalLM(cfg, torch.float32).cuda()
    g = torch.Generator().manual_seed(12)
    B, S, new = 2, 45, 6
    prompt = torch.randint(3, cfg.vocab_size, (B, S), generator=g)
    am = torch.ones(B, S, dtype=torch.int32)
    am[1, :7] = 0                                        # row 1 is left-padded (lwm/vision_chat.py:136-140)
    with torch.no_grad():
        out, logits = model.generate(prompt.cuda(), am.cuda(), max_new_tokens=new, max_length=64, return_logits=True)
    assert logits.dtype == torch.float32 and out.shape == (B, S + new)
    st = {n_: p.detach().float().cpu() for n_, p in model.named_parameters()}
    full_am = torch.cat([am, torch.ones(B, new, dtype=torch.int32)], 1)
    with torch.no_grad():
        ref = M.forward_logits(st, cfg, out.cpu(), full_am)          # positions are the mask's cumulative sum there too
    for j in range(new):
        r = ref[:, S - 1 + j]
        assert (logits[:, j].cpu() - r).abs().max().item() <= 1e-4 * r.abs().max().item(), j
        assert torch.equal(out[:, S + j].cpu(), r.argmax(-1)), j
    with pytest.raises(NotImplementedError, match="bf16"):
        model.generate(prompt.cuda(), am.cuda(), max_new_tokens=3, graph=True)


def test_f32_vision_text_slice_matches_cpu_oracle():
    """The vision-language harness (lwm/vision_llama.py: vte / wte choice, two heads, 0.5 * (vision CE + text CE),
    lwm/train.py:183-202) in float32 against the CPU f32 oracle model: the bf16 test's case (tests/test_gpu_vision_llama.py)
    at the f32 bound."""
    import torch
    from lwm_amd.vision_llama import VideoLLaMAConfig, VideoLLaMAForCausalLM
    from oracle import llama_model_ref as M
    from tests.test_gpu_vision_llama import _tokens_from_frames
    cfg = VideoLLaMAConfig(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
                           max_sequence_length=4096, theta=1e7, scan_mlp=False)
    toks, vmask, per = _tokens_from_frames(5, cfg.vocab_size, 3)
    S = toks.shape[1] - 1
    inp, tgt, ivm, tvm = toks[:, :-1], toks[:, 1:], vmask[:, :-1], vmask[:, 1:]
    lm = (torch.rand(1, S, generator=torch.Generator().manual_seed(9)) > 0.1).float()
    torch.manual_seed(0)
    model = VideoLLaMAForCausalLM(cfg, torch.float32).cuda()
    loss, met = model.loss(inp.cuda(), ivm.cuda(), tgt.cuda(), tvm.cuda(), lm.cuda(), chunk=512)
    loss.backward()
    st = {n_: p.detach().float().cpu().clone().requires_grad_(True) for n_, p in model.named_parameters()}
    rl, rmet = M.vision_text_loss(st, cfg, inp, ivm, tgt, tvm, lm)
    rl.backward()
    assert abs(loss.item() - rl.item()) <= 2e-5 * abs(rl.item()), (loss.item(), rl.item())
    for k in ("vision_loss", "text_loss"):
        assert abs(met[k].item() - rmet[k].item()) <= 2e-5 * abs(rmet[k].item()), k
    for n_, p in model.named_parameters():
        a, b = p.grad.cpu().flatten().double(), st[n_].grad.flatten().double()
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))
        assert cos >= 1 - 1e-5, (n_, cos)


def test_f32_inference_entry_points_take_the_reference_launchers_dtype():
    """python -m lwm_amd.cli.vision_chat / vision_generation --dtype=fp32 on the debug model: a few sampled tokens / one
    frame of codes decoded, in float32 throughout."""
    from lwm_amd.cli import vision_chat, vision_generation
    small = ["--load_llama_config=debug", "--mesh_dim=1,-1,1,1", "--dtype=fp32", "--tokenizer=synthetic"]
    ans = vision_chat.main(small + ["--prompt=What is the video about?", "--input_file=synthetic:2", "--max_n_frames=2",
                                    "--update_llama_config=dict(sample_mode='text',max_sequence_length=2048,vocab_size=32000)"],
                           max_new_tokens=4)
    assert isinstance(ans, str)


def test_every_row_of_the_headline_workload_bf16_against_the_f32_flavour():
    """BASELINE configs[1] at FULL size -- S = 32768, 32 heads, causal -- where the fp64 oracle can only visit windows
    (tests/test_gpu_attention.py): EVERY element of out, dq, dk, dv of the bf16 kernels is held to the float32 flavour of
    the op on the same (bf16-valued) inputs -- another kernel, another matrix instruction, another data layout, itself
    pinned to 1e-5 of the fp64 oracle above -- at the bf16 path's stated bounds (tests/_parity.py: 8e-3 of the tensor's
    maximum, 2.5e-2 per row of its own scale, cosine 0.9999).  dq per row against the f32 gradient for the SAVED bf16
    output (the residual the bf16 backward really has), as tests/_parity.py::check_dq does."""
    import torch
    from lwm_amd import ops
    from tests._parity import COS, ROW_FLOOR, ROW_TOL, STATS, TOL
    S, H = 32768, 32
    g = torch.Generator(device="cuda").manual_seed(1234)
    q, k, v, do = (torch.randn(1, S, H, 128, device="cuda", generator=g).to(torch.bfloat16) for _ in range(4))
    out, lse = ops.attn_fwd_block(q, k, v, causal=True)
    delta = ops.attn_bwd_delta(out, do, lse)
    dk, dv = ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True)
    dq = ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True)
    qf, kf, vf, dof = (t.float() for t in (q, k, v, do))
    ro, rl = ops.attn_fwd_block(qf, kf, vf, causal=True)
    assert (lse - rl).abs().max().item() <= 2e-3
    rdelta = ops.attn_bwd_delta(ro, dof, rl)
    rk, rv = ops.attn_bwd_dkdv_block(qf, kf, vf, dof, rl, rdelta, causal=True)
    sdelta = ops.attn_bwd_delta(out.float(), dof, lse)          # the residual the bf16 backward has: its own saved output
    rq_saved = ops.attn_bwd_dq_block(qf, kf, vf, dof, lse, sdelta, causal=True)
    torch.cuda.synchronize()

    def held(name, got, ref, row_tol=ROW_TOL, floor=ROW_FLOOR):
        got = got.float()
        gmax = ref.abs().max().clamp_min(1e-9)
        diff = (got - ref).abs()
        err = (diff.max() / gmax).item()
        row = (diff.amax(-1) / torch.maximum(ref.abs().amax(-1), floor * gmax)).max().item()
        cos = ((got.double() * ref.double()).sum() / (got.double().norm() * ref.double().norm()).clamp_min(1e-30)).item()
        STATS.append((f"{name}(S=32768,every-row,vs-f32-flavour)", err, row, cos))     # -> gpurun_out/parity_stats.json
        assert err <= TOL and row <= row_tol and cos >= COS, (name, err, row, cos)
        return err, row

    held("out", out, ro)
    held("dk", dk, rk)
    held("dv", dv, rv)
    held("dq", dq, rq_saved)
