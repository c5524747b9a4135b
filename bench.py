#!/usr/bin/env python
"""bench.py -- LWM-7B RingAttention hot path, forward+backward, tokens/s.

One "step" = the blockwise attention forward + backward of ALL 32 layers of
LWM-7B (32 heads x 128, lwm/llama.py:70-81) over one synthetic packed batch
(B=1), i.e. one pass of the north-star hot path.  q/k/v/dO are synthetic
N(0,1) bf16 already resident in HBM when the timed region starts.

  N=1 : BASELINE.json configs[1]  (S=32768, ring=1, no send/recv)
  N>1 : the SAME problem (S=32768) ring-sharded over N GPUs -- strong scaling, so that the per-N values of
        N = 1, 2, 4, 8 compare (attention cost is quadratic in S: a different S per N would not); zigzag
        ownership for causal balance, K/V and dK/dV exchanged over RCCL (mesh schedule by default,
        --schedule ring for the reference's pattern).  `--seq 131072` is BASELINE configs[2]'s problem
        (128K over the ring), `--seq 1048576 --packed` configs[4]'s.
        Also reported at N>1 (object `exchange`): the step re-run with a communicator
        that moves nothing (what the exchange costs on top of the launches) and one
        layer of the SAME problem on a single GPU (like-for-like strong scaling).

Launch: `python bench.py --gpus N ...` is enough -- with N > 1 and no torch.distributed
environment the script re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU); started under torchrun by someone
else (RANK / WORLD_SIZE set) it just takes its rank.

Prints ONE JSON line (rank 0).  `value` = whole-job tokens/s = S*steps/time.
At N=1 the line also carries `roofline`, `cpu_baseline` and secondary legs that are never
part of `value`: vqgan, packed, model_slice, config0_fp32, decode, generate, ring8_compute_model, elementwise.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_LAYERS, N_HEADS, HEAD_DIM, D_MODEL = 32, 32, 128, 4096
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md


def gemm_unit_flops(S):
    """One causal S x S x d_model GEMM 'unit' (SURVEY.md section 8d): fwd = 2 units,
    bwd = 5 units algorithmic (the two-kernel backward executes 7)."""
    return float(S) * S * D_MODEL


def _cpu_threads():
    """The one `cores` convention of this file: the threads a CPU leg actually computes with (torch's intra-op pool;
    the OpenMP oracle is given the same number)."""
    import torch
    return int(torch.get_num_threads())


def cpu_config1():
    """BASELINE configs[0] end to end on the host cores: the fp32 CPU model (oracle/llama_model_ref.py) as a 2-layer slice
    of LWM-7B (d_model 4096, 32 heads, FFN 11008, vocab 32000), B = 1, S = 4096, loss forward + backward, random weights
    N(0, 0.02^2).  One pass (~15-30 s).  FLOPs: 6 x (matmul parameters of the slice) x tokens + the causal attention's
    7 * S^2 * d_model per layer."""
    import types
    import torch
    from oracle import llama_model_ref as M
    S, d, H, F, V, L = 4096, D_MODEL, N_HEADS, 11008, 32000, 2
    cfg = types.SimpleNamespace(num_attention_heads=H, hidden_size=d, theta=1e4, max_sequence_length=S,
                                num_hidden_layers=L, rms_norm_eps=1e-6)
    g = torch.Generator().manual_seed(0)
    w = lambda *shape: (torch.randn(*shape, generator=g) * 0.02).requires_grad_(True)
    st = {"wte": w(V, d), "lm_head": w(d, V), "ln_f.kernel": torch.ones(d, requires_grad=True)}
    for i in range(L):
        p = f"h.{i}."
        st.update({p + "attention_norm.kernel": torch.ones(d, requires_grad=True), p + "ffn_norm.kernel": torch.ones(d, requires_grad=True),
                   p + "attention.wq": w(d, d), p + "attention.wk": w(d, d), p + "attention.wv": w(d, d), p + "attention.wo": w(d, d),
                   p + "feed_forward.w1": w(d, F), p + "feed_forward.w3": w(d, F), p + "feed_forward.w2": w(F, d)})
    tok = torch.randint(0, V, (1, S + 1), generator=g)
    t0 = time.perf_counter()
    loss, _ = M.forward_loss(st, cfg, tok[:, :-1], tok[:, 1:])
    loss.backward()
    dt = time.perf_counter() - t0
    matmul_params = L * (4 * d * d + 3 * d * F) + d * V
    flops = 6.0 * matmul_params * S + 7.0 * S * S * d * L
    return {"workload": "LWM-7B 2-layer slice + lm_head, B=1, S=4096, fp32, loss fwd+bwd on the host cores [BASELINE configs[0]]",
            "seconds": dt, "tokens_per_s": S / dt, "gflops": flops / dt / 1e9, "loss": float(loss.detach()),
            "kind": "port", "cores": _cpu_threads(), "sample": "oracle/llama_model_ref.forward_loss + backward, 1 pass"}


CPU_BASELINE_CONVENTION_R05 = ("value = S / (seconds of ONE timed pass at the workload's own S: 1 head, 1 layer, fwd+bwd) / 32 heads / 32 layers; "
                               "the pass is timed at 8 and at 32 threads and the faster one counts (thread sweep at S=4096 x 8 heads over "
                               "{8, 32, all} kept as context)")       # (the lines of round 5 carry this text)
CPU_BASELINE_CONVENTION = ("value = S / (wall seconds of ONE LAYER's 32 heads at the workload's own S, fwd+bwd, the heads run side by side "
                           "as floor(host threads / 8) processes of 8 BLAS threads each) / 32 layers; cores = the host's thread count. "
                           "(Rounds 1-5 timed ONE head on 8 or 32 threads and multiplied by 32 heads, leaving most of a many-core host "
                           "idle: kept as measured_at_one_head / value_one_head_convention for continuity.)")


def cpu_baseline(S_target, full=True):
    """Port of the reference's blockwise attention on PyTorch-CPU fp32 (oracle/attention_torch_cpu.py) on the host cores.
    THE CONVENTION (round 6, frozen; tests/test_bench_contract.py pins it): `value` is MEASURED at the workload's own
    sequence length with EVERY host thread at work -- the 32 heads of one layer are independent, and the port (many small
    batched matmuls over 1024 x 1024 tiles) runs best on about 8 threads, so floor(threads / 8) processes of 8 threads
    each run the heads side by side after a common barrier; value = S / (that wall time x 32 layers).  On a host too small
    to do 32 heads in the sample's budget, 4 heads per process are timed and the remaining heads counted as further
    rounds of the same (stated in `sample`).  Nothing is scaled in S.  The one-head figure of rounds 1-5 (one pass on 8 or
    32 threads x 32 heads x 32 layers) stays as `measured_at_one_head` / `value_one_head_convention`; `op_points`
    (S = 4096 x 8 heads, 8192 x 2, 16384 x 1) as context; `config1` is BASELINE configs[0] end to end."""
    import torch
    from oracle.attention_torch_cpu import blockwise_fwd_bwd
    g = torch.Generator().manual_seed(0)

    def op_point(S, H, budget_s, max_reps):
        q, k, v, do = (torch.randn(1, S, H, HEAD_DIM, generator=g) for _ in range(4))
        t0 = time.perf_counter()
        reps = 0
        while True:
            blockwise_fwd_bwd(q, k, v, do)
            reps += 1
            if time.perf_counter() - t0 > budget_s or reps >= max_reps:
                break
        dt = (time.perf_counter() - t0) / reps
        return {"S": S, "heads": H, "reps": reps, "seconds_per_pass": dt, "gflops": 7.0 * S * S * (H * HEAD_DIM) / dt / 1e9}

    w = [torch.randn(1, 1024, 8, HEAD_DIM, generator=g) for _ in range(4)]
    blockwise_fwd_bwd(*w)  # warm the BLAS threads
    # The port is many small batched matmuls and elementwise passes over 1024 x 1024 tiles: on a many-core host it runs
    # SLOWER with every thread than with a few (30 GFLOP/s on 128 threads, 130 on 8): the thread count is the sweep's best.
    all_threads = _cpu_threads()
    sweep = {}
    for t in sorted({min(8, all_threads), min(32, all_threads), all_threads}):
        torch.set_num_threads(t)
        blockwise_fwd_bwd(*w)
        sweep[t] = op_point(4096, 8, 3.0, 2)
    best = max(sweep, key=lambda t: sweep[t]["gflops"])
    # THE sample: the workload's S, one head, one layer, one pass -- at 8 and at 32 threads, the faster counts
    tries = {}
    for t in sorted({min(8, all_threads), min(32, all_threads)}):
        torch.set_num_threads(t)
        tries[t] = op_point(S_target, 1, 0.0, 1)
    sample_threads = max(tries, key=lambda t: tries[t]["gflops"])
    head = tries[sample_threads]
    torch.set_num_threads(best)
    points = [sweep[best]]
    if full:
        points += [op_point(8192, 2, 6.0, 2), op_point(16384, 1, 6.0, 1)]
    # THE value: one layer's heads side by side on every host thread
    from oracle.attention_torch_cpu import heads_in_parallel
    tpw = min(8, all_threads)
    workers = max(1, all_threads // tpw)
    hpw = min(-(-N_HEADS // workers), 4)               # (a small host: 4 heads per process, the rest counted as further rounds)
    parallel_error = None
    try:
        wall, heads_done = heads_in_parallel(S_target, workers, hpw, tpw)
    except Exception as e:      # noqa: BLE001 -- a host that cannot run the worker processes still gets a baseline:
        # the one-head pass of rounds 1-5 on its own threads, counted as `workers` = 1 (stated in `sample`)
        parallel_error = repr(e)[:300]
        workers, hpw, tpw = 1, 1, sample_threads
        wall, heads_done = head["seconds_per_pass"], 1
    layer_seconds = wall * N_HEADS / heads_done
    res = {
        "value": S_target / (layer_seconds * N_LAYERS),
        "unit": "tokens/s",
        "cores": all_threads if parallel_error is None else sample_threads,
        "kind": "port",
        "gflops": 7.0 * S_target * S_target * D_MODEL / layer_seconds / 1e9,
        "convention": CPU_BASELINE_CONVENTION,
        "measured_at": {"S": S_target, "layers": 1, "heads_timed": heads_done, "processes": workers, "threads_per_process": tpw,
                        "wall_seconds": wall, "layer_seconds": layer_seconds},
        "measured_at_one_head": {"S": S_target, "heads": 1, "layers": 1, "seconds": head["seconds_per_pass"], "threads": sample_threads,
                                 "seconds_by_threads": {str(t): round(v["seconds_per_pass"], 3) for t, v in tries.items()},
                                 "gflops": head["gflops"]},
        "value_one_head_convention": S_target / (head["seconds_per_pass"] * N_HEADS * N_LAYERS),
        "thread_sweep_gflops": {str(t): round(v["gflops"], 1) for t, v in sweep.items()},
        "op_points": points,
        "parallel_error": parallel_error,
        "sample": f"oracle/attention_torch_cpu.blockwise_fwd_bwd fp32, chunks 1024/1024: {heads_done} heads of ONE layer at S={S_target} as "
                  f"{workers} processes x {tpw} threads ({hpw} heads each) on a {all_threads}-thread host, {wall:.2f} s wall"
                  + ("" if heads_done == N_HEADS else f", the other {N_HEADS - heads_done} heads counted as further rounds of the same")
                  + "; x 32 layers (independent repetitions; no scaling in S)",
    }
    if full:
        try:
            torch.set_num_threads(all_threads)       # the dense model is large GEMMs: every thread
            res["config1"] = cpu_config1()
        except Exception as e:      # a baseline leg must not cost the bench line
            res["error"] = repr(e)
    torch.set_num_threads(all_threads)
    return res


VQGAN_ENC_GFLOP, VQGAN_DEC_GFLOP = 216.6, 477.4   # per 256x256 frame, SURVEY.md Appendix B
MFMA_F32_PEAK_TFLOPS = 157.3                      # exact-f32 MFMA, MI355X_MICROARCH.md


def vqgan_leg(torch, frames=64, reps=3, config4_frames=1020):
    """Secondary leg (not part of `value`): VQGAN encode/decode of synthetic
    256x256 frames U(-1,1), random weights of the default VQGANConfig
    (lwm/vqgan.py:62-77), frames resident in HBM; plus the C oracle on the host
    cores for one frame (cpu_baseline of this leg).  64 frames per call: frames are independent
    (BASELINE configs[3] tokenises 1020 of them) and the late-encoder / early-decoder layers have
    only 256-4096 output pixels per frame -- at 32 frames their 384 tiles of 128 x 128 fill 256 CUs 1.5 times, at 64
    exactly 3 times (round 3 sweep, same box: 32 / 64 / 128 frames = 439 / 457 / 461 encode, 209 / 213 / 214 decode
    frames/s; profiles/r03_vqgan_frames.txt)."""
    import numpy as np
    from lwm_amd.vqgan import VQGAN, VQGANConfig, random_params
    cfg = VQGANConfig.get_default_config()
    params = random_params(cfg, 0)
    vq = VQGAN(params=params, config=cfg)
    g = torch.Generator(device="cuda").manual_seed(0)
    px = torch.rand(frames, 256, 256, 3, generator=g, device="cuda") * 2 - 1
    _, idx = vq.encode(px)          # untimed warm-up at the timed shape (the caching allocator grows here, not in the timed calls)
    vq.decode(idx)
    torch.cuda.synchronize()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps, out

    t_enc, (_, idx) = timed(lambda: vq.encode(px))
    t_dec, _ = timed(lambda: vq.decode(idx))
    # BASELINE configs[3]: a 256K-token vision-language sequence = floor(262144 / 257) = 1020 frames
    # (lwm/vision_chat.py:91-108 tokenises them one by one); here in chunks of `frames`, frames resident in HBM
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_chunks = -(-config4_frames // frames)
    e0.record()
    n_codes = 0
    for c in range(n_chunks):
        n = min(frames, config4_frames - c * frames)
        _, ids = vq.encode(px[:n])
        n_codes += ids.numel()
    e1.record()
    torch.cuda.synchronize()
    t_cfg4 = e0.elapsed_time(e1) * 1e-3
    res = {
        "workload": f"VQGAN default config, {frames} frames 256x256, f32 (exact-f32 MFMA), random weights",
        "encode_frames_per_s": frames / t_enc, "decode_frames_per_s": frames / t_dec,
        "encode_tflops": frames * VQGAN_ENC_GFLOP / t_enc / 1e3,
        "decode_tflops": frames * VQGAN_DEC_GFLOP / t_dec / 1e3,
        "config4_tokenisation": {
            "workload": f"{config4_frames} frames 256x256 -> {n_codes} codes (+1 delimiter each = {config4_frames * 257} tokens: "
                        f"BASELINE configs[3]'s 256K vision-language sequence), encode in chunks of {frames}",
            "seconds": t_cfg4, "frames_per_s": config4_frames / t_cfg4,
            "tflops": config4_frames * VQGAN_ENC_GFLOP / t_cfg4 / 1e3},
        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_F32_PEAK_TFLOPS,
                     "achieved": frames * VQGAN_DEC_GFLOP / t_dec / 1e3,
                     "frac": frames * VQGAN_DEC_GFLOP / t_dec / 1e3 / MFMA_F32_PEAK_TFLOPS,
                     "kernel": "conv_patch_c256 / conv_patch_c128 / conv_igemm (decode pass)"},
    }
    try:
        os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_threads()))     # the one `cores` convention (set before the library loads)
        from oracle import vqgan_ref as R
        x1 = px[:1].cpu().numpy()
        t0 = time.perf_counter()
        _, ridx = R.encode(params, x1, cfg.as_dict())
        t1 = time.perf_counter()
        R.decode(params, ridx, cfg.as_dict())
        t2 = time.perf_counter()
        res["cpu_baseline"] = {"encode_frames_per_s": 1.0 / (t1 - t0), "decode_frames_per_s": 1.0 / (t2 - t1),
                               "cores": int(os.environ.get("OMP_NUM_THREADS", _cpu_threads())), "kind": "port",
                               "sample": "oracle/vqgan_ref (C, OpenMP) encode+decode of 1 frame"}
        res["indices_match_oracle"] = bool((idx[0].cpu().numpy() == ridx[0]).all())
    except Exception as e:  # the oracle is a checker; its absence must not break the bench line
        res["cpu_baseline"] = {"error": repr(e)}
    return res


ATTN_KERNEL_SOURCES = ("attn_common.h", "attn_fwd64.h", "attn_bwd.h", "attn_bwd64.h", "wave_ops.h")


def attn_kernel_stamp():
    """sha256 over the sources of the attention kernels of the main workload (lwm_amd/csrc/): scripts/summarise_pmc.py
    writes it into every PMC summary, and pmc_traffic() only accepts a summary whose stamp equals the tree's -- a
    counter pass of an older kernel cannot label a newer one (the round prefix of the file name is not consulted)."""
    import hashlib
    h = hashlib.sha256()
    for name in ATTN_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "lwm_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def pmc_traffic(kernel, S):
    """(HBM bytes per launch of `kernel`, profile file) from a committed PMC pass of this same command
    (profiles/*pmc_attention*.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate rocprofv3 --pmc
    passes) WHOSE KERNEL SOURCES ARE THE TREE'S (`kernel_source_stamp`, see attn_kernel_stamp).  bench.py cannot
    collect counters itself; (None, None) when no such profile is committed -- a stale file must not label a
    newer kernel."""
    return pmc_traffic_any(kernel, S, same_sources=True)


def pmc_traffic_any(kernel, S, same_sources):
    """same_sources=True: only a summary stamped with the tree's kernel sources counts (what `roofline.traffic` quotes);
    False: the newest stamped summary of OTHER sources -> (bytes, file, its stamp), reported beside a null `traffic` as
    `traffic_of_earlier_kernel_sources` so that the reader sees what the last counter pass measured and of which code."""
    if S != 32768:
        return (None, None) if same_sources else (None, None, None)
    import glob
    stamp = attn_kernel_stamp()
    best = (None, None) if same_sources else (None, None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_attention*.json"))):
        try:
            doc = json.load(open(f))
            theirs = doc.get("kernel_source_stamp")
            if (theirs == stamp) != same_sources or theirs is None:
                continue
            ks = doc["kernels"]
        except Exception:
            continue
        for name, d in ks.items():
            if name.startswith(kernel) and "hbm_traffic_bytes" in d:
                best = (d["hbm_traffic_bytes"], os.path.relpath(f, ROOT)) + (() if same_sources else (theirs,))
    return best


def packed_leg(torch, S=32768, layers=4):
    """Secondary leg: masked sequence packing (BASELINE config #5 style) on one GPU --
    the same S = 32768 batch cut into documents (log-uniform lengths in [S/256, S/4], seed 0),
    segment_ids driving the in-kernel document skipping.  FLOPs counted over visible pairs."""
    import numpy as np
    from lwm_amd.ring import HipBlockOps, SeqLayout, SingleComm, ring_backward, ring_forward
    rng = np.random.default_rng(0)
    seg = np.zeros((1, S), np.int32)
    pos, d, lens = 0, 0, []
    while pos < S:
        ln = min(int(np.exp(rng.uniform(np.log(S / 256), np.log(S / 4)))), S - pos)
        seg[:, pos:pos + ln] = d
        lens.append(ln)
        pos, d = pos + ln, d + 1
    segd = torch.from_numpy(seg).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: torch.randn(1, S, N_HEADS, HEAD_DIM, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    lay, comm = SeqLayout("contiguous", 1, S), SingleComm()

    def step():
        for _ in range(layers):
            out, lses = ring_forward(HipBlockOps, comm, q, k, v, layout=lay, causal=True, segment_ids=segd)
            ring_backward(HipBlockOps, comm, q, k, v, out, lses, do, layout=lay, causal=True, segment_ids=segd)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / layers
    flops = 7.0 * sum(l * l for l in lens) * D_MODEL
    return {"workload": f"attention fwd+bwd, S={S}, {len(lens)} packed documents (lengths {min(lens)}..{max(lens)}), "
                        f"per layer", "ms_per_layer": dt * 1e3, "tokens_per_s_32_layers": S / (dt * N_LAYERS),
            "algorithmic_tflops": flops / dt / 1e12,
            "dense_equivalent_speedup": "see kernels above: the same S unpacked takes ~36 ms per layer"}


def model_slice_leg(torch, S=32768):
    """Secondary leg: the 2-layer slice of LWM-7B (BASELINE config #1's model) at S = 32768, one
    forward+backward through the harness (lwm_amd/llama.py): HIP RMSNorm / RoPE / RingAttention /
    SwiGLU gate / chunked loss + hipBLASLt projections.  Shows the hot path in position."""
    from lwm_amd.llama import LLaMAConfig, LLaMAForCausalLM
    cfg = LLaMAConfig.load_config("7b", num_hidden_layers=2, max_sequence_length=S, theta=1e7)
    torch.manual_seed(0)
    model = LLaMAForCausalLM(cfg).cuda()
    tok = torch.randint(0, cfg.vocab_size, (1, S + 1), device="cuda")

    def step():
        model.zero_grad(set_to_none=True)
        loss, _ = model.loss(tok[:, :-1], tok[:, 1:], chunk=8192)
        loss.backward()
        return loss

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": f"LWM-7B 2-layer slice + lm_head, B=1, S={S}, bf16, fwd+bwd", "ms": dt * 1e3,
            "tokens_per_s": S / dt, "loss": float(loss.detach())}


def config0_fp32_leg(torch, S=4096):
    """Secondary leg: BASELINE configs[0] -- LWM-7B, 2-layer slice, seq = 4096, bs = 1, FLOAT32 (the reference's default
    --dtype, lwm/train.py:36) -- loss forward + backward on the MI355X through the f32 flavour of the path (csrc/attn_f32.h
    on the exact-f32 matrix instruction, csrc/elem_f32.h, library GEMMs in f32): the SAME workload, dtype and FLOP count as
    `cpu_baseline.config1` (the reference-shaped CPU model on the host cores), so the two figures of one line compare like
    with like.  Also the f32 attention op alone at that S, 32 heads, against the f32 MFMA peak (157.3 TF/s dense)."""
    from lwm_amd import ops
    from lwm_amd.llama import LLaMAConfig, LLaMAForCausalLM
    cfg = LLaMAConfig.load_config("7b", num_hidden_layers=2, max_sequence_length=S)
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = LLaMAForCausalLM(cfg, torch.float32)
    tok = torch.randint(0, cfg.vocab_size, (1, S + 1), device="cuda")

    def step():
        model.zero_grad(set_to_none=True)
        loss, _ = model.loss(tok[:, :-1], tok[:, 1:], chunk=4096)
        loss.backward()
        return loss

    step()
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    d, F, V, L = D_MODEL, 11008, cfg.vocab_size, 2
    flops = 6.0 * (L * (4 * d * d + 3 * d * F) + d * V) * S + 7.0 * S * S * d * L
    del model
    g = torch.Generator(device="cuda").manual_seed(1)
    q, k, v, do = (torch.randn(1, S, N_HEADS, HEAD_DIM, device="cuda", generator=g) for _ in range(4))
    o, lse = ops.attn_fwd_block(q, k, v, causal=True)
    delta = ops.attn_bwd_delta(o, do, lse)

    def attn():
        ops.attn_fwd_block(q, k, v, causal=True)
        ops.attn_bwd_delta(o, do, lse, delta)
        ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True)
        ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True)

    attn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        attn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    a_tf = 7.0 * S * S * D_MODEL / (ms * 1e-3) / 1e12
    return {"workload": f"LWM-7B 2-layer slice + lm_head, B=1, S={S}, fp32 (--dtype=fp32), loss fwd+bwd on one MI355X [BASELINE configs[0]]",
            "ms": dt * 1e3, "tokens_per_s": S / dt, "gflops": flops / dt / 1e9, "loss": float(loss.detach()), "dtype": "f32",
            "attention_f32": {"workload": f"attention fwd+bwd, f32 operands, S={S}, 32 heads x 128, causal, one layer",
                              "ms_per_layer": ms,
                              "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": 157.3, "achieved": a_tf, "frac": a_tf / 157.3,
                                           "kernel": "attn_fwd_f32_kernel + attn_bwd_dq_f32_kernel + attn_bwd_dkdv_f32_kernel",
                                           "note": "algorithmic 7 units over the three launches (9 executed); f32 MFMA dense peak"}}}


def model_full_leg(torch, S=32768, layers=N_LAYERS, mlp_chunk=8192, scan_mlp=False):
    """Secondary leg: ALL 32 layers of LWM-7B (6.74 B parameters, random
    init, bf16) at S = 32768, one forward+backward through the harness on ONE MI355X -- embedding, 32 x
    (RMSNorm, QKV, RoPE, RingAttention ring=1, wo, RMSNorm, blockwise SwiGLU FFN with chunk recompute),
    final norm, chunked lm_head + cross-entropy.  No optimizer step (the hot path ends at the gradients).
    Model FLOPs per token: 6 x 6.74e9 dense + 7 x S x d_model x L attention."""
    from lwm_amd.llama import LLaMAConfig, LLaMAForCausalLM
    # scan_mlp / scan_mlp_chunk_size are the reference's own knobs (lwm/llama.py:673-678, :728-734): True = FFN in
    # sequence chunks with its activations recomputed in the backward (the TPU memory setting; chunk 1024 by
    # default, 8192 in scripts/run_train_vision_text.sh), False = one pass, activations kept (scripts/run_sample_*.sh).
    # With 288 GB per GPU the 32K-token step fits without recompute (144 GiB peak); measured on MI355X:
    # scan_mlp=False 2.39 s, chunk 32768 / 16384 / 8192 / 4096 / 1024 with recompute 2.48 / 2.51 / 2.55 / 2.61 / 2.86 s.
    torch.cuda.reset_peak_memory_stats()
    cfg = LLaMAConfig.load_config("7b", num_hidden_layers=layers, max_sequence_length=S, theta=1e7,
                                  scan_mlp_chunk_size=mlp_chunk, scan_mlp=scan_mlp)
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = LLaMAForCausalLM(cfg)
    n_params = sum(p.numel() for p in model.parameters())
    tok = torch.randint(0, cfg.vocab_size, (1, S + 1), device="cuda")

    from lwm_amd.llama_ops import use_tuned_gemms, weights_changed
    tuned = use_tuned_gemms()

    def step():
        model.zero_grad(set_to_none=True)
        weights_changed()      # no optimizer step here: every step re-lays its kernels for the GEMMs as a training step would
        loss, _ = model.loss(tok[:, :-1], tok[:, 1:], chunk=8192)
        loss.backward()
        return loss

    step()
    torch.cuda.synchronize()
    peak0 = torch.cuda.max_memory_allocated()
    t0 = time.perf_counter()
    loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dense = 6.0 * n_params * S
    attn = 7.0 * gemm_unit_flops(S) * layers
    out = {"workload": f"LWM-7B, all {layers} layers + embedding + lm_head ({n_params / 1e9:.2f} B parameters), B=1, S={S}, "
                       f"bf16, forward+backward, one GPU, scan_mlp={scan_mlp}",
           "scan_mlp": scan_mlp, "scan_mlp_chunk_size": mlp_chunk, "tuned_gemm_solutions": tuned,
           "ms_per_step": dt * 1e3, "tokens_per_s": S / dt, "loss": float(loss.detach()),
           "model_tflops": (dense + attn) / dt / 1e12, "attention_share_of_flops": attn / (dense + attn),
           "peak_hbm_gib": peak0 / 2 ** 30}
    # the leg's own roofline object: model FLOPs (dense 6 x params x S + 7 attention GEMM units per layer) over the wall
    # clock of the timed step against the dense bf16 MFMA peak, and where the step's kernel time goes, by class, from ONE
    # further step run under torch.profiler (device activity only; not the timed step)
    out["roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": 2500.0, "achieved": out["model_tflops"],
                       "frac": out["model_tflops"] / 2500.0, "flops_per_step": dense + attn,
                       "time_shares": model_time_shares(torch, step)}
    # the hand-written weight-gradient GEMM inside the step (lwm_wgrad_bf16: dW = x^T g for every Dense kernel, lm_head
    # included): 2 S K N FLOPs per kernel over the kernel time the profiler saw for it in that extra step
    dense_kernel_params = sum(p.numel() for n, p in model.named_parameters() if p.dim() == 2 and not n.endswith("wte"))
    wg_ms = (out["roofline"]["time_shares"].get("kernel_ms") or {}).get("wgrad_gemm_hip")
    out["wgrad_gemm"] = {"kernel": "lwm::wgrad_bf16_kernel (+ wgrad_fixup_kernel)", "flops_per_step": 2.0 * S * dense_kernel_params,
                         "kernel_ms_per_step": wg_ms, "bound": "mfma", "peak": 2500.0, "unit": "TFLOP/s",
                         "achieved": None if not wg_ms else 2.0 * S * dense_kernel_params / (wg_ms * 1e-3) / 1e12,
                         "frac": None if not wg_ms else 2.0 * S * dense_kernel_params / (wg_ms * 1e-3) / 1e12 / 2500.0,
                         "algorithmic_bytes_per_call": "2 (S K + S N + K N): MFMA-bound at every shape of the step",
                         "enabled": os.environ.get("LWM_WGRAD_HIP", "1") == "1", "profile": "profiles/r06_wgrad.md"}
    del model
    torch.cuda.empty_cache()
    return out


def kernel_class(name):
    """A device kernel's class in the LWM-7B step (the table of profiles/r06_model_full.md)."""
    if "lwm::attn_" in name:
        return "attention_hip"
    if name.startswith(("Cijk_", "Custom_Cijk_")):
        return "library_gemm"
    if "lwm::wgrad_" in name:
        return "wgrad_gemm_hip"          # the weight gradients: lwm_wgrad_bf16 (csrc/gemm_wgrad.h)
    if "lwm::" in name:
        return "elementwise_hip"
    return "torch_elementwise_copy"


def model_time_shares(torch, step):
    """{class: ms of device kernel time in one step} + shares, measured live with torch.profiler (kineto over roctracer) on
    one extra step.  Skipped -- with the reason -- under rocprofv3 (two tracers in one process) or LWM_BENCH_NO_PROFILER=1;
    the committed rocprofv3 table of the same leg is profiles/r06_model_full_table.txt."""
    if os.environ.get("LWM_BENCH_NO_PROFILER") == "1":
        return {"skipped": "LWM_BENCH_NO_PROFILER=1"}
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {"skipped": "rocprofv3 is attached to this process"}
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        ms = {}
        for ev in prof.events():
            if getattr(ev, "device_type", None) is not None and "cuda" in str(ev.device_type).lower():
                dur = float(getattr(ev, "device_time_total", 0.0) or getattr(ev, "cuda_time_total", 0.0) or 0.0)
                if dur <= 0.0:
                    tr = getattr(ev, "time_range", None)
                    dur = float(tr.elapsed_us()) if tr is not None else 0.0
                ms[kernel_class(ev.name)] = ms.get(kernel_class(ev.name), 0.0) + dur / 1e3
        total = sum(ms.values())
        if total <= 0.0:
            return {"skipped": "torch.profiler recorded no device activity"}
        return {"source": "torch.profiler, one extra step (device kernels only)", "kernel_ms": {k: round(v, 2) for k, v in ms.items()},
                "share": {k: round(v / total, 4) for k, v in ms.items()}, "kernel_ms_total": round(total, 2)}
    except Exception as e:        # noqa: BLE001 -- a diagnostic; the leg's numbers stand without it
        return {"skipped": f"{type(e).__name__}: {str(e)[:200]}"}


def packed_1m_leg(torch, S=1 << 20, layers=2):
    """Optional leg (--packed-1m): BASELINE configs[4]'s problem on ONE GPU -- a 1,048,576-token batch of packed
    documents (log-uniform lengths in [4K, 256K]), attention forward+backward of `layers`
    layers; FLOPs counted over visible (same-document, causal) pairs only."""
    import numpy as np
    from lwm_amd.ring import HipBlockOps, SeqLayout, SingleComm, ring_backward, ring_forward
    rng = np.random.default_rng(0)
    seg = np.zeros((1, S), np.int32)
    pos, d, lens = 0, 0, []
    while pos < S:
        ln = min(int(np.exp(rng.uniform(np.log(4096), np.log(262144)))), S - pos)
        seg[:, pos:pos + ln] = d
        lens.append(ln)
        pos, d = pos + ln, d + 1
    segd = torch.from_numpy(seg).cuda()
    g = torch.Generator(device="cuda").manual_seed(2)
    mk = lambda: torch.randn(1, S, N_HEADS, HEAD_DIM, generator=g, device="cuda", dtype=torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    lay, comm = SeqLayout("contiguous", 1, S), SingleComm()

    def layer():
        out, lses = ring_forward(HipBlockOps, comm, q, k, v, layout=lay, causal=True, segment_ids=segd)
        ring_backward(HipBlockOps, comm, q, k, v, out, lses, do, layout=lay, causal=True, segment_ids=segd)

    layer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(layers):
        layer()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / layers
    flops = 7.0 * sum(float(l) * l for l in lens) * D_MODEL
    return {"workload": f"attention fwd+bwd, S={S}, {len(lens)} packed documents (lengths {min(lens)}..{max(lens)}), "
                        f"32 heads, per layer, one GPU [BASELINE configs[4] problem]",
            "s_per_layer": dt, "tokens_per_s_32_layers": S / (dt * N_LAYERS), "algorithmic_tflops": flops / dt / 1e12}


def decode_leg(torch, K=131072):
    """Secondary leg: one cached-decode attention step of LWM-7B (Q = 1, 32 heads)
    over a K-token KV cache resident in HBM -- ringattention_inference's path
    (lwm/llama.py:571-614).  HBM-bound: algorithmic bytes = the K and V cache read
    once = 2*K*4096*2 B (+ the (B,1,1,K) u8 mask)."""
    from lwm_amd import ops
    from lwm_amd.ring import _pick_splits
    g = torch.Generator(device="cuda").manual_seed(0)
    mk = lambda n: torch.randn(1, n, N_HEADS, HEAD_DIM, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    k, v, q = mk(K), mk(K), mk(1)
    mask = torch.ones(1, 1, K, dtype=torch.uint8, device="cuda")
    ns = _pick_splits(1, 1, N_HEADS, K)
    fn = lambda: ops.attn_combine(*ops.attn_fwd_splitk(q, k, v, k_splits=ns, dense_mask=mask))
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    nbytes = 2.0 * K * D_MODEL * 2 + K
    return {"workload": f"decode attention step, Q=1, cache K={K}, 32 heads x 128, bf16, k_splits={ns}",
            "us_per_layer_step": t * 1e6,
            "roofline": {"bound": "hbm", "unit": "GB/s", "achieved": nbytes / t / 1e9, "peak": 8000.0,
                         "frac": nbytes / t / 1e9 / 8000.0, "kernel": "attn_decode_kernel + attn_combine_kernel"}}


def generate_leg(torch, layers=4, prompt=2048, new=136, max_length=32768, short=8):
    """Secondary leg: greedy decoding through the KV cache on a `layers`-layer slice of LWM-7B (d_model 4096,
    32 heads, FFN 11008, vocab 32000): milliseconds per generated token with the one-token step issued
    kernel by kernel, and with the same step captured once in a hipGraph and replayed
    (LLaMAForCausalLM.generate(graph=True)).  The prefill and the first eight tokens are outside the timing."""
    import time as _t
    from lwm_amd.llama import LLaMAConfig, LLaMAForCausalLM
    cfg = LLaMAConfig.load_config("7b", num_hidden_layers=layers, max_sequence_length=max_length, theta=1e7)
    with torch.device("cuda"):
        model = LLaMAForCausalLM(cfg)
    ids = torch.randint(0, cfg.vocab_size, (1, prompt), device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    out = {}
    for name, graph in (("eager", False), ("hipgraph", True)):
        def run(n):
            torch.cuda.synchronize()
            t0 = _t.perf_counter()
            toks = model.generate(ids, max_new_tokens=n, max_length=max_length, graph=graph)
            torch.cuda.synchronize()
            return _t.perf_counter() - t0, toks
        run(3)                                   # warm
        # per-token time = (long run - short run) / extra tokens: prefill and graph capture cancel.  128 extra tokens:
        # at 0.5 ms per token a 32-token difference drowns in the run-to-run noise of the 2048-token prefill
        t_short = min(run(short)[0] for _ in range(2))
        t_long, toks = min((run(new) for _ in range(2)), key=lambda r: r[0])
        out[name + "_ms_per_token"] = (t_long - t_short) / (new - short) * 1e3
        out[name + "_tokens"] = toks[0, prompt:prompt + 8].tolist()
    out["same_tokens"] = out.pop("eager_tokens") == out.pop("hipgraph_tokens")
    out["workload"] = (f"greedy decode, {layers}-layer slice of LWM-7B, prompt {prompt}, cache max_length {max_length} "
                       f"(attention runs over the whole cache, lwm/llama.py:571-614), B=1, bf16")
    out["speedup"] = out["eager_ms_per_token"] / out["hipgraph_ms_per_token"]
    return out


def packed_documents(S, seed=0, lo_frac=256, hi_frac=4):
    """the synthetic packing of BASELINE configs[4]: document lengths log-uniform in
    [S / lo_frac, S / hi_frac] until S tokens; -> (segment ids (1, S) int32 numpy, lengths)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    seg = np.zeros((1, S), np.int32)
    pos, d, lens = 0, 0, []
    while pos < S:
        ln = min(int(np.exp(rng.uniform(np.log(S / lo_frac), np.log(S / hi_frac)))), S - pos)
        seg[:, pos:pos + ln] = d
        lens.append(ln)
        pos, d = pos + ln, d + 1
    return seg, lens


def ring_model_leg(torch, n=8, S=131072, schedule="mesh", driver="c", packed=False, layout="zigzag", reps=2):
    """What the compute side of an n-rank ring costs, measured on ONE GPU: every rank's launches (same shapes, offsets
    and masks; the exchange replaced by a transport that moves nothing) are run in turn for one layer.  The slowest
    rank bounds the job: tokens/s <= S / (max_r ms_per_layer * 32 layers).  What the 8-GPU run adds on top is exchange
    time that is not hidden (bench `exchange.exposed_ms_per_step` at N > 1).
    driver "c" = the product path on RCCL groups (lwm_ring_attn_fwd / _bwd; the direct schedule's gathered form where
    it applies: `form`); "python" = lwm_amd/ring.py's PER-PAIR launch list (one launch per segment pair: what rounds 1-4
    ran; LWM_RING_FORM=pairs keeps that driver from taking its own gathered form).
    packed: BASELINE configs[4]'s 15-document packing -- FLOPs are then counted over visible pairs only."""
    from lwm_amd.ring import HipBlockOps, SeqLayout, ring_backward, ring_forward
    from lwm_amd.ring_c import CRing
    seg, lens = None, None
    if packed:
        seg_np, lens = packed_documents(S)
        seg = torch.from_numpy(seg_np).cuda()
    if layout == "balanced":        # an ownership table from the document lengths (what a loader knows): 4 chunks per rank
        from lwm_amd.ring import balanced_layout
        lay = balanced_layout(n, S, lens, chunks_per_rank=4)
    else:
        lay = SeqLayout(layout, n, S)
    c = lay.local_len
    g = torch.Generator(device="cuda").manual_seed(4321)
    mk = lambda: torch.randn(1, c, N_HEADS, HEAD_DIM, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    per_rank, forms = [], []
    for r in range(n):
        if driver == "c":
            ring = CRing.null(r, n, layout=lay if lay.kind == "table" else layout, schedule="direct" if schedule in ("mesh", "direct") else "ring")

            def layer():
                o, l = ring.forward(q, k, v, causal=True, segment_ids=seg)
                ring.backward(q, k, v, o, l, do, causal=True, segment_ids=seg)
        else:
            comm = NullComm(rank=r, size=n, schedule=schedule)

            def layer():
                keep = os.environ.get("LWM_RING_FORM")
                os.environ["LWM_RING_FORM"] = "pairs"
                try:
                    out, lses = ring_forward(HipBlockOps, comm, q, k, v, layout=lay, causal=True, segment_ids=seg)
                    ring_backward(HipBlockOps, comm, q, k, v, out, lses, do, layout=lay, causal=True, segment_ids=seg)
                finally:
                    os.environ.pop("LWM_RING_FORM") if keep is None else os.environ.__setitem__("LWM_RING_FORM", keep)

        layer()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            layer()
        e1.record()
        torch.cuda.synchronize()
        per_rank.append(e0.elapsed_time(e1) / reps)
        if driver == "c":
            forms.append(ring.last_form)
            ring.close()
    worst = max(per_rank)
    flops_layer = 7.0 * (gemm_unit_flops(S) if not packed else sum(gemm_unit_flops(l) for l in lens))
    return {"workload": f"compute side of an {n}-rank {layout} ring at S={S} ({schedule} schedule, {driver} driver"
                        + (f", {len(lens)} packed documents" if packed else "") + "), each rank's launches run on this GPU, "
                        "1 layer, no exchange",
            "driver": driver, "form": ("per pair" if driver != "c" else "gathered" if all(forms) else "per pair" if not any(forms) else "mixed"),
            "per_rank_ms_per_layer": [round(x, 3) for x in per_rank],
            "imbalance_max_over_mean": worst / (sum(per_rank) / n),
            "compute_bound_tokens_per_s": S / (worst * 1e-3 * N_LAYERS),
            "compute_bound_tflops_per_gpu": flops_layer / n / (worst * 1e-3) / 1e12}


XGMI_LINK_GBPS = 153.0          # one xGMI link, one direction (MI355X_MICROARCH.md); 7 links per GPU, point to point
XGMI_LINK_EFFICIENCY = 0.75     # what a large send/recv is assumed to reach of that -- an ASSUMPTION, stated in the line


def predicted_scaling_leg(torch, known=None):
    """A MODEL of the 1 -> 8 curve from what ONE GPU can measure -- labelled as such, never a measured scaling number:
      compute: every rank's launch list of the product path (C driver, zigzag ownership, direct schedule = the N > 1
               default) run on this GPU with a transport that moves nothing (ring_model_leg); the slowest rank counts;
               N = 1 is one directly timed layer (fwd+bwd) at the same S;
      link:    the bytes the slowest-sending rank posts per layer (lwm_ring_planned_bytes, fwd + bwd; the backward's K/V
               re-fetch is dropped when the gathered K/V is kept, LWM_RING_KEEP_KV_MB), spread over min(N - 1, 7) xGMI
               links at XGMI_LINK_GBPS x XGMI_LINK_EFFICIENCY;
      predicted tokens/s = S / (32 layers x max(compute, link))  [exchange fully hidden]  and  S / (32 x (compute + link))
               [nothing hidden]; efficiency = that / (N x the N = 1 figure): strong scaling at fixed S, BASELINE's metric.
    S = 32768 is what `--gpus N` times (configs[1] at every N), S = 131072 is configs[2]."""
    from lwm_amd import _capi
    from lwm_amd._lib import lib
    from lwm_amd.ring import HipBlockOps, SeqLayout, SingleComm, ring_backward, ring_forward
    L = lib()
    known = known or {}
    out = {"kind": "model, not measured", "link_GBps_assumed": XGMI_LINK_GBPS * XGMI_LINK_EFFICIENCY,
           "link_note": f"{XGMI_LINK_GBPS:.0f} GB/s per xGMI link and direction x {XGMI_LINK_EFFICIENCY} assumed efficiency, "
                        "min(N-1, 7) links per rank", "by_S": {}}
    for S in (32768, 131072):
        g = torch.Generator(device="cuda").manual_seed(99)
        mk = lambda: torch.randn(1, S, N_HEADS, HEAD_DIM, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
        q, k, v, do = mk(), mk(), mk(), mk()
        lay, comm = SeqLayout("contiguous", 1, S), SingleComm()

        def layer():
            o, lses = ring_forward(HipBlockOps, comm, q, k, v, layout=lay, causal=True)
            ring_backward(HipBlockOps, comm, q, k, v, o, lses, do, layout=lay, causal=True)

        layer()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 if S <= 32768 else 2
        e0.record()
        for _ in range(reps):
            layer()
        e1.record()
        torch.cuda.synchronize()
        ms1 = e0.elapsed_time(e1) / reps
        del q, k, v, do
        torch.cuda.empty_cache()
        base = S / (ms1 * 1e-3 * N_LAYERS)
        rows = {"1": {"compute_ms_per_layer": round(ms1, 3), "source": "one layer fwd+bwd timed directly on this GPU",
                      "link_ms_per_layer": 0.0, "bytes_sent_per_rank_per_layer": 0, "predicted_tokens_per_s": base,
                      "predicted_tokens_per_s_nothing_hidden": base, "efficiency": 1.0,
                      "tflops_per_gpu": 7.0 * gemm_unit_flops(S) / (ms1 * 1e-3) / 1e12}}
        for n in (2, 4, 8):
            m = known.get((n, S)) or ring_model_leg(torch, n=n, S=S, reps=3 if S <= 32768 else 2)
            comp = max(m["per_rank_ms_per_layer"])
            c = S // n
            lay_code, sched = _capi.RING_LAYOUT["zigzag"], _capi.RING_SCHEDULE["direct"]
            sent = [int(L.lwm_ring_planned_bytes(lay_code, sched, n, r, 1, c, N_HEADS, HEAD_DIM, 1, 0)) +
                    int(L.lwm_ring_planned_bytes(lay_code, sched, n, r, 1, c, N_HEADS, HEAD_DIM, 1, 1)) for r in range(n)]
            kv_again = [int(L.lwm_ring_planned_bytes(lay_code, sched, n, r, 1, c, N_HEADS, HEAD_DIM, 1, 0)) for r in range(n)]
            keep_ok = int(L.lwm_ring_kv_keep_bytes(1, c, N_HEADS, HEAD_DIM, n)) <= float(os.environ.get("LWM_RING_KEEP_KV_MB", "1024")) * (1 << 20)
            if keep_ok:          # the backward reads the kept K/V instead of fetching it again
                sent = [a - b for a, b in zip(sent, kv_again)]
            link = max(sent) / (min(n - 1, 7) * XGMI_LINK_GBPS * 1e9 * XGMI_LINK_EFFICIENCY) * 1e3
            hid, exposed = S / (max(comp, link) * 1e-3 * N_LAYERS), S / ((comp + link) * 1e-3 * N_LAYERS)
            rows[str(n)] = {"compute_ms_per_layer": round(comp, 3), "imbalance_max_over_mean": round(m["imbalance_max_over_mean"], 4),
                            "source": "slowest rank of the product's launch list, all ranks run on this GPU, no exchange",
                            "bytes_sent_per_rank_per_layer": max(sent), "kv_kept_for_backward": bool(keep_ok),
                            "link_ms_per_layer": round(link, 3), "bound": "compute" if comp >= link else "link",
                            "predicted_tokens_per_s": hid, "predicted_tokens_per_s_nothing_hidden": exposed,
                            "efficiency": hid / (n * base), "efficiency_nothing_hidden": exposed / (n * base),
                            "tflops_per_gpu": 7.0 * gemm_unit_flops(S) / n / (comp * 1e-3) / 1e12}
        out["by_S"][str(S)] = rows
    return out


def elementwise_leg(torch, S=32768):
    """Secondary leg: RoPE (q and k) and RMSNorm fwd at LWM-7B shapes, HBM-bound.
    Algorithmic bytes: RoPE 2 tensors x (read + write) x S*4096*2 B (+ the table);
    RMSNorm read + write S*4096*2 B."""
    from lwm_amd.llama_ops import RMSNorm, apply_rotary_emb, precompute_freqs_cis
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(1, S, N_HEADS, HEAD_DIM, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    tab = precompute_freqs_cis(HEAD_DIM, S, 1e7, device="cuda")
    pos = torch.arange(S, device="cuda", dtype=torch.int32)[None].contiguous()
    norm = RMSNorm(D_MODEL).cuda()
    h = x.reshape(S, D_MODEL)

    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    with torch.no_grad():
        t_rope = timed(lambda: apply_rotary_emb(x, x, tab, pos))
        t_norm = timed(lambda: norm(h))
    n = S * D_MODEL * 2.0
    return {"workload": f"RoPE(q,k) and RMSNorm forward, S={S}, d_model=4096, bf16",
            "rope_GBps": (4 * n + S * 512.0) / t_rope / 1e9, "rmsnorm_GBps": 2 * n / t_norm / 1e9,
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                         "achieved": 2 * n / t_norm / 1e9, "frac": 2 * n / t_norm / 1e9 / 8000.0,
                         "kernel": "rmsnorm_fwd_kernel"}}


class NullComm:
    """Same rank/size/schedule as the real communicator, but nothing moves: rotate() hands the
    local tensors back and exchange_async() leaves the receive buffers as allocated.  Used once,
    after the timed region, to price the exchange (bench `exchange` object), and by ring_model_leg
    to run any rank's launches of an N-rank ring on one GPU."""

    def __init__(self, real=None, *, rank=0, size=1, schedule="mesh"):
        if real is not None:
            rank, size, schedule = real.rank, real.size, real.schedule
        self.rank, self.size, self.schedule = rank, size, schedule

    class _H:
        def __init__(self, bufs):
            self.bufs = bufs

        def wait(self):
            return self.bufs

    def rotate(self, tensors):
        return NullComm._H(list(tensors))

    def exchange_async(self, sends, recvs):
        return NullComm._H([t for _, t in recvs])


def HostStagedComm(torch, dist, Base, schedule):
    """--backend gloo (dry run): gloo moves host memory, so every message is staged device -> host
    -> peer -> device.  Only the control flow of the N > 1 path is being exercised."""

    class _H:
        def __init__(self, reqs, pairs):
            self.reqs, self.pairs = reqs, pairs

        def wait(self):
            for r in self.reqs:
                r.wait()
            self.reqs = []
            for dst, host in self.pairs:
                dst.copy_(host)
            self.pairs = []
            return getattr(self, "bufs", None)

    class Staged(Base):
        def exchange_async(self, sends, recvs):
            ops_ = [dist.P2POp(dist.isend, t.cpu(), peer) for peer, t in sends]
            pairs = []
            for peer, buf in recvs:
                host = torch.empty(buf.shape, dtype=buf.dtype)
                pairs.append((buf, host))
                ops_.append(dist.P2POp(dist.irecv, host, peer))
            h = _H(dist.batch_isend_irecv(ops_) if ops_ else [], pairs)
            h.bufs = [b for _, b in recvs]
            h.keep = [op.tensor for op in ops_]     # host copies stay alive until wait()
            return h

        def rotate(self, tensors):
            bufs = [torch.empty_like(t) for t in tensors]
            return self.exchange_async([((self.rank + 1) % self.size, t) for t in tensors],
                                       [((self.rank - 1) % self.size, b) for b in bufs])

    return Staged(None, schedule=schedule)


class KernelTimer:
    """HIP events (torch.cuda.Event on the stream the kernels are launched on)
    around every kernel launch of the timed region, aggregated per kernel."""

    def __init__(self, torch):
        self.torch = torch
        self.spans = {}
        self.enabled = False

    def run(self, name, fn, *a, **kw):
        if not self.enabled:
            return fn(*a, **kw)
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        self.spans.setdefault(name, []).append((e0, e1))
        return r

    def summary(self):
        out = {}
        for name, evs in self.spans.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms)}
        return out


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sck:
        sck.bind(("127.0.0.1", 0))
        return sck.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: become the launcher --
    N ranks of this same script under torch.distributed.run on 127.0.0.1 -- and hand back its exit code.
    The children see RANK / WORLD_SIZE and therefore never come here."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL peer mappings need it on this driver
    env.setdefault("NCCL_DEBUG", "WARN")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    env["LWM_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print("[bench] self-launch: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def fail_line(args, stage, err, extra=None):
    """A diagnosable line instead of a hang or a bare traceback: the driver's JSON parser gets
    {"error": ...} with the stage that failed and what this rank could see."""
    import traceback
    rec = {"metric": "tokens/sec fwd+bwd, LWM-7B RingAttention hot path (32 layers x 32 heads x 128)",
           "value": None, "unit": "tokens/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
           "error": {"stage": stage, "type": type(err).__name__, "message": str(err)[:2000],
                     "rank": int(os.environ.get("RANK", "0")), "world_size": int(os.environ.get("WORLD_SIZE", "1")),
                     "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                     "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "HIP_VISIBLE_DEVICES",
                                                            "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}}}
    if extra:
        rec["error"].update(extra)
    print(json.dumps(rec), flush=True)
    traceback.print_exception(type(err), err, err.__traceback__, file=sys.stderr)


class Watchdog:
    """A hang becomes a diagnosable line.  The ring exchange has met RCCL on one GPU and real processes over the IPC
    transport, never several GPUs: a stuck peer exchange there would block every rank inside a stream wait or a collective,
    where no exception can reach it.  While a stage is armed, a timer thread that finds it unfinished after `seconds` prints
    the {"error": ...} line (rank 0 on stdout, the others on stderr only) and leaves the process without waiting for the
    GPU -- the launcher then ends the job instead of the driver's clock."""

    def __init__(self, args, seconds):
        self.args, self.seconds, self.limit, self.timer, self.stage, self.on_fire = args, seconds, seconds, None, None, None

    def arm(self, stage, seconds=None, on_fire=None):
        """on_fire(stage, error) -> exit code replaces the error line (a stage after which the line's value exists)"""
        import threading
        self.disarm()
        self.stage, self.on_fire, self.limit = stage, on_fire, seconds or self.seconds
        self.timer = threading.Timer(self.limit, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
        self.timer = None

    def _fire(self):
        err = TimeoutError(f"stage '{self.stage}' did not finish within {self.limit:.0f} s on this rank "
                           f"(LWM_BENCH_WATCHDOG_S sets the limit; --driver python takes the torch.distributed ring)")
        if self.on_fire is not None:
            code = 7
            try:
                code = self.on_fire(self.stage, err)
                sys.stdout.flush()
            finally:
                os._exit(code)
        if int(os.environ.get("RANK", "0")) == 0:
            fail_line(self.args, "watchdog: " + self.stage, err)
        else:
            print(f"[bench] rank {os.environ.get('RANK')}: {err}", file=sys.stderr, flush=True)
        os._exit(7)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seq", type=int, default=0, help="override global sequence length")
    ap.add_argument("--layers", type=int, default=N_LAYERS, help="(debug only; default = full 32)")
    ap.add_argument("--layout", default="zigzag", choices=["zigzag", "contiguous"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vqgan", action="store_true", help="skip the secondary VQGAN leg")
    ap.add_argument("--schedule", default=None, choices=["ring", "mesh"],
                    help="K/V exchange schedule for N > 1 (default: mesh; lwm_amd/ring.py)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = DRY RUN of the N > 1 code path on a box with fewer than N GPUs: the ranks "
                         "share the visible devices and stage every message through host memory")
    ap.add_argument("--packed", action="store_true",
                    help="masked sequence packing (BASELINE config #5 style): documents log-uniform in "
                         "[S/256, S/4]; FLOPs are counted over visible pairs only")
    ap.add_argument("--driver", default=None, choices=["c", "python"],
                    help="N > 1: who drives the exchange.  c (default on GPUs) = the C-ABI ring driver (lwm_ring_attn_fwd/bwd: "
                         "RCCL send/recv or the IPC transport on a side HIP stream, zigzag / contiguous ownership, ring / direct "
                         "schedule); python = lwm_amd/ring.py over torch.distributed (the only choice for --backend gloo without "
                         "--transport ipc)")
    ap.add_argument("--c-ring", action="store_true", help="same as --driver c")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "ipc"],
                    help="C driver only: rccl = ncclSend/ncclRecv; ipc = the library's CU-free transport (peer mailboxes mapped "
                         "through hipIpcMemHandles + hipMemcpyAsync + stream memory operations).  With --backend gloo the ranks may "
                         "share GPUs: `--gpus 4 --backend gloo --transport ipc` runs the whole C-driver path on a 1-GPU box")
    ap.add_argument("--no-configs2", action="store_true", help="N > 1: skip the S = 131072 leg (BASELINE configs[2]) of the line")
    ap.add_argument("--init-timeout", type=int, default=180,
                    help="seconds before a stuck RCCL rendezvous / first collective is reported as an error line")
    ap.add_argument("--no-full-model", action="store_true", help="skip the N=1 leg `model_full` (all 32 layers of LWM-7B)")
    ap.add_argument("--packed-1m", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--no-packed-1m", action="store_true",
                    help="N = 1: skip the 1M-token packed-documents leg (BASELINE configs[4]'s problem on one GPU, ~25 s)")
    ap.add_argument("--force-configs34", action="store_true",
                    help="(functional check) run the configs3 / configs4 legs even when the ranks share a GPU (--backend gloo "
                         "--transport ipc): the IPC mailboxes are then sized for c = 1048576 / N")
    ap.add_argument("--no-transport-trials", action="store_true",
                    help="N > 1: skip the last secondary leg (a few layers under every transport / schedule pair of the C driver)")
    ap.add_argument("--no-configs34", action="store_true",
                    help="N > 1: skip the S = 262144 (BASELINE configs[3]) and packed S = 1048576 (configs[4]) legs of the line")
    args = ap.parse_args()

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from lwm_amd import ops
    from lwm_amd.ring import (HipBlockOps, SeqLayout, SingleComm, TorchRingComm, ring_backward,
                              ring_forward)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus "
                         f"must agree (plain `python bench.py --gpus {args.gpus}` launches the ranks itself)")
    shared = args.backend == "gloo"             # the ranks may share devices
    dry = shared and args.transport != "ipc"    # ... and stage their messages through host memory (Python driver)
    if args.c_ring:
        args.driver = "c"
    if args.driver is None:
        args.driver = "python" if dry else "c"
    if args.driver == "c" and dry:
        raise SystemExit("--driver c needs a GPU transport: --backend nccl, or --backend gloo --transport ipc")
    n_dev = torch.cuda.device_count()
    if n_dev == 0 or (not shared and local_rank >= n_dev):
        fail_line(args, "devices", RuntimeError(f"rank {rank} (local {local_rank}) sees {n_dev} GPU(s); "
                                                f"--gpus {args.gpus} needs one GPU per rank (--backend gloo is the dry run)"))
        sys.exit(3)
    dev_index = local_rank % n_dev if shared else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rccl_ranks_seen = ranks_seen = None
    if world > 1 and int(os.environ.get("LWM_RING_RESERVE_CUS", "0") or 0) > 0:
        # the attention kernels on a stream that leaves k CUs alone (for RCCL's send/recv kernels): priced in this line
        from lwm_amd.ring_c import reserved_cu_stream
        torch.cuda.set_stream(reserved_cu_stream(int(os.environ["LWM_RING_RESERVE_CUS"]), dev))
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")     # a stuck collective aborts, it does not hang
        tmo = datetime.timedelta(seconds=args.init_timeout)
        try:
            if dry:
                dist.init_process_group("gloo", timeout=tmo)
                comm = HostStagedComm(torch, dist, TorchRingComm, args.schedule)
            elif shared:
                dist.init_process_group("gloo", timeout=tmo)      # bootstrap only: the data moves through the IPC transport
                comm = None
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)
                comm = TorchRingComm(dist.group.WORLD, schedule=args.schedule)
            # first contact: every rank contributes 1 through the backend that will carry the exchange
            # (RCCL over xGMI, or gloo in the dry run) and one neighbour send/recv goes round the ring
            one = torch.ones(1, dtype=torch.float32, device="cpu" if shared else dev)
            dist.all_reduce(one)
            ranks_seen = int(one.item())
            # RCCL saw these ranks only when the backend IS RCCL; a gloo bootstrap reports under its own name, so that a
            # dry run cannot be mistaken for the first real multi-GPU line
            rccl_ranks_seen = ranks_seen if dist.get_backend() == "nccl" else None
            if not shared:
                tok = torch.full((1,), float(rank), device=dev)
                got = torch.empty(1, device=dev)
                for r_ in dist.batch_isend_irecv([dist.P2POp(dist.isend, tok, (rank + 1) % world),
                                                  dist.P2POp(dist.irecv, got, (rank - 1) % world)]):
                    r_.wait()
                torch.cuda.synchronize()
                if int(got.item()) != (rank - 1) % world:
                    raise RuntimeError(f"ring send/recv probe: expected {(rank - 1) % world}, received {got.item()}")
        except Exception as e:
            fail_line(args, "init_process_group / first collective", e, {"backend": args.backend})
            os._exit(4)          # do not wait on a process group that never formed
    else:
        comm = SingleComm()

    S = args.seq or 32768        # the SAME problem at every N (strong scaling); --seq 131072 = BASELINE configs[2]
    layout = SeqLayout(args.layout, world, S)
    c = layout.local_len
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    mk = lambda: torch.randn(1, c, N_HEADS, HEAD_DIM, generator=g, device=dev,
                             dtype=torch.float32).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    timer = KernelTimer(torch)
    segment_ids, doc_sq = None, None
    if args.packed:
        import numpy as np
        rng = np.random.default_rng(0)
        seg = np.zeros((1, S), np.int32)
        pos, d, lens = 0, 0, []
        while pos < S:
            ln = int(np.exp(rng.uniform(np.log(S / 256), np.log(S / 4))))
            ln = min(ln, S - pos)
            seg[:, pos:pos + ln] = d
            lens.append(ln)
            pos, d = pos + ln, d + 1
        segment_ids = torch.from_numpy(seg).to(dev)
        doc_sq = float(sum(l * l for l in lens))

    class TimedOps(HipBlockOps):
        fwd = staticmethod(lambda *a, **kw: timer.run("attn_fwd64_kernel", ops.attn_fwd_block, *a, **kw))
        bwd_delta = staticmethod(lambda *a, **kw: timer.run("attn_bwd_delta_kernel", ops.attn_bwd_delta, *a, **kw))
        bwd_dq = staticmethod(lambda *a, **kw: timer.run("attn_bwd_dq4_kernel", ops.attn_bwd_dq_block, *a, **kw))
        bwd_dkdv = staticmethod(lambda *a, **kw: timer.run("attn_bwd_dkdv4_kernel", ops.attn_bwd_dkdv_block, *a, **kw))

    c_ring = None
    sched_c = "ring" if args.schedule == "ring" else "direct"
    dog = Watchdog(args, float(os.environ.get("LWM_BENCH_WATCHDOG_S", "600")))
    if world > 1:
        dog.arm(f"ring set-up and first layer ({args.driver} driver, {args.backend})")
    S2 = 131072                     # BASELINE configs[2]'s sequence: a second leg of every N > 1 line
    if args.driver == "c" and world > 1:
        from lwm_amd.ring_c import CRing
        c_max = max(c, S2 // world if not args.no_configs2 else 0)      # (configs3 / configs4 do not run over the IPC transport ...
        if args.force_configs34:                                       #  ... unless asked to)
            c_max = max(c_max, (1 << 20) // world)
        # The C driver has met RCCL on one GPU and real processes over IPC, never several GPUs: its set-up (every collective
        # of CRing's bootstrap is entered by every rank whatever failed on it before) and its first layer run under a
        # collective vote, and a rank that fails takes everybody to the torch.distributed driver instead of killing the line.
        ok, why = 1, ""
        try:
            CRing.probe(args.transport)
        except Exception as e:      # noqa: BLE001
            ok, why = 0, repr(e)[:500]
        vote = torch.tensor([ok], dtype=torch.int32, device="cpu" if shared else dev)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN)
        if int(vote.item()) == 1:
            try:
                c_ring = CRing(dist.group.WORLD, transport=args.transport, layout=args.layout, schedule=sched_c,
                               ipc_slot_bytes=c_max * N_HEADS * HEAD_DIM * 4, ipc_slots=8)      # (B = 1: 4 messages per pair and group; 8 with the 4-chunk ownership table of the packed leg)
            except Exception as e:      # noqa: BLE001
                ok, why = 0, repr(e)[:500]
        ring_setup_failed = why if not ok else None

    driver_fallback = None
    if args.driver == "c" and world > 1:
        ok, why = (1, "") if c_ring is not None else (0, ring_setup_failed or "the ring could not be set up on another rank")
        try:
            if c_ring is None:
                raise RuntimeError(why)
            o_, l_ = c_ring.forward(q, k, v, causal=True, segment_ids=segment_ids)
            c_ring.backward(q, k, v, o_, l_, do, causal=True, segment_ids=segment_ids)
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            ok, why = 0, repr(e)[:500]
        vote = torch.tensor([ok], dtype=torch.int32, device="cpu" if shared else dev)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN)
        if int(vote.item()) == 0:
            if comm is None:
                fail_line(args, "C ring driver, first layer", RuntimeError(why or "another rank failed"), {"transport": args.transport})
                os._exit(5)
            driver_fallback = {"from": f"C driver ({args.transport}, {sched_c})", "to": "lwm_amd/ring.py over torch.distributed",
                               "first_error_on_this_rank": why or None,
                               "consequence": "the per-pair Python driver launches one kernel per segment pair instead of the C driver's "
                                              "gathered form: 709 against 824-849 TF/s per GPU on the compute side of 8 ranks at "
                                              "S = 32768 (ring8_compute_model_32k_per_pair / ring8_compute_model_32k in the N = 1 line, "
                                              "profiles/r05_ring_shards.md) -- this line's value is NOT the product's default path"}
            if c_ring is not None:
                c_ring.close()
            c_ring = None

    def step(ring=None, ten=None, lay=None, seg=None, layers=None):
        ring = c_ring if ring is None else ring
        q_, k_, v_, do_ = (q, k, v, do) if ten is None else ten
        lay = layout if lay is None else lay
        seg = segment_ids if ten is None else seg
        for _ in range(args.layers if layers is None else layers):
            if ring is not None:
                tab = lay if lay.kind == "table" else None      # (an ownership table travels with the call)
                o_, l_ = ring.forward(q_, k_, v_, causal=True, segment_ids=seg, layout=tab)
                ring.backward(q_, k_, v_, o_, l_, do_, causal=True, segment_ids=seg, layout=tab)
                continue
            out, lses = ring_forward(TimedOps, comm, q_, k_, v_, layout=lay, causal=True, segment_ids=seg)
            ring_backward(TimedOps, comm, q_, k_, v_, out, lses, do_, layout=lay, causal=True, segment_ids=seg)

    def null_ring(schedule=None):
        """the same C driver with a transport that moves nothing (buffers left as allocated): what the step costs when
        every transfer is free"""
        from lwm_amd import _capi
        from lwm_amd.ring_c import CRing
        ok = lambda *a: 0
        t = _capi.LwmRingTransport(None, _capi.RING_GROUP_FN(ok), _capi.RING_SEND_FN(ok), _capi.RING_SEND_FN(ok), _capi.RING_GROUP_FN(ok))
        ring = CRing(rank=rank, size=world, transport=t, layout=args.layout, schedule=schedule or sched_c)
        ring._null = True           # its "received" K/V is N(0,1) bf16, not whatever the allocation held (see CRing.null)
        return ring

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if shared else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if world > 1:
        dog.arm("warm-up and timed steps")
    def all_ranks_ok(ok):
        """a collective every rank enters whatever happened on it before: one rank's failure (an allocation, say) must
        skip a leg everywhere instead of leaving the others inside the leg's exchange"""
        if world == 1:
            return bool(ok)
        vote = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if shared else dev)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN)
        return int(vote.item()) == 1

    for _ in range(args.warmup):
        step()
    barrier()
    timer.enabled = world == 1
    sent_before = c_ring.bytes_sent if c_ring is not None else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    dog.disarm()
    sent_timed = (c_ring.bytes_sent - sent_before) if c_ring is not None else 0
    exchange = configs2 = configs3 = configs4 = transport_trials = None

    def main_line():
        """the line as far as it is known: the timed region is over when this is first called; the secondary
        configurations of an N > 1 run are whatever they are at that moment (the watchdog may call it early)"""
        ms_per_step = elapsed * 1e3 / args.steps
        tokens_per_s = S * args.steps / elapsed
        unit = gemm_unit_flops(S) if doc_sq is None else doc_sq * D_MODEL   # visible pairs only when packed
        algo_flops_step = 7.0 * unit * args.layers
        res = {
            "metric": "tokens/sec fwd+bwd, LWM-7B RingAttention hot path (32 layers x 32 heads x 128)",
            "value": tokens_per_s,
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16 (f32 logits/softmax/accumulate)",
            "data": "synthetic N(0,1) q/k/v/dO, seed 1234, resident in HBM",
            "config": {
                "workload": (f"LWM-7B attention fwd+bwd, {args.layers} layers, B=1, S={S}, H=32, D=128, "
                             f"causal, ring={world}" + (f", layout={layout.kind}" if world > 1 else "")
                             + (f", masked packing ({len(lens)} documents)" if args.packed else "")
                             + (" [BASELINE configs[1]]" if world == 1 and S == 32768 and not args.packed else "")
                             + (" [BASELINE configs[2] problem]" if world > 1 and S == 131072 else "")
                             + (" [BASELINE configs[1] problem, ring-sharded]" if world > 1 and S == 32768 else "")),
                "seq_len": S, "ring": world, "layers": args.layers,
                "exchange_schedule": (f"{sched_c} (C-ABI driver, {args.transport} on a side stream)" if c_ring is not None else
                                      getattr(comm, "schedule", None)) if world > 1 else None,
            },
            "tokens_per_s_per_gpu": tokens_per_s / world,
            "exchange": exchange,
            "transport_trials": transport_trials,
            "configs2": configs2,
            "configs3": configs3,
            "configs4": configs4,
            "driver_fallback": driver_fallback,
            "rccl_ranks_seen": rccl_ranks_seen,
            "bootstrap_ranks_seen": ranks_seen if world > 1 else None,
            "bootstrap_backend": dist.get_backend() if world > 1 else None,
            "dry_run": ("ranks share devices, messages staged through host memory; timings are not xGMI" if dry else
                        "ranks share devices (IPC transport inside one GPU); timings are not xGMI" if shared and world > 1 else None),
            "path_algorithmic_tflops_per_gpu": algo_flops_step / (ms_per_step * 1e-3) / 1e12 / world,
        }
        return res

    if world > 1:
        # From here on the line's value exists: a secondary configuration that hangs must not cost it.  Rank 0 prints the
        # line as far as it is known and every rank leaves with code 0.
        def late(stage, err):
            if rank == 0:
                print(json.dumps(dict(main_line(), secondary_legs_error={"stage": stage, "message": str(err)})), flush=True)
            return 0
        dog.arm("secondary configurations of the N > 1 line (configs2 / configs3 / configs4)",
                float(os.environ.get("LWM_BENCH_WATCHDOG_LATE_S", "900")), on_fire=late)


    exchange = None
    if world > 1:
        try:
            # the same launches with the exchange removed (buffers left as allocated): what the step
            # would cost if every transfer were free.  exposed = what the xGMI traffic adds on top.
            if c_ring is None:
                real_comm, comm = comm, NullComm(comm)
                step()
                barrier()
                t0 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                compute_only = max_over_ranks(time.perf_counter() - t0)
                comm = real_comm
                exchange = {"schedule": comm.schedule, "compute_only_ms_per_step": compute_only * 1e3,
                            "exposed_ms_per_step": (elapsed / args.steps - compute_only) * 1e3,
                            "overlap_efficiency": compute_only / (elapsed / args.steps)}
            else:
                nr = null_ring()
                step(nr)
                barrier()
                t0 = time.perf_counter()
                step(nr)
                torch.cuda.synchronize()
                compute_only = max_over_ranks(time.perf_counter() - t0)
                nr.close()
                exchange = {"schedule": f"{sched_c} (C-ABI driver, {args.transport})", "transport": args.transport,
                            "bytes_sent_per_rank_per_step": sent_timed / args.steps,
                            "compute_only_ms_per_step": compute_only * 1e3,
                            "exposed_ms_per_step": (elapsed / args.steps - compute_only) * 1e3,
                            "overlap_efficiency": compute_only / (elapsed / args.steps)}
            # The N=1 line of this bench is BASELINE configs[1] (S=32768); attention cost is quadratic
            # in S, so tokens/s at different S do not compare.  For a like-for-like strong-scaling
            # figure every rank also times ONE layer of THIS problem (same S) on its GPU alone.
            del q, k, v, do
            lay1 = SeqLayout("contiguous", 1, S)
            g1 = torch.Generator(device=dev).manual_seed(99)
            mk1 = lambda: torch.randn(1, S, N_HEADS, HEAD_DIM, generator=g1, device=dev,
                                      dtype=torch.float32).to(torch.bfloat16)
            q1, k1, v1, do1 = mk1(), mk1(), mk1(), mk1()

            def one_layer():
                o, l = ring_forward(HipBlockOps, SingleComm(), q1, k1, v1, layout=lay1, causal=True,
                                    segment_ids=segment_ids)
                ring_backward(HipBlockOps, SingleComm(), q1, k1, v1, o, l, do1, layout=lay1, causal=True,
                              segment_ids=segment_ids)

            one_layer()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            one_layer()
            torch.cuda.synchronize()
            t_layer = max_over_ranks(time.perf_counter() - t0)
            one_gpu_tps = S / (t_layer * args.layers)
            exchange["same_problem_on_1_gpu"] = {
                "ms_per_layer": t_layer * 1e3, "tokens_per_s": one_gpu_tps,
                "speedup": (S * args.steps / elapsed) / one_gpu_tps,
                "strong_scaling_efficiency": (S * args.steps / elapsed) / one_gpu_tps / world,
                "note": "1 layer of the same problem, ring=1, timed on every rank after the timed region "
                        "(max over ranks), x layers"}
        except Exception as e:      # the main line must still be printed
            exchange = dict(exchange or {}, error=repr(e))

    configs2 = None
    if world > 1 and not args.no_configs2 and S != S2 and not args.packed:
        # BASELINE configs[2] (S = 131072 over the ring) at THIS N, in the same line: `value` stays the S = 32768 strong-scaling
        # series, this object is the configuration the metric names.
        err2, ten2 = None, None
        try:
            lay2 = SeqLayout(args.layout, world, S2)
            c2 = lay2.local_len
            g2 = torch.Generator(device=dev).manual_seed(4321 + rank)
            ten2 = [torch.randn(1, c2, N_HEADS, HEAD_DIM, generator=g2, device=dev, dtype=torch.float32).to(torch.bfloat16)
                    for _ in range(4)]
        except Exception as e:      # noqa: BLE001
            err2 = repr(e)[:500]
        if not all_ranks_ok(err2 is None):
            ten2 = None
            configs2 = {"error": err2 or "the set-up failed on another rank", "skipped_on_every_rank": True}
    if world > 1 and not args.no_configs2 and S != S2 and not args.packed and configs2 is None:
        try:
            sent0 = c_ring.bytes_sent if c_ring is not None else None
            step(ten=ten2, lay=lay2)
            barrier()
            t0 = time.perf_counter()
            n2 = 2
            for _ in range(n2):
                step(ten=ten2, lay=lay2)
            barrier()
            el2 = max_over_ranks(time.perf_counter() - t0)
            configs2 = {
                "workload": f"LWM-7B attention fwd+bwd, {args.layers} layers, B=1, S={S2}, H=32, D=128, causal, ring={world}, "
                            f"layout={lay2.kind} [BASELINE configs[2]" + ("" if world == 8 else f" at N={world}") + "]",
                "seq_len": S2, "steps": n2, "ms_per_step": el2 * 1e3 / n2, "tokens_per_s": S2 * n2 / el2,
                "tokens_per_s_per_gpu": S2 * n2 / el2 / world,
                "path_algorithmic_tflops_per_gpu": 7.0 * gemm_unit_flops(S2) * args.layers / (el2 / n2) / 1e12 / world,
                "exchange": ({"schedule": f"{sched_c} (C-ABI driver, {args.transport})",
                              "bytes_sent_per_rank_per_step": (c_ring.bytes_sent - sent0) / (n2 + 1)} if c_ring is not None
                             else {"schedule": getattr(comm, "schedule", None)}),
            }
        except Exception as e:      # noqa: BLE001 -- a secondary leg must not cost the line
            configs2 = dict(configs2 or {}, error=repr(e)[:500])
        ten2 = None
        torch.cuda.empty_cache()
        # the same problem on one GPU, one layer: local work on every rank (no collective inside the guarded part; a dry
        # run with N ranks on ONE GPU may not have room for N copies of it)
        err1, tl, t1 = None, 0.0, None
        try:
            g1 = torch.Generator(device=dev).manual_seed(98)
            t1 = [torch.randn(1, S2, N_HEADS, HEAD_DIM, generator=g1, device=dev, dtype=torch.float32).to(torch.bfloat16)
                  for _ in range(4)]
            lay1 = SeqLayout("contiguous", 1, S2)

            def one_layer2():
                o, l = ring_forward(HipBlockOps, SingleComm(), t1[0], t1[1], t1[2], layout=lay1, causal=True)
                ring_backward(HipBlockOps, SingleComm(), t1[0], t1[1], t1[2], o, l, t1[3], layout=lay1, causal=True)

            one_layer2()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            one_layer2()
            torch.cuda.synchronize()
            tl = time.perf_counter() - t0
        except Exception as e:      # noqa: BLE001
            err1 = repr(e)[:500]
        t1 = None
        torch.cuda.empty_cache()
        ok1 = all_ranks_ok(err1 is None)
        tl = max_over_ranks(tl)
        if ok1 and "tokens_per_s" in configs2:
            tps1 = S2 / (tl * args.layers)
            configs2["same_problem_on_1_gpu"] = {"ms_per_layer": tl * 1e3, "tokens_per_s": tps1,
                                                 "speedup": configs2["tokens_per_s"] / tps1,
                                                 "strong_scaling_efficiency": configs2["tokens_per_s"] / tps1 / world}
        elif not ok1:
            configs2["same_problem_on_1_gpu"] = {"error": err1 or "failed on another rank"}

    def other_config(tag, Sx, layers_x, packed_docs):
        """One more configuration of BASELINE.json in the same N > 1 line: the sequence ring at S = Sx over these N GPUs,
        `layers_x` layers (scaled to 32 and labelled), one warm-up step + one timed step."""
        # The set-up (a 1M-token shard is 1 GiB per tensor) may fail on ONE rank: the ranks vote before the first step, so
        # that a failure skips the leg everywhere instead of leaving the others inside the ring's exchange.
        err, tenx, segx, lens = None, None, None, None
        try:
            pair_units = 0.5 * float(Sx) * Sx       # causal: half the square
            if packed_docs:
                sg, lens = packed_documents(Sx, lo_frac=Sx // 4096, hi_frac=Sx // 262144)      # 4096 .. 262144 tokens
                pair_units = 0.5 * sum(float(l) * l for l in lens)
            if packed_docs and args.layout == "zigzag" and c_ring is not None:
                # a packed batch brings its own ownership (what a loader knows): chunks handed out by visible pairs
                from lwm_amd.ring import balanced_layout
                layx = balanced_layout(world, Sx, lens, chunks_per_rank=4)
            else:
                layx = SeqLayout(args.layout, world, Sx)
            cx = layx.local_len
            gx = torch.Generator(device=dev).manual_seed(8765 + rank)
            tenx = [torch.randn(1, cx, N_HEADS, HEAD_DIM, generator=gx, device=dev, dtype=torch.float32).to(torch.bfloat16)
                    for _ in range(4)]
            if packed_docs:
                segx = torch.from_numpy(sg).to(dev)
        except Exception as e:      # noqa: BLE001
            err = repr(e)[:500]
        vote = torch.tensor([0 if err else 1], dtype=torch.int32, device="cpu" if shared else dev)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN)
        if int(vote.item()) == 0:
            del tenx
            torch.cuda.empty_cache()
            return {"error": err or "the set-up failed on another rank", "skipped_on_every_rank": True}
        try:
            sent0 = c_ring.bytes_sent if c_ring is not None else None
            step(ten=tenx, lay=layx, seg=segx, layers=layers_x)
            barrier()
            t0 = time.perf_counter()
            step(ten=tenx, lay=layx, seg=segx, layers=layers_x)
            barrier()
            el = max_over_ranks(time.perf_counter() - t0)
            out = {
                "workload": f"LWM-7B attention fwd+bwd, B=1, S={Sx}, H=32, D=128, causal" +
                            (f", {len(lens)} packed documents (4096..262144 tokens)" if packed_docs else "") +
                            f", ring={world}, layout={layx.kind}; {layers_x} layers timed, scaled to 32 [BASELINE {tag}" +
                            ("" if world == 8 else f" at N={world}") + "]",
                "seq_len": Sx, "layers_timed": layers_x, "steps": 1, "ms_per_layer": el * 1e3 / layers_x,
                "tokens_per_s": Sx / (el / layers_x * N_LAYERS), "tokens_per_s_per_gpu": Sx / (el / layers_x * N_LAYERS) / world,
                "path_algorithmic_tflops_per_gpu": 7.0 * 2.0 * pair_units * D_MODEL / (el / layers_x) / 1e12 / world,
                "exchange": ({"schedule": f"{sched_c} (C-ABI driver, {args.transport})",
                              "bytes_sent_per_rank_per_layer": (c_ring.bytes_sent - sent0) / (2 * layers_x)} if c_ring is not None
                             else {"schedule": getattr(comm, "schedule", None)}),
            }
            del tenx
            return out
        except Exception as e:      # noqa: BLE001 -- a secondary leg must not cost the line
            return {"error": repr(e)[:500]}

    configs3 = configs4 = None
    if world > 1 and not args.no_configs34 and not args.packed and (not shared or args.force_configs34):
        torch.cuda.empty_cache()
        configs3 = other_config("configs[3] sequence (262144 tokens; its VQGAN tokenisation is the `vqgan` leg of the N = 1 line)", 262144, 2, False)
        torch.cuda.empty_cache()
        configs4 = other_config("configs[4]", 1 << 20, 2, True)

    if world > 1 and c_ring is not None and not args.no_transport_trials and not args.packed:
        # LAST of the secondary legs (a hang here costs only this object: the late watchdog prints the line as it stands):
        # the same few layers under EVERY (transport, schedule) pair the C driver has -- RCCL send/recv kernels or the CU-free
        # IPC transport, the direct schedule or the neighbour ring -- against the same launches with a transport that moves
        # nothing: four exposed_ms_per_step figures in one line, and which pair a real run should take.
        try:
            from lwm_amd.ring_c import CRing
            lt = 4
            gq = torch.Generator(device=dev).manual_seed(777 + rank)
            tq = [torch.randn(1, c, N_HEADS, HEAD_DIM, generator=gq, device=dev, dtype=torch.float32).to(torch.bfloat16)
                  for _ in range(4)]

            def run_layers(ring):
                step(ring, ten=tq, lay=layout, layers=1)
                barrier()
                t0_ = time.perf_counter()
                step(ring, ten=tq, lay=layout, layers=lt)
                barrier()
                return max_over_ranks(time.perf_counter() - t0_) / lt

            compute = {}
            for sc in ("direct", "ring"):
                nr = null_ring(sc)
                compute[sc] = run_layers(nr)
                nr.close()
            pairs = {}
            for tr in (("ipc",) if shared else ("rccl", "ipc")):
                for sc in ("direct", "ring"):
                    ok, why, ring, mine = 1, "", None, (tr, sc) == (args.transport, sched_c)
                    try:
                        CRing.probe(tr)
                    except Exception as e:      # noqa: BLE001
                        ok, why = 0, repr(e)[:300]
                    if all_ranks_ok(ok):
                        try:
                            ring = c_ring if mine else CRing(dist.group.WORLD, transport=tr, layout=args.layout, schedule=sc,
                                                             ipc_slot_bytes=c * N_HEADS * HEAD_DIM * 4, ipc_slots=8)
                        except Exception as e:      # noqa: BLE001
                            ok, why = 0, repr(e)[:300]
                        if all_ranks_ok(ok):
                            try:
                                per_layer = run_layers(ring)
                                pairs[f"{tr}/{sc}"] = {
                                    "ms_per_layer": per_layer * 1e3, "compute_only_ms_per_layer": compute[sc] * 1e3,
                                    "exposed_ms_per_step": (per_layer - compute[sc]) * 1e3 * args.layers,
                                    "tokens_per_s": S / (per_layer * args.layers), "timed_the_value": mine}
                            except Exception as e:      # noqa: BLE001
                                pairs[f"{tr}/{sc}"] = {"error": repr(e)[:300]}
                        else:
                            pairs[f"{tr}/{sc}"] = {"error": why or "the set-up failed on another rank"}
                        if ring is not None and not mine:
                            ring.close()
                    else:
                        pairs[f"{tr}/{sc}"] = {"error": why or "the transport is not available on another rank"}
            good = {k_: v_ for k_, v_ in pairs.items() if "tokens_per_s" in v_}
            transport_trials = {"layers_timed": lt, "pairs": pairs,
                                "fastest": max(good, key=lambda k_: good[k_]["tokens_per_s"]) if good else None,
                                "value_was_timed_on": f"{args.transport}/{sched_c}",
                                "reserved_cus": int(os.environ.get("LWM_RING_RESERVE_CUS", "0") or 0),
                                "note": "a few layers per pair AFTER the timed region; rerun with --transport / --schedule set to "
                                        "`fastest` when it differs from value_was_timed_on"}
            del tq
        except Exception as e:      # noqa: BLE001 -- a secondary leg must not cost the line
            transport_trials = {"error": repr(e)[:500]}

    dog.disarm()
    if rank == 0:
        res = main_line()
        ms_per_step = res["ms_per_step"]
        unit = gemm_unit_flops(S) if doc_sq is None else doc_sq * D_MODEL
        if world == 1:
            ks = timer.summary()
            res["kernels"] = ks
            # dominant kernel by total time; algorithmic FLOPs per launch: fwd = 2 GEMM
            # units; the backward's 5 algorithmic units are apportioned to its two launches
            # by executed share (dkdv 4/7, dq 3/7) -- DESIGN.md "Work accounting".
            algo_units = {"attn_fwd64_kernel": 2.0, "attn_bwd_dkdv4_kernel": 5.0 * 4 / 7, "attn_bwd_dq4_kernel": 5.0 * 3 / 7}
            exec_units = {"attn_fwd64_kernel": 2.0, "attn_bwd_dkdv4_kernel": 4.0, "attn_bwd_dq4_kernel": 3.0}
            cand = {n: d for n, d in ks.items() if n in algo_units}
            dom = max(cand, key=lambda n: cand[n]["total_ms"])
            avg_s = cand[dom]["avg_ms"] * 1e-3
            achieved = algo_units[dom] * unit / avg_s / 1e12
            res["roofline"] = {
                "kernel": dom, "bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                "traffic": pmc_traffic(dom, S)[0], "traffic_profile": pmc_traffic(dom, S)[1],
                "traffic_source": ("committed rocprofv3 --pmc profile of the same command (bench.py cannot collect counters)"
                                   if pmc_traffic(dom, S)[0] is not None else
                                   "none: no committed rocprofv3 --pmc summary carries the stamp of the tree's kernel sources "
                                   f"({attn_kernel_stamp()}); bench.py cannot collect counters"),
                "traffic_of_earlier_kernel_sources": (None if pmc_traffic(dom, S)[0] is not None or pmc_traffic_any(dom, S, False)[0] is None else
                                                      dict(zip(("bytes", "profile", "kernel_source_stamp"), pmc_traffic_any(dom, S, False)))),
                "avg_launch_ms": cand[dom]["avg_ms"],
                "executed_tflops": exec_units[dom] * unit / avg_s / 1e12,
                "all_kernels_algorithmic_tflops": {
                    n: algo_units[n] * unit / (d["avg_ms"] * 1e-3) / 1e12 for n, d in cand.items()},
            }
            def leg(fn, *a, **kw):
                """a secondary leg must not cost the line: its failure is reported in its own object"""
                try:
                    return fn(*a, **kw)
                except Exception as e:      # noqa: BLE001
                    torch.cuda.empty_cache()
                    return {"error": repr(e)[:500]}

            if not args.no_cpu_baseline:
                res["cpu_baseline"] = leg(cpu_baseline, S)
            if not args.no_vqgan:
                res["vqgan"] = leg(vqgan_leg, torch)
                res["packed"] = leg(packed_leg, torch)
                res["model_slice"] = leg(model_slice_leg, torch)
                res["config0_fp32"] = leg(config0_fp32_leg, torch)      # BASELINE configs[0] in its own dtype, beside cpu_baseline.config1
                if not args.no_full_model:
                    res["model_full"] = leg(model_full_leg, torch)
                if not args.no_packed_1m:
                    res["packed_1m"] = leg(packed_1m_leg, torch)
                res["decode"] = leg(decode_leg, torch)
                res["generate"] = leg(generate_leg, torch)
                res["ring8_compute_model"] = leg(ring_model_leg, torch)
                res["ring8_compute_model_32k"] = leg(ring_model_leg, torch, S=32768, reps=5)     # what `--gpus 8` runs by default
                res["ring8_compute_model_32k_per_pair"] = leg(ring_model_leg, torch, S=32768, driver="python", reps=5)   # (the launch list of rounds 1-4)
                if not args.no_packed_1m:
                    # BASELINE configs[4] under ring 8: 1,048,576 tokens in 15 packed documents, zigzag -- does the
                    # ownership that balances a full causal triangle balance documents too?
                    res["ring8_compute_model_packed_1m_zigzag"] = leg(ring_model_leg, torch, S=1 << 20, packed=True, reps=1)
                    # ... no (max / mean 1.9): an ownership table from the document lengths, 4 chunks per rank by visible pairs
                    res["ring8_compute_model_packed_1m"] = leg(ring_model_leg, torch, S=1 << 20, packed=True, layout="balanced", reps=1)
                res["elementwise"] = leg(elementwise_leg, torch)
                # the 1 -> 8 curve as a MODEL (compute per rank measured here, link time from planned bytes): what the first
                # real 8-GPU run is to be held against
                known = {}
                for key_, n_, S_ in (("ring8_compute_model", 8, 131072), ("ring8_compute_model_32k", 8, 32768)):
                    if isinstance(res.get(key_), dict) and "per_rank_ms_per_layer" in res[key_]:
                        known[(n_, S_)] = res[key_]
                res["predicted_scaling"] = leg(predicted_scaling_leg, torch, known)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
