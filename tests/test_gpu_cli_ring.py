"""The PRODUCT entry point over a sequence ring: `python -m lwm_amd.cli.train --mesh_dim=1,1,1,8` as 8 real processes
that share the one GPU of a test box (gloo for the bootstrap and the gradient all-reduce, the library's IPC transport
under the C ring driver for the K/V exchange -- LWM_DIST_BACKEND=gloo LWM_RING_TRANSPORT=ipc), against the same command
on one process.

What must hold ("Next round" item 1):
  * the harness, the loader slice and `ringattention` agree on the ownership rule -- zigzag by default -- without being
    told (no layout flag on the command line);
  * the exchange goes through the C-ABI driver (lwm_ring_attn_fwd / _bwd), not the torch.distributed driver;
  * loss and EVERY parameter gradient of the step equal the 1-process run's: both are bf16 evaluations of the same
    function, so each is held to the fp32 CPU model (oracle/llama_model_ref.py, the criterion of
    tests/test_gpu_llama_model.py) with the ring run inside the global bound of tests/_parity.py of it or at most twice
    as far from it as the 1-process run, and the two to each other within twice that bound and its cosine;
  * the attention launches of the 8 ranks, timed one rank at a time, are balanced: max / mean <= 1.05 (the reference's
    contiguous ownership, lwm/llama.py:560-562, gives ~1.9 at n = 8: asserted too, as the control).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = 8192            # the step that is compared (the fp32 CPU model is its judge)
S_BALANCE = 65536   # the shard shape the balance report times (c = 8192 per rank at n = 8) ...
H_BALANCE = 32      # ... with LWM-7B's 32 heads (the debug model's 2 heads make 64 workgroups for 256 CUs: the longest
#                     workgroup, not the work, would set the time)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _argv(n, dump, extra=(), vision=False):
    kind = "json_vision" if vision else "json"
    return [sys.executable, "-m", "lwm_amd.cli.train", f"--mesh_dim=1,1,1,{n}", "--dtype=bf16", "--total_steps=1",
            "--log_freq=1", "--load_llama_config=debug", "--seed=11",
            f"--update_llama_config=dict(vocab_size=512,max_sequence_length={S},theta=1000000" + (",vision_vocab_size=256" if vision else "") + ")",
            f"--train_dataset.type={kind}", f"--train_dataset.{kind}_dataset.seq_length={S}",
            f"--train_dataset.{kind}_dataset.batch_size=1", "--optimizer.adamw_optimizer.lr=1e-4",
            f"--lwm_dump_grads={dump}", *(["--modality=vision,text"] if vision else []), *extra]


def _run(n, dump, env_extra=None, extra=(), timeout=900, vision=False):
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if n == 1:
        r = subprocess.run(_argv(1, dump, extra, vision), cwd=ROOT, env=base, capture_output=True, text=True, timeout=timeout)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        return [r.stdout + r.stderr]
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(base, RANK=str(rank), WORLD_SIZE=str(n), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo",
                   LWM_DIST_BACKEND="gloo", LWM_RING_TRANSPORT="ipc", **(env_extra or {}))
        procs.append(subprocess.Popen(_argv(n, dump, extra, vision), cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.returncode for p in procs]
    if not all(c == 0 for c in codes):
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        for r_, o in enumerate(outs):
            with open(os.path.join(ROOT, "gpurun_out", f"cli_ring_fail_rank{r_}.log"), "w") as f:
                f.write(o)
    assert all(c == 0 for c in codes), (codes, [o[-2500:] for o in outs])
    return outs


def _balance(out):
    """The ownership rule the job ran under (its own report line names it), then the per-rank attention time of that rule's
    launch lists measured HERE, in a process that has the GPU to itself: inside the job the ranks' processes share the one
    GPU of a test box, and their idle contexts cost the measuring process scheduler time slices (the job's own figures
    are 3x too long and erratic: lwm_amd/cli/train.py::balance_report)."""
    import torch
    from lwm_amd.cli.train import balance_report
    from lwm_amd.llama import LLaMAConfig
    line = [l for l in out.splitlines() if l.startswith("LWM_BALANCE ")][-1]
    job = json.loads(line[len("LWM_BALANCE "):])
    n = len(job["ms_per_rank"])
    cfg = LLaMAConfig.load_config("debug")
    bal = balance_report(cfg, 1, S_BALANCE // n, n, torch.device("cuda", 0), heads=H_BALANCE, layout=job["layout"])
    bal["inside_the_job_on_a_shared_gpu"] = job
    return bal


@pytest.mark.gpu
def test_train_cli_on_an_8_rank_ring_equals_one_process(tmp_path):
    import torch
    from tests import _parity
    ref_f, ring_f = str(tmp_path / "ref.pt"), str(tmp_path / "ring.pt")
    _run(1, ref_f)
    outs = _run(8, ring_f, env_extra={"LWM_RING_DRIVER": "c"}, extra=("--lwm_balance_report", f"--lwm_balance_seq={S_BALANCE}", f"--lwm_balance_heads={H_BALANCE}"))
    # the ownership rule and the driver the product chose by itself
    assert "layout zigzag" in outs[0], outs[0][-1500:]
    assert "'driver': 'c'" in outs[0] and "'transport': 'ipc'" in outs[0] and "'layout': 'zigzag'" in outs[0], outs[0][-1500:]
    ref, ring = torch.load(ref_f), torch.load(ring_f)
    assert torch.equal(ref["tokens"], ring["tokens"]) and all(torch.equal(ref["params"][n], ring["params"][n]) for n in ref["params"])
    assert set(ring["grads"]) == set(ref["grads"]) == set(ref["params"]) and len(ref["grads"]) >= 20
    # the judge of both: the fp32 CPU model on the same parameters and batch
    from lwm_amd.llama import LLaMAConfig
    from oracle import llama_model_ref as M
    cfg = LLaMAConfig.load_config("debug").update(dict(vocab_size=512, max_sequence_length=S, theta=1000000))
    st = {n: p.clone().requires_grad_(True) for n, p in ref["params"].items()}
    tok = ref["tokens"]
    rl, ra = M.forward_loss(st, cfg, tok[:, :-1], tok[:, 1:])
    rl.backward()
    for run in (ref, ring):
        assert abs(run["loss"] - rl.item()) <= 1e-2 * abs(rl.item()), (run["loss"], rl.item())
        assert abs(run["metrics"]["accuracy"] - ra.item()) <= 2e-3
    assert abs(ring["loss"] - ref["loss"]) <= 2e-3 * abs(ref["loss"]), (ring["loss"], ref["loss"])
    worst = {}
    for name, g_ref in ref["grads"].items():
        a, b, t = ring["grads"][name].double().flatten(), g_ref.double().flatten(), st[name].grad.double().flatten()
        assert t.abs().max() > 0, name
        e_ring, e_ref = ((a - t).abs().max() / t.abs().max()).item(), ((b - t).abs().max() / t.abs().max()).item()
        cos = lambda x, y: float((x @ y) / (x.norm() * y.norm()).clamp_min(1e-30))
        assert cos(a, t) >= 0.99 and cos(b, t) >= 0.99, (name, cos(a, t), cos(b, t))            # tests/test_gpu_llama_model.py
        assert abs(float(a.norm() / t.norm()) - 1) <= 5e-2 and abs(float(b.norm() / t.norm()) - 1) <= 5e-2, name
        # the ring run sits inside the attention parity bound of the truth, or at most twice as far from it as the 1-process
        # run (8 bf16 partial gradients summed in f32 instead of one bf16 matmul: measured 0.0075 against 0.0051 on wte)
        assert e_ring <= max(_parity.TOL, 2 * e_ref), (name, e_ring, e_ref)
        _parity.check(f"grad {name}", a.numpy(), b.numpy(), tol=2 * _parity.TOL, row_tol=None)   # ... and the two agree
        worst[name] = (_parity.STATS[-1][1], e_ring, e_ref)
    bal = _balance(outs[0])
    assert bal["layout"] == "zigzag" and len(bal["ms_per_rank"]) == 8
    assert bal["max_over_mean"] <= 1.05, bal
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "cli_ring8.json"), "w") as f:
        w = max(worst, key=lambda n: worst[n][0])
        json.dump({"loss_fp32_cpu": rl.item(), "loss_1proc": ref["loss"], "loss_ring8": ring["loss"],
                   "worst_grad": w, "ring_vs_1proc_err_over_max": worst[w][0],
                   "ring_vs_fp32_err_over_max": max(v[1] for v in worst.values()),
                   "1proc_vs_fp32_err_over_max": max(v[2] for v in worst.values()), "balance": bal}, f, indent=1)


@pytest.mark.gpu
def test_train_cli_contiguous_ownership_is_the_unbalanced_control(tmp_path):
    """LWM_SP_LAYOUT=contiguous (the reference's blocks): same loss, and the imbalance the default removes."""
    import torch
    ref_f, ring_f = str(tmp_path / "ref.pt"), str(tmp_path / "ring.pt")
    _run(1, ref_f)
    outs = _run(4, ring_f, env_extra={"LWM_SP_LAYOUT": "contiguous"}, extra=("--lwm_balance_report", f"--lwm_balance_seq={S_BALANCE}", f"--lwm_balance_heads={H_BALANCE}"))
    assert "layout contiguous" in outs[0]
    ref, ring = torch.load(ref_f), torch.load(ring_f)
    assert abs(ring["loss"] - ref["loss"]) <= 2e-3 * abs(ref["loss"])
    bal = _balance(outs[0])
    assert bal["layout"] == "contiguous" and bal["max_over_mean"] >= 1.3, bal


@pytest.mark.gpu
def test_train_cli_vision_text_on_a_ring(tmp_path):
    """--modality=vision,text over 4 ranks: two embedding tables, two heads, two masked losses whose target counts differ
    from rank to rank (the vision block sits mid-sequence) -- each rank's loss is its share of the per-sequence means
    (the counts are summed over the ring), RoPE positions come from the ownership rule: the 1-process figures."""
    import torch
    ref_f, ring_f = str(tmp_path / "ref.pt"), str(tmp_path / "ring.pt")
    _run(1, ref_f, vision=True)
    outs = _run(4, ring_f, vision=True)
    assert "layout zigzag" in outs[0]
    ref, ring = torch.load(ref_f), torch.load(ring_f)
    assert abs(ring["loss"] - ref["loss"]) <= 2e-3 * abs(ref["loss"]), (ring["loss"], ref["loss"])
    for k in ("vision_loss", "text_loss", "vision_acc", "text_acc"):
        assert abs(ring["metrics"][k] - ref["metrics"][k]) <= 2e-3 * max(1.0, abs(ref["metrics"][k])), (k, ring["metrics"][k], ref["metrics"][k])
    for name, g_ref in ref["grads"].items():
        a, b = ring["grads"][name].double().flatten(), g_ref.double().flatten()
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))
        assert cos >= 0.9999, (name, cos)
