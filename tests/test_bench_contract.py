"""The bench line the driver consumes: the newest committed profiles/*_bench.json (written by
the driver's command `python bench.py --gpus 1 --steps 20 --warmup 5` on an MI355X, scripts/gpu_round_check.sh) must carry every field of the
contract, with the metric and workload BASELINE.json names, a roofline object for the dominant kernel and a
CPU baseline from the same run -- and bench.py must keep the command-line contract."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    """rNN_bench.json is the line of round NN's last evidence pass; rNNx_bench.json (x = a, b, ...) are earlier passes."""
    import re
    files = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "*_bench.json")):
        m = re.fullmatch(r"r(\d+)([a-z]*)_bench\.json", os.path.basename(f))
        if m:
            files.append(((int(m.group(1)), m.group(2) == "", m.group(2)), f))
    assert files, "no committed bench line under profiles/"
    return json.loads(open(max(files)[1]).read().strip().splitlines()[-1])


def test_committed_bench_line_follows_the_contract():
    d = _latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "tokens/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and "synthetic" in d["data"]
    assert "S=32768" in d["config"]["workload"] and "configs[1]" in d["config"]["workload"]
    assert "model" not in d["config"]
    assert abs(d["value"] - 32768 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 1e9          # HBM bytes per launch of the dominant kernel
    # achieved = algorithmic FLOPs per launch / measured launch duration (dK/dV carries 4/7 of the 5 backward units)
    unit = 2.0 * 32768 ** 2 * 4096 / 2
    assert abs(r["achieved"] - (5.0 * 4 / 7) * unit / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    # round 3: BASELINE configs[0] end to end and the op points ride in the same object, one `cores` convention
    # (`cores` = the threads a leg computed with: the blockwise port runs at the best of a thread sweep, the dense model
    # and the OpenMP VQGAN oracle on every thread)
    assert c["config1"]["tokens_per_s"] > 0 and "configs[0]" in c["config1"]["workload"] and c["config1"]["cores"] >= c["cores"]
    assert [p["S"] for p in c["op_points"]] == [4096, 8192, 16384]
    assert d["vqgan"]["cpu_baseline"]["cores"] == c["config1"]["cores"]
    # round 4: the one-wave-per-SIMD backward kernels; the 1M-token leg in the default line; where `traffic` comes from;
    # cpu_baseline.value = the port's fastest operating point, scaled
    assert {"attn_fwd64_kernel", "attn_bwd_delta_kernel", "attn_bwd_dkdv4_kernel", "attn_bwd_dq4_kernel"} <= set(d["kernels"])
    assert r["kernel"] == "attn_bwd_dkdv4_kernel" and "traffic_source" in r
    assert d["packed_1m"]["s_per_layer"] > 0 and "S=1048576" in d["packed_1m"]["workload"]
    if "convention" not in c:       # a line of round 4: the port's fastest op point, scaled by the S^2 law
        assert r["traffic_profile"] is None or r["traffic_profile"].startswith("profiles/r04")
        best = max(p["gflops"] for p in c["op_points"])
        assert abs(c["gflops"] - best) < 1e-6 * best
        return
    # round 5: THE convention of cpu_baseline.value, frozen -- measured at the workload's own S (one head, one layer, one
    # pass), multiplied out over 32 heads x 32 layers; nothing is scaled in S
    sys.path.insert(0, ROOT)
    import bench
    m = c["measured_at"]
    if c["convention"] == bench.CPU_BASELINE_CONVENTION_R05:       # a line of round 5: one head on 8 or 32 threads, multiplied out
        assert (m["S"], m["heads"], m["layers"]) == (32768, 1, 1) and "extrapolated" not in c["sample"]
        assert abs(c["value"] - 32768 / (m["seconds"] * 32 * 32)) < 1e-6 * c["value"]
        assert abs(c["gflops"] - 7.0 * 32768 ** 2 * 128 / m["seconds"] / 1e9) < 1e-6 * c["gflops"]
        assert c["cores"] in (8, 32) or c["cores"] == c["config1"]["cores"]
    else:
        # round 6, frozen: one layer's 32 heads side by side on EVERY host thread (floor(threads / 8) processes x 8 threads)
        assert c["convention"] == bench.CPU_BASELINE_CONVENTION
        assert c["cores"] == c["config1"]["cores"] and m["processes"] * m["threads_per_process"] <= c["cores"]
        assert (m["S"], m["layers"]) == (32768, 1) and 1 <= m["heads_timed"] <= 32 and "extrapolated" not in c["sample"]
        assert abs(m["layer_seconds"] - m["wall_seconds"] * 32 / m["heads_timed"]) < 1e-9 * m["layer_seconds"]
        assert abs(c["value"] - 32768 / (m["layer_seconds"] * 32)) < 1e-6 * c["value"]
        assert abs(c["gflops"] - 7.0 * 32768 ** 2 * 4096 / m["layer_seconds"] / 1e9) < 1e-6 * c["gflops"]
        one = c["measured_at_one_head"]                            # the earlier convention's figure stays beside it
        assert abs(c["value_one_head_convention"] - 32768 / (one["seconds"] * 32 * 32)) < 1e-6 * c["value_one_head_convention"]
        # round 6: the full-model leg carries its own roofline object; the 1 -> 8 curve rides as a labelled MODEL
        mf = d["model_full"]["roofline"]
        assert mf["bound"] == "mfma" and mf["peak"] == 2500.0 and abs(mf["frac"] - mf["achieved"] / mf["peak"]) < 1e-9
        assert abs(mf["achieved"] - d["model_full"]["model_tflops"]) < 1e-9 and "time_shares" in mf
        if "share" in mf["time_shares"]:
            assert abs(sum(mf["time_shares"]["share"].values()) - 1.0) < 1e-3
            assert {"attention_hip", "library_gemm"} <= set(mf["time_shares"]["share"])
        wg = d["model_full"].get("wgrad_gemm")         # round 6, second half: the hand-written weight-gradient GEMM inside the step
        if wg is not None and wg["kernel_ms_per_step"]:
            assert wg["bound"] == "mfma" and abs(wg["achieved"] - wg["flops_per_step"] / (wg["kernel_ms_per_step"] * 1e-3) / 1e12) < 1e-6
            assert abs(wg["frac"] - wg["achieved"] / 2500.0) < 1e-9 and wg["achieved"] < 2500.0
            assert "wgrad_gemm_hip" in mf["time_shares"]["share"]
        ps = d["predicted_scaling"]
        assert ps["kind"] == "model, not measured" and set(ps["by_S"]) == {"32768", "131072"}
        for S_, rows in ps["by_S"].items():
            assert set(rows) == {"1", "2", "4", "8"} and rows["1"]["efficiency"] == 1.0
            for n_ in ("2", "4", "8"):
                r_ = rows[n_]
                assert r_["bytes_sent_per_rank_per_layer"] > 0 and r_["link_ms_per_layer"] > 0 and r_["bound"] in ("compute", "link")
                # (a modelled efficiency a little above 1 is the single launch at S = 131072 running a few per cent below the
                #  shard launches, profiles/r06_null_transport_fill.txt -- not a claim of super-linear scaling)
                assert 0.0 < r_["efficiency_nothing_hidden"] <= r_["efficiency"] <= 1.12
    # roofline.traffic comes from a PMC summary stamped with the kernel sources of the tree (bench.attn_kernel_stamp)
    if r["traffic_profile"] is not None:
        prof = json.load(open(os.path.join(ROOT, r["traffic_profile"])))
        assert "kernel_source_stamp" in prof
    # ... and the tree as it is: a summary of OTHER kernel sources never becomes `traffic`; it is reported beside a null one
    got, other = bench.pmc_traffic("attn_bwd_dkdv4_kernel", 32768), bench.pmc_traffic_any("attn_bwd_dkdv4_kernel", 32768, False)
    for byts, prof, *stamp in (got, other):
        if prof is not None:
            doc = json.load(open(os.path.join(ROOT, prof)))
            assert (doc["kernel_source_stamp"] == bench.attn_kernel_stamp()) == (not stamp) and byts > 1e9
    # the ring-8 compute models run the product's launch list (the C driver, gathered form) and carry BASELINE configs[4]
    assert d["ring8_compute_model_32k"]["driver"] == "c" and d["ring8_compute_model_32k"]["form"] == "gathered"
    assert "packed documents" in d["ring8_compute_model_packed_1m"]["workload"]


def test_counter_evidence_is_of_the_kernel_sources_that_ship():
    """the NEWEST committed PMC summary of the attention kernels (profiles/r*_pmc_attention_1layer.json)
    must be stamped with the sha256 of the kernel sources of THIS tree (bench.attn_kernel_stamp) -- a change under
    lwm_amd/csrc/attn_* without a fresh counter pass (scripts/gpu_pmc_attention.sh + summarise_pmc.py) fails here, instead
    of surfacing as `roofline.traffic: null` in the driver's line."""
    import re
    sys.path.insert(0, ROOT)
    import bench
    files = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_attention_1layer.json")):
        m = re.fullmatch(r"r(\d+)([a-z]*)_pmc_attention_1layer\.json", os.path.basename(f))
        if m:
            files.append(((int(m.group(1)), m.group(2) == "", m.group(2)), f))
    assert files
    newest = max(files)[1]
    doc = json.load(open(newest))
    assert doc["kernel_source_stamp"] == bench.attn_kernel_stamp(), \
        f"{os.path.basename(newest)} is of other kernel sources: re-run scripts/gpu_pmc_attention.sh + summarise_pmc.py"
    byts, prof = bench.pmc_traffic("attn_bwd_dkdv4_kernel", 32768)
    assert prof is not None and byts > 1e9
    for kname in ("attn_fwd64_kernel", "attn_bwd_dkdv4_kernel", "attn_bwd_dq4_kernel"):
        kk = doc["kernels"][kname]
        assert 0.3 < kk["mfma_util"] < 1.0 and kk["hbm_traffic_bytes"] > 1e9


def test_bench_cli_contract_without_a_gpu():
    """`python bench.py --gpus N` must launch its own N ranks (the driver's N > 1 command may be the plain
    one); under a launcher --gpus must agree with WORLD_SIZE; and with no GPU to give to a rank the run ends
    with a diagnosable JSON error line and a non-zero code -- never a hang, never a bare traceback."""
    bench = os.path.join(ROOT, "bench.py")
    # (1) launched by someone else with the wrong size
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    # (2) plain command, no torch.distributed environment: self-launch under torch.distributed.run
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo",
                        "--init-timeout", "60"], capture_output=True, text=True, env=env, timeout=600)
    assert "self-launch" in r.stderr and "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    import torch
    if not torch.cuda.is_available():
        # both ranks report, from inside the launched processes, that they have no device
        assert r.returncode != 0
        assert lines and all(l["value"] is None and l["error"]["stage"] == "devices" for l in lines)
        assert {l["error"]["world_size"] for l in lines} == {2} and {l["error"]["rank"] for l in lines} == {0, 1}
    else:
        # (a gloo bootstrap is not RCCL: the field that names RCCL stays null in a dry run)
        assert r.returncode == 0 and lines[-1]["n_gpus"] == 2 and lines[-1]["bootstrap_ranks_seen"] == 2
        assert lines[-1]["rccl_ranks_seen"] is None and lines[-1]["bootstrap_backend"] == "gloo"
    h = subprocess.run([sys.executable, bench, "--help"], capture_output=True, text=True, timeout=300)
    for flag in ("--gpus", "--steps", "--warmup", "--driver", "--transport", "--schedule", "--layout", "--no-configs2",
                 "--no-configs34", "--no-packed-1m"):
        assert flag in h.stdout


def test_bench_watchdog_turns_a_hang_into_an_error_line():
    """A stage of the N > 1 path that never returns (a stuck peer exchange cannot raise) ends the process with the
    {"error": ...} line on rank 0's stdout, a note on the other ranks' stderr, and exit code 7."""
    body = ("import sys, time, argparse; sys.path.insert(0, %r); import bench; "
            "d = bench.Watchdog(argparse.Namespace(gpus=2, steps=1, warmup=0), 0.3); d.arm('quick stage'); d.disarm(); "
            "d.arm('stuck stage'); time.sleep(30)" % ROOT)
    for rank in ("0", "1"):
        env = dict(os.environ, RANK=rank, WORLD_SIZE="2")
        r = subprocess.run([sys.executable, "-c", body], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 7
        lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        if rank == "0":
            assert len(lines) == 1 and lines[0]["value"] is None
            assert lines[0]["error"]["stage"] == "watchdog: stuck stage" and lines[0]["error"]["type"] == "TimeoutError"
        else:
            assert not lines and "stuck stage" in r.stderr
    # a stage after which the line's value exists: the callback prints what is known and chooses the exit code
    body = ("import sys, time, json, argparse; sys.path.insert(0, %r); import bench; "
            "d = bench.Watchdog(argparse.Namespace(gpus=2, steps=1, warmup=0), 60.0); "
            "d.arm('late stage', 0.3, on_fire=lambda stage, err: (print(json.dumps({'value': 1.0, 'late': stage})), 0)[1]); "
            "time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", body], capture_output=True, text=True, env=dict(os.environ, RANK="0"), timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip()) == {"value": 1.0, "late": "late stage"}
