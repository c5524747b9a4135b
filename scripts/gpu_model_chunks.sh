# model_full (bench.py) across the reference's FFN knobs: scan_mlp (chunk + recompute) on / off, chunk size
cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, bench
for kw in (dict(scan_mlp=False), dict(mlp_chunk=8192), dict(mlp_chunk=1024)):
    r = bench.model_full_leg(torch, **kw)
    print(kw, round(r["ms_per_step"]), "ms", round(r["tokens_per_s"]), "tok/s", round(r["model_tflops"]), "TF", round(r["peak_hbm_gib"], 1), "GiB", r["loss"], flush=True)
PY
