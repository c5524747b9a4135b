# round 3, call 28: two-kernel vs fused-backward flavour, driver-style 20-step runs on ONE box (A B A B)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c28; rm -rf $O; mkdir -p $O
for i in 1 2; do
for f in two-kernel-bwd fused-bwd; do
  timeout 200 python bench.py --$f --steps 20 --warmup 5 --no-vqgan --no-full-model --no-cpu-baseline 2> /dev/null < /dev/null | tail -1 > $O/bench_${f}_$i.json
done
done
python - <<'PY'
import json, glob
for fn in sorted(glob.glob("gpurun_out/r3c28/bench_*.json")):
    d = json.loads(open(fn).read().strip())
    print(fn.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 1), {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()})
PY
