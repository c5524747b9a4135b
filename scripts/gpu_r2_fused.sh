#!/bin/bash
# round 2: the one-launch backward on the hardware -- parity, determinism, A/B timing, HBM traffic
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_attention.py -q -x -k "fwd_bwd_vs_oracle or fused or carries" > gpurun_out/fused_tests.log 2>&1
echo "rc=$?" >> gpurun_out/fused_tests.log
tail -4 gpurun_out/fused_tests.log
for mode in "--fused-bwd" "--two-kernel-bwd"; do
  timeout 300 python bench.py --steps 2 --warmup 1 --layers 8 --no-cpu-baseline --no-vqgan $mode > gpurun_out/ab_fused$mode.json 2> gpurun_out/ab_fused$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_fused$mode.json").read().strip().splitlines()[-1])
print("mode='$mode' tok/s(32L-equiv) %.0f" % (d['value']*8/32), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()}, d['roofline']['kernel'], round(d['roofline']['frac'],3))
PY
done
if [ -n "$FUSED_PMC" ]; then
  cd /tmp && export TMPDIR=/tmp
  R=$GRAFT_REPO_ROOT
  for P in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc_f2 -o $P -- python $R/bench.py --steps 1 --warmup 0 --layers 1 --no-cpu-baseline --no-vqgan --fused-bwd > /dev/null 2>&1
    python - <<PY
import csv,glob
tot=0
for f in glob.glob("$R/gpurun_out/pmc_f2/${P}_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_bwd_fused" in r["Kernel_Name"] and r["Counter_Name"]=="$P": tot+=float(r["Counter_Value"])
print("$P fused kernel: %.2f GB" % (tot*1024*(2 if "$P"=="FETCH_SIZE" else 1)/1e9))
PY
  done
fi
