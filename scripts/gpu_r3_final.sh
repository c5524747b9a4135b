# Round-3 evidence pass: the whole GPU suite + smoke, the driver-style bench line, rocprofv3 kernel stats of the
# bench command (4 layers), the four PMC passes (1 layer; separate passes, kernel trace only).
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3final; rm -rf $O; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 < /dev/null | tail -25) > $O/tests.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | tail -3) > $O/smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err < /dev/null
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03 -- python $R/bench.py --steps 1 --warmup 0 --layers 4 --no-cpu-baseline --no-full-model 2>&1 < /dev/null | tail -2) > $O/prof.log
rm -rf $R/gpurun_out/pmc
PMC_DIR=pmc bash $R/scripts/gpu_pmc_attention.sh < /dev/null
cd $R
for f in tests smoke; do echo "=== $f"; cat $O/$f.log; done
tail -4 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3final/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],1), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
print({k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
print("cpu_baseline", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["cpu_baseline"].items() if k in ("value", "cores", "gflops")}, d["cpu_baseline"].get("config1", {}).get("seconds"))
for k in ("model_full","model_slice","vqgan","packed","decode","generate"):
    v=d.get(k)
    if v: print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,str,list))})
PY
find $O/prof -name "*kernel_stats.csv" -exec head -8 {} \; < /dev/null | cut -c1-150
ls gpurun_out/pmc | head
