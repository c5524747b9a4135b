#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ring_c.py -m gpu -q > gpurun_out/ringc.log 2>&1; echo "rc=$?" >> gpurun_out/ringc.log; tail -6 gpurun_out/ringc.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err; tail -4 gpurun_out/bench_r2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r2.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],1), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
print({k: round(v["avg_ms"],3) for k,v in d["kernels"].items()})
for k in ("model_full","model_slice","vqgan","packed","decode","generate"):
    v=d.get(k); 
    if v: print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,str,list))}, v.get("config4_tokenisation",""))
PY
timeout 300 python bench.py --gpus 2 --backend gloo --steps 1 --warmup 0 --layers 1 --seq 4096 > gpurun_out/dry2_r2.json 2> gpurun_out/dry2_r2.err; echo "dry rc=$?"; tail -c 600 gpurun_out/dry2_r2.json
