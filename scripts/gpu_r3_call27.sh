# round 3, call 27: `bench.py --gpus 8` through the C driver over IPC with all 8 ranks on ONE GPU (functional: the N = 8
# line incl. configs2 at c = 16384), then the 20-step line of the fused-backward flavour
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c27; rm -rf $O; mkdir -p $O
timeout 420 python bench.py --gpus 8 --backend gloo --transport ipc --steps 1 --warmup 1 --layers 2 > $O/bench_ipc8.json 2> $O/bench_ipc8.err < /dev/null
tail -c 2200 $O/bench_ipc8.json; grep -v "socket.cpp\|Gloo\|amdgpu.ids" $O/bench_ipc8.err | tail -4
timeout 300 python bench.py --fused-bwd --steps 20 --warmup 5 --no-vqgan --no-full-model --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err < /dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c27/bench_fused.json").read().strip().splitlines()[-1])
print("fused flavour:", round(d["value"]), d["ms_per_step"], {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()}, d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
PY
