# round 3, call 15: fused backward partial-store cache policy (nt | sc1 | sc0 sc1), then the attention leg of bench.py both ways
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c15; rm -rf $O; mkdir -p $O
for lib in lwm_amd/liblwm_hip.so build/ab/liblwm_sc1.so build/ab/liblwm_sc0sc1.so lwm_amd/liblwm_hip.so; do
  timeout 200 $R/scripts/micro/fused_bench $R/$lib 32768 32 3 fused >> $O/fused_timing.txt 2>&1 < /dev/null
done
cat $O/fused_timing.txt
cd $R
timeout 600 python bench.py --fused-bwd --no-vqgan --no-full-model --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_fused.json 2> $O/bench_fused.err < /dev/null
timeout 600 python bench.py --two-kernel-bwd --no-vqgan --no-full-model --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_two.json 2> $O/bench_two.err < /dev/null
python - <<'PY'
import json
for n in ("fused", "two"):
    try:
        d = json.loads(open(f"gpurun_out/r3c15/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["value"]), d["ms_per_step"], {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()})
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/bench_fused.err
