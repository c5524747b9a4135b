# round 3, call 29: the fused decode layers (LWM_DECODE_FUSED=1): parity tests + the generate leg with and without
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c29; rm -rf $O; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_llama_ops.py tests/test_gpu_hf_anchor.py -q -x -k "gemv or hf or fused or graph or greedy" 2>&1 < /dev/null | tail -6 > $O/pytest.txt
cat $O/pytest.txt
for f in 0 1; do
LWM_DECODE_FUSED=$f timeout 120 python - >> $O/generate_ab.txt 2>&1 < /dev/null <<'PY'
import os, json, torch, bench
r = bench.generate_leg(torch)
print("LWM_DECODE_FUSED=" + os.environ["LWM_DECODE_FUSED"], json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "workload"}))
PY
done
grep DECODE_FUSED $O/generate_ab.txt
