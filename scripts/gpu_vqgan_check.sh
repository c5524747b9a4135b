# VQGAN verification pass: GPU parity tests of the VQGAN kernels, then the bench's VQGAN leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vqgan.py -x -q > gpurun_out/vq_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/vq_tests.log
timeout 600 python bench.py --steps 1 --warmup 0 --layers 1 --no-full-model --no-cpu-baseline > gpurun_out/vq_bench.json 2> gpurun_out/vq_bench.err
tail -3 gpurun_out/vq_tests.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/vq_bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["vqgan"], indent=1))
PY
