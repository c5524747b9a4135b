# the decode-projection GEMV: parity tests, the generation anchors (HF tokens), and the bench's generate leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_llama_ops.py tests/test_gpu_hf_anchor.py tests/test_gpu_llama_model.py tests/test_gpu_vision_llama.py -q -x > gpurun_out/gemv_tests.log 2>&1; echo "rc=$?" >> gpurun_out/gemv_tests.log
tail -4 gpurun_out/gemv_tests.log
python - <<'PY'
import torch, bench, json
print(json.dumps(bench.generate_leg(torch)))
PY
