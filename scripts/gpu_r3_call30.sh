R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c30; rm -rf $O; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_hf_anchor.py tests/test_gpu_infer.py -q -x 2>&1 < /dev/null | tail -6 > $O/pytest.txt
cat $O/pytest.txt
LWM_DECODE_FUSED=1 timeout 200 python -m pytest tests/test_gpu_hf_anchor.py tests/test_cli.py tests/test_gpu_vision_llama.py -q -x 2>&1 < /dev/null | tail -4 > $O/pytest_fused_env.txt
cat $O/pytest_fused_env.txt
