R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c9; rm -rf $O; mkdir -p $O
timeout 120 $R/scripts/micro/fused_bench $R/lwm_amd/liblwm_hip.so 32768 32 6 two > $O/timing.txt 2>&1
cat $O/timing.txt
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_infer.py tests/test_gpu_ring_sim.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
cat $O/pytest.txt
