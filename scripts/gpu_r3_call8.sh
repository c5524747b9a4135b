R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c8; rm -rf $O; mkdir -p $O
timeout 60 $R/scripts/micro/fused_bench $R/lwm_amd/liblwm_hip.so 2048 8 1 two > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
cat $O/smoke.txt
if grep -q "smoke rc=0" $O/smoke.txt; then
  for lib in $R/lwm_amd/liblwm_hip.so $R/build/ab/liblwm_dqold.so; do
    timeout 120 $R/scripts/micro/fused_bench $lib 32768 32 4 two >> $O/timing.txt 2>&1
  done
  cat $O/timing.txt
  timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
  cat $O/pytest.txt
fi
