# round 3, call 14: fused backward, partial store late in the step (vmcnt(1)) vs early (vmcnt(0))
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c14; rm -rf $O; mkdir -p $O
for lib in lwm_amd/liblwm_hip.so build/ab/liblwm_early.so lwm_amd/liblwm_hip.so build/ab/liblwm_early.so; do
  timeout 200 $R/scripts/micro/fused_bench $R/$lib 32768 32 3 all >> $O/fused_timing.txt 2>&1 < /dev/null
done
cat $O/fused_timing.txt
(cd $R && timeout 400 python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | tail -4) > $O/pytest.txt < /dev/null
cat $O/pytest.txt
