# round 3, call 11: the fused backward with bf16 dq partials + streaming reduction (no atomics)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/r3c11; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | tail -8 > $O/pytest.txt
cat $O/pytest.txt
for lib in lwm_amd/liblwm_hip.so build/ab/liblwm_plainstore.so build/ab/liblwm_nostore.so; do
  timeout 300 scripts/micro/fused_bench $R/$lib 32768 32 3 all >> $O/fused_timing.txt 2>&1
done
timeout 300 scripts/micro/fused_bench $R/lwm_amd/liblwm_hip.so 32768 32 3 fused 8 >> $O/fused_timing.txt 2>&1
cat $O/fused_timing.txt
