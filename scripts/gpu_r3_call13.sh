# round 3, call 13: fused backward, query-tile-major partial slots: timing + per-kernel stats
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c13; rm -rf $O; mkdir -p $O
(cd $R && timeout 300 python -m pytest tests/test_gpu_attention.py -x -q -k "fused or case or packed" 2>&1 | tail -4) > $O/pytest.txt < /dev/null
cat $O/pytest.txt
timeout 200 $R/scripts/micro/fused_bench $R/lwm_amd/liblwm_hip.so 32768 32 3 all > $O/fused_timing.txt 2>&1 < /dev/null
cat $O/fused_timing.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- $R/scripts/micro/fused_bench $R/lwm_amd/liblwm_hip.so 32768 32 3 fused > $O/prof.log 2>&1 < /dev/null
find $O/prof -name '*kernel_stats.csv' -exec cat {} \; > $O/kernel_stats.txt < /dev/null
head -5 $O/kernel_stats.txt
