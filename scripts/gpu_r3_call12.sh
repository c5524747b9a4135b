# round 3, call 12: per-kernel times of the fused backward (main launch, reduction, lse2)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c12; rm -rf $O; mkdir -p $O
for lib in lwm_amd/liblwm_hip.so build/ab/liblwm_nostore.so; do
  n=$(basename $lib .so)
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -o run -- $R/scripts/micro/fused_bench $R/$lib 32768 32 3 fused > $O/$n.log 2>&1 < /dev/null
  echo "== $n" >> $O/kernel_stats.txt
  find $O/prof_$n -name '*kernel_stats.csv' -exec cat {} \; >> $O/kernel_stats.txt < /dev/null
done
cat $O/kernel_stats.txt < /dev/null
