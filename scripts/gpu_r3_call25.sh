# round 3, call 25: forward, only the exponent fmas of phase 2b deferred to the top of the next iteration
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3c25; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
for v in d0 d1; do
  timeout 100 $R/scripts/micro/fused_bench $R/build/ab/liblwm_$v.so 32768 32 5 two 2>&1 < /dev/null | cut -c1-100 | sed "s#.*/build/ab/##" >> $O/fwd_timing.txt
done
done
cat $O/fwd_timing.txt
cd $R
LWM_HIP_LIB=$R/build/ab/liblwm_d1.so timeout 300 python -m pytest tests/test_gpu_attention.py -q -x 2>&1 < /dev/null | tail -3
