# conv_bench on random and on all-zero operands (same instruction stream: separates the clock from the kernel), with the shader
# clock and socket power sampled beside it -- product library and every build/ab/liblwmv_*.so
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/conv_power; rm -rf $O; mkdir -p $O
for lib in $R/lwm_amd/liblwm_hip.so $R/build/ab/liblwmv_*.so; do
  for amp in 1 0; do
    echo "### $(basename $lib) operand amplitude x$amp" >> $O/timing.txt
    ( for i in $(seq 1 8); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | egrep -i "sclk|Socket Graphics Package Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.4; done ) > $O/smi_$(basename $lib)_$amp.txt &
    S=$!
    LWM_BENCH_AMP=$amp timeout 200 $R/scripts/micro/conv_bench $lib 32 12 2>&1 | head -12 >> $O/timing.txt
    wait $S
    sort $O/smi_$(basename $lib)_$amp.txt | uniq -c | sort -rn | head -4 >> $O/timing.txt
  done
done
cat $O/timing.txt
