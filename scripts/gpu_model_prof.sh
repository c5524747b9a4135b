# rocprofv3 kernel statistics of the bench's model_full leg (32 layers of LWM-7B, S = 32768, fwd+bwd)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/mprof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mprof -o m -- python $R/bench.py --steps 1 --warmup 0 --layers 1 --no-cpu-baseline > $R/gpurun_out/mprof.log 2>&1
find $R/gpurun_out/mprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/model_kernel_stats.csv
find $R/gpurun_out/mprof -name "*kernel_trace.csv" -size +1M -delete
head -30 $R/gpurun_out/model_kernel_stats.csv | cut -c1-110
tail -c 600 $R/gpurun_out/mprof.log
