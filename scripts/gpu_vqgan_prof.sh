# rocprofv3 kernel statistics of the bench's VQGAN leg (1 attention layer rides along)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/vqprof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vqprof -o vq -- python $R/bench.py --steps 1 --warmup 0 --layers 1 --no-full-model --no-cpu-baseline > $R/gpurun_out/vqprof.log 2>&1
find $R/gpurun_out/vqprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/vq_kernel_stats.csv
find $R/gpurun_out/vqprof -name "*kernel_trace.csv" -size +1M -delete
head -25 $R/gpurun_out/vq_kernel_stats.csv | cut -c1-150
