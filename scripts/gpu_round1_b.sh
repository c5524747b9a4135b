mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > gpurun_out/tests.log
(LWM_DKDV_WAVES=4 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/bench_w4.log
(LWM_DKDV_WAVES=8 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/bench_w8.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01b -- python $R/bench.py --steps 1 --warmup 0 --layers 4 --no-cpu-baseline 2>&1 | tail -5) > $R/gpurun_out/prof.log
cd $R
for f in tests bench_w4 bench_w8 prof; do echo "=== $f"; cat gpurun_out/$f.log; done
find gpurun_out/prof -name "*stats*" | head; cat $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -20
