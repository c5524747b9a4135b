# Round-1 full GPU pass: all GPU tests, smoke, bench, rocprof stats, PMC passes.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/tests.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/smoke.log
(timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -1) > gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01i -- python $R/bench.py --steps 1 --warmup 0 --layers 4 --no-cpu-baseline 2>&1 | tail -2) > $R/gpurun_out/prof.log
bash $R/scripts/gpu_pmc_attention.sh
cd $R
for f in tests smoke bench; do echo "=== $f"; cat gpurun_out/$f.log; done
head -12 $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) | cut -c1-160
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 1 --layers 2 --seq 16384 --backend gloo 2>&1 | tail -1 | cut -c1-400) > gpurun_out/dry2.log; echo "=== dry-run N=2"; cat gpurun_out/dry2.log
