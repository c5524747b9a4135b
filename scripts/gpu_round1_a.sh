mkdir -p gpurun_out
(timeout 400 python -m pytest tests/test_gpu_probe.py -m gpu -q 2>&1 | tail -40) > gpurun_out/probe.log
(timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q -k "not full_size" 2>&1 | tail -80) > gpurun_out/attn.log
(timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q -k "full_size" 2>&1 | tail -40) > gpurun_out/attn_full.log
(timeout 1200 python bench.py --steps 2 --warmup 1 2>&1 | tail -20) > gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --layers 2 --no-cpu-baseline 2>&1 | tail -15) > $GRAFT_REPO_ROOT/gpurun_out/prof.log
cd $GRAFT_REPO_ROOT
for f in probe attn attn_full bench prof; do echo "=== $f"; cat gpurun_out/$f.log; done
find gpurun_out/prof -name "*stats*" | head; 
