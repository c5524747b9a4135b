# Round-1 GPU pass C: all GPU tests, bench, rocprof stats, PMC passes.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > gpurun_out/tests.log
(timeout 600 python bench.py --steps 3 --warmup 1 2>&1 | tail -3) > gpurun_out/bench.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > gpurun_out/smoke.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01c -- python $R/bench.py --steps 1 --warmup 0 --layers 4 --no-cpu-baseline 2>&1 | tail -3) > $R/gpurun_out/prof.log
B="python $R/bench.py --steps 1 --warmup 0 --layers 1 --no-cpu-baseline --no-vqgan"
i=1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  (timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc -o pass$i -- $B 2>&1 | tail -3) > $R/gpurun_out/pmc_pass$i.log
  i=$((i+1))
done
cd $R
for f in tests bench smoke prof; do echo "=== $f"; cat gpurun_out/$f.log; done
cat $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -30
ls gpurun_out/pmc* | head
