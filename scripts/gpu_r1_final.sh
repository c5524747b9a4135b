# Round-1 full GPU pass: all GPU tests, smoke, bench, rocprof stats, PMC passes.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/tests.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/smoke.log
(timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -1) > gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof $R/gpurun_out/pmc
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01d -- python $R/bench.py --steps 1 --warmup 0 --layers 4 --no-cpu-baseline 2>&1 | tail -2) > $R/gpurun_out/prof.log
B="python $R/bench.py --steps 1 --warmup 0 --layers 1 --no-cpu-baseline --no-vqgan"
i=1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  (timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc -o pass$i -- $B 2>&1 | tail -2) > $R/gpurun_out/pmc_pass$i.log
  i=$((i+1))
done
cd $R
for f in tests smoke bench; do echo "=== $f"; cat gpurun_out/$f.log; done
head -12 $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) | cut -c1-160
