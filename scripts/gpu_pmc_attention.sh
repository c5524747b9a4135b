# The four counter passes of the attention kernels (1 layer, one launch per kernel); summarise with
# `python scripts/summarise_pmc.py <tag>` afterwards (here, on the merged gpurun_out/pmc).
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --layers 1 --no-cpu-baseline --no-vqgan"
OUT=${PMC_DIR:-pmc}; rm -rf $R/gpurun_out/$OUT; i=1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  (timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/$OUT -o pass$i -- $B $PMC_BENCH_ARGS 2>&1 | tail -2) > $R/gpurun_out/${OUT}_pass$i.log
  i=$((i+1))
done
