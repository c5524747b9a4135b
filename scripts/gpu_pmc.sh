# PMC passes over one fwd+bwd layer (S=32768) -- counters in their own runs, kernel-trace only.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
B="python $R/bench.py --steps 1 --warmup 0 --layers 1 --no-cpu-baseline"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES"
i=1
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT"; do
  (timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc -o pass$i -- $B 2>&1 | tail -3) > $R/gpurun_out/pmc/pass$i.log
  i=$((i+1))
done
cd $R; ls gpurun_out/pmc; tail -2 gpurun_out/pmc/pass*.log
