# Round 3, third GPU call: attn_fwd64 with in-phase DMA and balanced fillers: timing (A/B: DMA at the top, prescaled Q,
# the 8-wave kernel), parity tests, SQ counters of the forward.
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3c3; rm -rf $O; mkdir -p $O
timeout 60 $R/scripts/micro/fused_bench $R/lwm_amd/liblwm_hip.so 2048 8 1 two > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
cat $O/smoke.txt
if grep -q "smoke rc=0" $O/smoke.txt; then
  for lib in $R/lwm_amd/liblwm_hip.so $R/build/ab/liblwm_f4dmatop.so $R/build/ab/liblwm_f4pre.so $R/build/ab/liblwm_fwdold.so; do
    timeout 120 $R/scripts/micro/fused_bench $lib 32768 32 6 two >> $O/timing.txt 2>&1
  done
  cat $O/timing.txt
  cd $R
  timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_infer.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
  cat $O/pytest.txt
  cd /tmp
  for name in product fwdold; do
    lib=$R/build/ab/liblwm_$name.so; [ "$name" = product ] && lib=$R/lwm_amd/liblwm_hip.so
    i=1
    for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
      (timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_$name -o pass$i -- $R/scripts/micro/fused_bench $lib 32768 32 1 two 2>&1 | tail -2) > $O/pmc_${name}_pass$i.log
      i=$((i+1))
    done
  done
  python3 - <<PY
import csv, glob, collections, os
csv.field_size_limit(1 << 30)
with open("$O/counters.txt", "w") as f:
    for d in sorted(glob.glob("$O/pmc_*/")):
        acc = collections.OrderedDict()
        for fn in sorted(glob.glob(d + "**/pass*_counter_collection.csv", recursive=True)):
            for row in csv.DictReader(open(fn, newline="")):
                if "attn_fwd" not in row["Kernel_Name"]:
                    continue
                key = (row["Kernel_Name"].split("(")[0][-24:], row["Counter_Name"])
                acc.setdefault(key, []).append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            f.write(f"{os.path.basename(d[:-1])} {k} {c} = {sum(v)/len(v):.5g} (x{len(v)})\n")
PY
  cat $O/counters.txt
fi
