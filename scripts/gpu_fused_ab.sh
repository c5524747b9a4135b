# One GPU call: the L2 hand-off probe (timings only) and scripts/micro/fused_bench over the product library and
# every build/ab/liblwm_*.so variant (scripts/ab_build.sh).  AB_PMC=1 adds FETCH/WRITE/TCC counter passes of the
# fused launch for the libraries named in AB_PMC_LIBS.
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/fused_ab; rm -rf $O; mkdir -p $O
timeout 120 $R/scripts/micro/l2_handoff 512 > $O/l2_timing.txt 2>&1
for lib in $R/lwm_amd/liblwm_hip.so $R/build/ab/liblwm_*.so; do
  timeout 120 $R/scripts/micro/fused_bench $lib ${AB_S:-32768} 32 ${AB_REPS:-3} ${AB_WHAT:-all} >> $O/timing.txt 2>&1
done
if [ -n "$AB_PMC_LIBS" ]; then
  for name in $AB_PMC_LIBS; do
    lib=$R/build/ab/liblwm_$name.so; [ "$name" = product ] && lib=$R/lwm_amd/liblwm_hip.so
    i=1
    for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
      (timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_$name -o pass$i -- $R/scripts/micro/fused_bench $lib 32768 32 1 fused 2>&1 | tail -2) > $O/pmc_${name}_pass$i.log
      i=$((i+1))
    done
  done
  python3 - <<PY
import csv, glob, collections, os
csv.field_size_limit(1 << 30)
with open("$O/counters.txt", "w") as f:
    for d in sorted(glob.glob("$O/pmc_*/")):
        acc = collections.OrderedDict()
        for fn in sorted(glob.glob(d + "pass*_counter_collection.csv")):
            n = collections.Counter()
            for row in csv.DictReader(open(fn, newline="")):
                if "attn_bwd_fused_kernel" not in row["Kernel_Name"]:
                    continue
                acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                n[row["Counter_Name"]] = max(n[row["Counter_Name"]], int(row["Dispatch_Id"]))
        # two launches per run (warm-up + 1 rep): halve
        f.write(os.path.basename(d[:-1]) + "  " + "  ".join(f"{k}={v / 2:.4g}" for k, v in acc.items()) + "  (per launch)\n")
PY
fi
cat $O/l2_timing.txt $O/timing.txt; [ -f $O/counters.txt ] && cat $O/counters.txt
