# L2 hand-off probe (scripts/micro/l2_handoff.hip): timings, then one rocprofv3 --pmc pass per counter set.
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/l2probe; rm -rf $O; mkdir -p $O
T=${L2_TILES:-512}
timeout 120 $R/scripts/micro/l2_handoff $T > $O/timing.txt 2>&1
i=1
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  (timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O -o pass$i -- $R/scripts/micro/l2_handoff $T 2>&1 | tail -3) > $O/pass$i.log
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
acc = collections.OrderedDict()
for fn in sorted(glob.glob("$O/pass*_counter_collection.csv")):
    for row in csv.DictReader(open(fn, newline="")):
        k = row["Kernel_Name"]
        if "handoff" not in k:
            continue
        d = acc.setdefault(k, collections.OrderedDict())
        d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
with open("$O/counters.txt", "w") as f:
    for k, d in acc.items():
        f.write(k.split("(")[0] + "  " + "  ".join(f"{n}={v:.4g}" for n, v in d.items()) + "\n")
PY
cat $O/timing.txt $O/counters.txt
