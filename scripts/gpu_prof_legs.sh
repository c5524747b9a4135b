#!/bin/bash
# rocprofv3 kernel trace of bench.py's rank-by-rank ring model (scripts/gpu_ring_legs.py) and the per-launch list of a
# few ranks:  gpurun --timeout 400 -- 'CASE=32768:c:0:zigzag TAG=r05b bash scripts/gpu_prof_legs.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export TMPDIR=/tmp
O=$R/gpurun_out/${TAG:-legs}; mkdir -p "$O"
CASE=${CASE:-32768:c:0:zigzag}
cd /tmp; rm -rf /tmp/prof_legs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_legs -o ks -- python "$R/scripts/gpu_ring_legs.py" "$CASE" > "$O/prof_legs.log" 2>&1
f=$(find /tmp/prof_legs -name '*kernel_stats.csv' | head -1)
t=$(find /tmp/prof_legs -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] || { echo "no kernel stats"; tail -20 "$O/prof_legs.log"; exit 1; }
cp "$f" "$O/legs_kernel_stats.csv"
python - "$t" <<'PY' | tee "$O/legs_launches.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
att = [r for r in rows if "lwm::" in r["Kernel_Name"]]
per = len(att) // 8 // 4          # 8 ranks x (1 warm-up + 3 timed layers)
print("launches of lwm:: kernels:", len(att), "per rank and layer:", per)
for rank in (0, 3, 7):
    seg = att[(rank * 4 + 3) * per:(rank * 4 + 4) * per]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print(f"rank {rank}: span {(t1 - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us")
    for r in seg:
        g = r.get("Grid_Size") or r.get("Grid_Size_X")
        print(f"    {r['Kernel_Name'].split('(')[0].replace('lwm::', ''):32s} grid {g:>9s}  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us")
PY
