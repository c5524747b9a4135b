"""rocprofv3 kernel trace of bench.model_full_leg (one warm-up + one timed step, identical launches) -> the per-class table
of ONE step: attention kernels, library GEMMs by kernel / grid, this library's elementwise kernels, torch elementwise
kernels, and the gap to the wall clock.   usage: model_full_table.py <kernel_trace.csv> [model_full.json]"""
import csv
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
wall = None
if len(sys.argv) > 2:
    try:
        wall = json.load(open(sys.argv[2]))
    except Exception:
        wall = None
INIT = ("distribution_elementwise", "randint", "random_", "normal_")


def short(n):
    n = n.split("(")[0]
    n = n.replace("void at::native::", "at::").replace("void lwm::", "").replace("lwm::", "")
    if n.startswith(("Cijk", "Custom_Cijk")):
        m = re.search(r"(Cijk_A[a-z]+_B[a-z]+)", n)
        mt = re.search(r"MT(\d+x\d+x\d+)", n)
        sk = re.search(r"_(SK\d)_", n)
        return "gemm " + (m.group(1) if m else "?") + " MT" + (mt.group(1) if mt else "?") + (" " + sk.group(1) if sk else "")
    return n[:70]


def klass(n):
    if n.startswith("attn_"):
        return "attention (hand-written HIP)"
    if n.startswith("gemm "):
        return "library GEMM (hipBLASLt)"
    if n.startswith("wgrad_"):
        return "weight-gradient GEMM (hand-written HIP)"
    if n.startswith("at::"):
        return "torch elementwise / copy"
    return "elementwise (hand-written HIP)"


agg = defaultdict(lambda: [0, 0.0])
t_first, t_last = None, None
for r in rows:
    name = r["Kernel_Name"]
    if any(k in name for k in INIT):
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    t_first = s if t_first is None else min(t_first, s)
    t_last = e if t_last is None else max(t_last, e)
    g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    key = (short(name), g)
    agg[key][0] += 1
    agg[key][1] += (e - s) / 1e6
steps = 2.0
by_class = defaultdict(float)
for (n, g), (c, ms) in agg.items():
    by_class[klass(n)] += ms / steps
total = sum(by_class.values())
print(f"kernel time per step (ms), by class  [trace of 2 identical steps / 2]")
for k, v in sorted(by_class.items(), key=lambda kv: -kv[1]):
    print(f"  {k:36s} {v:9.2f}  {100 * v / total:5.1f} %")
print(f"  {'sum of kernels':36s} {total:9.2f}")
if wall:
    w = wall["ms_per_step"]
    print(f"  wall clock of the timed step          {w:9.2f}   (gaps between kernels: {w - total:.2f} ms = {100 * (w - total) / w:.1f} %;"
          f" profiler attached)")
print()
print(f"{'kernel':72s} {'grid':>10s} {'calls/step':>10s} {'avg ms':>9s} {'ms/step':>9s}")
for (n, g), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if ms / steps < 0.3:
        continue
    print(f"{n:72s} {g:>10s} {c / steps:10.1f} {ms / c:9.4f} {ms / steps:9.2f}")
