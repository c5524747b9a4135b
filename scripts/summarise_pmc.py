"""gpurun_out/pmc/pass*_counter_collection.csv (one rocprofv3 --pmc pass per counter set, written by
scripts/gpu_r1_final.sh) -> profiles/<tag>_pmc_attention_1layer.json.   usage: summarise_pmc.py r01e

Per lwm:: kernel (one launch each at --layers 1): every counter summed over the launch's rows, the
launch duration seen in each pass, and the derived figures: fetch_bytes = FETCH_SIZE KiB x 1024 x 2
(the gfx950 correction for 16 B/lane reads, MI355X_MICROARCH.md HBM section), write_bytes =
WRITE_SIZE KiB x 1024, mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01x"
dirs = sys.argv[2:] or ["pmc"]          # gpurun_out/<dir>: several runs (e.g. pmc pmc_fused) are merged by kernel name
csv.field_size_limit(1 << 30)
kern = {}
files = []
for d in dirs:
    files += sorted(glob.glob(os.path.join(ROOT, "gpurun_out", d, "pass*_counter_collection.csv")))
owner = {}       # kernel -> the run directory it is taken from (a kernel that appears in several runs counts once)
for fn in files:
    pas = re.search(r"(pass\d+)_", os.path.basename(fn)).group(1)
    run = os.path.basename(os.path.dirname(fn))
    with open(fn, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if not name.startswith("lwm::attn"):
                continue
            short = name.split("(")[0].replace("lwm::", "")
            if owner.setdefault(short, run) != run:
                continue
            k = kern.setdefault(short, {"duration_ns_by_pass": {}, "run": run})
            k[row["Counter_Name"]] = k.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            k["VGPR_Count"] = int(row["VGPR_Count"]) + int(row.get("Accum_VGPR_Count") or 0)
            k["duration_ns_by_pass"][pas] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
for k in kern.values():
    if "FETCH_SIZE" in k:
        k["fetch_bytes"] = k["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in k:
        k["write_bytes"] = k["WRITE_SIZE"] * 1024
    if "fetch_bytes" in k and "write_bytes" in k:
        k["hbm_traffic_bytes"] = k["fetch_bytes"] + k["write_bytes"]
    if k.get("GRBM_GUI_ACTIVE"):
        k["mfma_util"] = k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (k["GRBM_GUI_ACTIVE"] / 8 * 1024)
        k["lds_idx_active_frac"] = k.get("SQ_LDS_IDX_ACTIVE", 0.0) / (k["GRBM_GUI_ACTIVE"] / 8 * 256)
sys.path.insert(0, ROOT)
from bench import attn_kernel_stamp      # noqa: E402  (the stamp bench.py checks before it quotes this file)

out = {
    "kernel_source_stamp": attn_kernel_stamp(),
    "command": "rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 1 --warmup 0 --layers 1 --no-cpu-baseline "
               "--no-vqgan  (one pass per counter set; S=32768, 32 heads, 1 layer, one launch per kernel)",
    "notes": "FETCH_SIZE/WRITE_SIZE are KiB as reported; fetch_bytes applies the gfx950 x2 correction for wide (16 B/lane) "
             "reads; WRITE_SIZE is taken as reported. SQ_* cycle counters are quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES "
             "and SQ_LDS_IDX_ACTIVE (cycles, summed over SIMDs / CUs); GRBM_GUI_ACTIVE is summed over the 8 XCDs. "
             "Counter passes run at lower clocks than un-profiled runs: compare ratios, not durations.",
    "kernels": kern,
}
dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_attention_1layer.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
for n, k in kern.items():
    print(f"{n:28s} mfma_util {k.get('mfma_util', 0):.3f}  lds_active {k.get('lds_idx_active_frac', 0):.3f}  "
          f"hbm {k.get('hbm_traffic_bytes', 0) / 1e9:.2f} GB")
print("wrote", dst)
