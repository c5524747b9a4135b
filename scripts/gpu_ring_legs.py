"""One GPU call: the rank-by-rank compute models of bench.py (ring_model_leg) for a list of cases.
usage: python scripts/gpu_ring_legs.py  [case ...]   case = S:driver:packed(0/1):layout   (default: the round-5 set)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

cases = sys.argv[1:] or ["32768:python:0:zigzag", "32768:c:0:zigzag", "131072:python:0:zigzag", "131072:c:0:zigzag",
                         "1048576:c:1:zigzag", "1048576:c:1:balanced"]
for case in cases:
    S, driver, packed, layout = case.split(":")
    S = int(S)
    try:
        r = bench.ring_model_leg(torch, n=8, S=S, driver=driver, packed=bool(int(packed)), layout=layout, reps=1 if S > 200000 else 3)
    except Exception as e:  # noqa: BLE001
        r = {"error": repr(e)[:800]}
    print(case, json.dumps(r), flush=True)
    torch.cuda.empty_cache()
