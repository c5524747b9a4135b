"""Runs scripts/micro/mfma_timing.hip on the GPU box:  cycles per MFMA for one or two waves per SIMD.

Build first (here):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -shared -fPIC -I lwm_amd/csrc \
        scripts/micro/mfma_timing.hip -o build/ab/libmfma_timing.so
"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "build", "ab", "libmfma_timing.so")
import torch
lib = C.CDLL(so)
lib.mfma_time_run.argtypes = [C.c_int] * 5 + [C.c_void_p] * 3
out = torch.zeros(8, dtype=torch.int64, device="cuda"); sink = torch.zeros(4, device="cuda")
IT = 200
print("nacc mode(0 none,1 b128,2 tr) nvalu | cycles/MFMA  1 wave/SIMD | 2 waves/SIMD (per SIMD: /2)")
for nacc, mode, nv in [(1,0,0),(2,0,0),(4,0,0),(1,1,0),(2,1,0),(4,1,0),(2,2,0),(4,2,0),(4,0,2),(4,0,4),(4,0,6),(4,0,8),(4,1,2),(4,1,4),(4,1,6),(2,1,4)]:
    row = []
    for threads in (256, 512):
        for _ in range(2):
            rc = lib.mfma_time_run(nacc, mode, nv, threads, IT, out.data_ptr(), sink.data_ptr(), None)
            assert rc == 0, rc
            torch.cuda.synchronize()
        row.append(float(out[: threads // 64].max()) / (IT * 16))
    print(f"{nacc:4d} {mode:4d} {nv:5d} | {row[0]:8.1f} | {row[1]:8.1f}  ({row[1] / 2:.1f} per MFMA on the shared pipe)")
