"""lwm_wgrad_bf16 (hand-written: both operands read where they lie) against hipBLASLt on the weight-gradient shapes of the
LWM-7B step at S = 32768: x.t() @ g as it is, and with the narrow operand transposed first (what the harness did in round 6's
first half).  Correctness against torch on the way.    gpurun -- 'python scripts/gpu_wgrad_bench.py'"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from lwm_amd import _capi  # noqa: E402
from lwm_amd._lib import lib  # noqa: E402
from lwm_amd.llama_ops import transpose2d  # noqa: E402

S, d, f = 32768, 4096, 11008


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    L = lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    only = os.environ.get("WG_SHAPES", "wqkv,wo,w13,w2").split(",")       # WG_MINE_ONLY=1: skeleton variants (no check, no library)
    mine_only = os.environ.get("WG_MINE_ONLY") == "1"
    for tag, K, N in (("wqkv", d, 3 * d), ("wo", d, d), ("w13", d, 2 * f), ("w2", f, d)):
        if tag not in only:
            continue
        x = (torch.randn(S, K, device="cuda") * 0.5).to(torch.bfloat16)
        g = (torch.randn(S, N, device="cuda") * 0.5).to(torch.bfloat16)
        dw = torch.empty(K, N, device="cuda", dtype=torch.bfloat16)
        ok = K % 256 == 0 and N % 256 == 0

        nws = L.lwm_wgrad_workspace_bytes(S, K, N)
        ws = torch.empty(max(nws, 16), dtype=torch.uint8, device="cuda")

        def mine():
            _capi.check(L, L.lwm_wgrad_bf16(x.data_ptr(), K, g.data_ptr(), N, dw.data_ptr(), N, S, K, N, ws.data_ptr(), nws, st),
                        "lwm_wgrad_bf16")

        fl = 2.0 * S * K * N
        if mine_only:
            t_me = timed(mine, 10)
            print(f"{tag:5s} lwm_wgrad_bf16 {t_me:7.3f} ms {fl / t_me / 1e9:7.1f} TF/s  [{os.environ.get('LWM_HIP_LIB', 'tree')}]", flush=True)
            continue
        t_lib = timed(lambda: torch.matmul(x.t(), g))
        if K <= N:
            t_tr = timed(lambda: torch.matmul(transpose2d(x), g))
        else:
            t_tr = timed(lambda: torch.matmul(x.t(), transpose2d(g).t()))
        line = f"{tag:5s} K={K:6d} N={N:6d}  hipBLASLt x.t()@g {t_lib:7.3f} ms {fl / t_lib / 1e9:7.1f} TF/s | narrow transpose + hipBLASLt {t_tr:7.3f} ms {fl / t_tr / 1e9:7.1f}"
        if ok:
            t_me = timed(mine)
            # element-wise: a correctly rounded bf16 result is within 2^-8 of the f32 sum (relative; elements near zero are
            # measured against 1 % of the largest instead) -- the f32 reference in column chunks (an (S, N) f32 g is 2.9 GB)
            big = max(float(torch.matmul(x.t().float(), g[:, c0:c0 + 4096].float()).abs().max()) for c0 in range(0, N, 4096))
            err = 0.0
            for c0 in range(0, N, 4096):
                ref = torch.matmul(x.t().float(), g[:, c0:c0 + 4096].float())
                err = max(err, float(((dw[:, c0:c0 + 4096].float() - ref).abs() / ref.abs().clamp_min(0.01 * big)).max()))
                del ref
            line += f" | lwm_wgrad_bf16 {t_me:7.3f} ms {fl / t_me / 1e9:7.1f} TF/s  max rel err {err:.2e} (half a bf16 ulp: <= 2^-8 = 3.91e-03)"
        else:
            line += " | lwm_wgrad_bf16: shape not a multiple of 256"
        print(line, flush=True)
        del x, g, dw


if __name__ == "__main__":
    main()
