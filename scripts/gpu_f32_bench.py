"""Timing of the float32 attention kernels (csrc/attn_f32.h) against their roofline (f32 MFMA, 157.3 TF/s dense):
algorithmic FLOPs 2 / 5 units of 2 S^2 d_model (causal), HIP events over 5 launches.  scripts/gpu_f32_check.sh."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwm_amd import ops  # noqa: E402


def ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {}
for S in (4096, 8192):
    H = 32
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v, do = (torch.randn(1, S, H, 128, device="cuda", generator=g) for _ in range(4))
    o, lse = ops.attn_fwd_block(q, k, v, causal=True)
    delta = ops.attn_bwd_delta(o, do, lse)
    unit = S * S * 4096.0      # causal GEMM unit: 2 * S^2/2 * d_model
    t_f = ms(lambda: ops.attn_fwd_block(q, k, v, causal=True))
    t_q = ms(lambda: ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True))
    t_k = ms(lambda: ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True))
    out[S] = dict(fwd_ms=round(t_f, 3), dq_ms=round(t_q, 3), dkdv_ms=round(t_k, 3),
                  fwd_tflops=round(2 * unit / t_f / 1e9, 1), bwd_algorithmic_tflops=round(5 * unit / (t_q + t_k) / 1e9, 1),
                  executed_tflops=dict(fwd=round(2 * unit / t_f / 1e9, 1), dq=round(3 * unit / t_q / 1e9, 1),
                                       dkdv=round(4 * unit / t_k / 1e9, 1)),
                  frac_of_f32_mfma_peak_157=round(7 * unit / (t_f + t_q + t_k) / 1e9 / 157.3, 3))
print(json.dumps({"f32_attention_32_heads": out}))
