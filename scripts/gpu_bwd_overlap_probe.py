"""The two backward kernels of one attention layer (dK/dV and dQ: independent given the row statistics) on ONE stream against
TWO streams (the second kernel's workgroups fill the first one's tail), LWM-7B shape at S = 32768 and at ring-shard sizes.
    gpurun -- 'python scripts/gpu_bwd_overlap_probe.py'"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from lwm_amd import ops  # noqa: E402

H, D = 32, 128


def run(S, reps=8):
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: torch.randn(1, S, H, D, generator=g, device="cuda", dtype=torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()
    out, lse = ops.attn_fwd_block(q, k, v, causal=True)[:2]
    delta = ops.attn_bwd_delta(out, do, lse)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()

    def seq():
        ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True, dk=dk, dv=dv)
        ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True, dq=dq)

    def par():
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True, dk=dk, dv=dv)
        with torch.cuda.stream(side):
            ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True, dq=dq)
        ev2 = torch.cuda.Event(); ev2.record(side)
        main.wait_event(ev2)

    def par_rev():
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            ops.attn_bwd_dkdv_block(q, k, v, do, lse, delta, causal=True, dk=dk, dv=dv)
        ops.attn_bwd_dq_block(q, k, v, do, lse, delta, causal=True, dq=dq)
        ev2 = torch.cuda.Event(); ev2.record(side)
        main.wait_event(ev2)

    res = {}
    for name, fn in (("one stream", seq), ("two streams", par), ("two streams, dQ first", par_rev), ("one stream again", seq)):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps
        chk = float(dq.float().abs().sum() + dk.float().abs().sum() + dv.float().abs().sum())
        print(f"S={S:6d} {name:24s} {res[name]:8.3f} ms   checksum {chk:.6e}", flush=True)


for S in (32768, 4096, 2048):
    run(S, reps=8 if S == 32768 else 40)
