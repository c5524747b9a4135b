"""One rank of tests/test_ring_ipc.py: a real process (gloo for the bootstrap only) driving the C ring driver over the
library's IPC transport on the GPU all ranks share.  argv: S H layout schedule packed(0/1) big(0/1)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    S, H, layout, schedule, packed, big = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from lwm_amd.ring import SeqLayout, SingleComm, ring_attention
    from lwm_amd.ring_c import CRing
    B, D = 1, 128
    g = torch.Generator().manual_seed(7)
    mk = lambda: torch.randn(B, S, H, D, generator=g).to(torch.bfloat16)
    q, k, v, do = mk(), mk(), mk(), mk()             # the same on every rank
    seg = None
    bounds = [0, (5 * S) // 16, (17 * S) // 32, (25 * S) // 32, S]
    if packed:
        seg = torch.bucketize(torch.arange(S), torch.tensor(bounds[1:-1]), right=True).to(torch.int32)[None].contiguous()
    if layout == "table":       # an ownership table from the document lengths: 4 chunks per rank, up to 8 messages per pair and group
        from lwm_amd.ring import balanced_layout
        lay = balanced_layout(world, S, [b - a for a, b in zip(bounds[:-1], bounds[1:])] if packed else None, chunks_per_rank=4)
        assert lay.kind == "table"
    else:
        lay = SeqLayout(layout, world, S)
    idx = lay.global_index(rank)
    c = S // world
    ring = CRing(dist.group.WORLD, transport="ipc", layout=lay if layout == "table" else layout, schedule=schedule,
                 ipc_slot_bytes=B * c * H * D * 4, ipc_slots=8)
    ql, kl, vl, dol = (t[:, idx].contiguous().cuda() for t in (q, k, v, do))
    segd = None if seg is None else seg.cuda()
    for _ in range(2):       # twice: message counters and slots carry over between calls
        out, lse = ring.forward(ql, kl, vl, causal=True, segment_ids=segd)
        dq, dk, dv = ring.backward(ql, kl, vl, out, lse, dol, causal=True, segment_ids=segd)
    torch.cuda.synchronize()
    sent = ring.bytes_sent
    parts = [t.cpu().contiguous().view(torch.int32) for t in (out, dq, dk, dv)]   # (gloo moves int32, not bf16)
    gathered = [[torch.empty_like(p) for _ in range(world)] if rank == 0 else None for p in parts]
    for p, gl in zip(parts, gathered):
        dist.gather(p, gl, dst=0)
    sent_all = [None] * world
    dist.all_gather_object(sent_all, sent)
    ring.close()
    if rank == 0:
        full = []
        for gl in gathered:
            t = torch.zeros(B, S, H, D, dtype=torch.bfloat16)
            for r in range(world):
                t[:, lay.global_index(r)] = gl[r].view(torch.bfloat16)
            full.append(t)
        # the single-device driver on the same data (same kernels)
        q1, k1, v1 = (t.cuda().requires_grad_(True) for t in (q, k, v))
        o1 = ring_attention(q1, k1, v1, causal=True, segment_ids=segd, comm=SingleComm())
        o1.backward(do.cuda())
        for name, a, b in zip(("out", "dq", "dk", "dv"), full, (o1.detach(), q1.grad, k1.grad, v1.grad)):
            err = ((a.float() - b.float().cpu()).abs().max() / b.float().abs().max()).item()
            assert err <= 8e-3, (name, err)
        from oracle import attention_ref as R
        from tests._parity import check, check_dq
        f = lambda t, rows, h: t[:, rows, h:h + 1].float().numpy()
        out, dq, dk, dv = full
        if not big:
            fa = lambda t: t.float().numpy()
            sg = None if seg is None else seg.numpy()
            ro, _ = R.dense_attention(fa(q), fa(k), fa(v), causal=True, seg_q=sg, seg_k=sg)
            rq, rk, rv, rqx = R.dense_attention_bwd(fa(q), fa(k), fa(v), fa(do), causal=True, seg_q=sg, seg_k=sg, out_saved=fa(out))
            for name, a, b in zip(("out", "dk", "dv"), (out, dk, dv), (ro, rk, rv)):
                check(f"{name} ipc ring n={world}", fa(a), b)
            check_dq(f"dq ipc ring n={world}", fa(dq), rq, rqx)
        else:
            # windows: the last rows of every document against that document alone (complete for out / dq of the rows and
            # dk / dv of the keys in the window)
            for i, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
                h, qa, w0 = i % H, b - 256, b - 256      # (the window's own rows are all the queries its out / dq / dk / dv need)
                rows, keys, win = slice(qa, b), slice(a, b), slice(w0, b)
                ro, _ = R.dense_attention(f(q, rows, h), f(k, keys, h), f(v, keys, h), causal=True, q_start=qa - a)
                rq, rk, rv, rqx = R.dense_attention_bwd(f(q, rows, h), f(k, keys, h), f(v, keys, h), f(do, rows, h), causal=True, q_start=qa - a,
                                                        out_saved=f(out, rows, h))
                check(f"out ipc doc {i}", f(out, win, h), ro[:, w0 - qa:])
                check_dq(f"dq ipc doc {i}", f(dq, win, h), rq[:, w0 - qa:], rqx[:, w0 - qa:])
                check(f"dk ipc doc {i}", f(dk, win, h), rk[:, w0 - a:])
                check(f"dv ipc doc {i}", f(dv, win, h), rv[:, w0 - a:])
        print(f"IPC_RING_OK n={world} S={S} c={c} H={H} {layout} {schedule} packed={packed} bytes_sent_per_rank={sent_all}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
