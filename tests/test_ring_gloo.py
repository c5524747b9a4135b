"""The N>1 path on CPU: world_size-2 (and 4) gloo processes run the product ring
driver (lwm_amd/ring.py: schedule, carries, K/V and dK/dV rotation, zigzag
ownership) with the per-block kernels replaced by the oracle stand-in, and the
result must equal single-device dense attention (ring n == ring 1)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, layout_kind, causal, packed, schedule, q_out, B=1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lwm_amd.ring import SeqLayout, TorchRingComm, ring_attention
        from tests._standin import OracleBlockOps
        torch.manual_seed(0)
        S, H, D = 64 * world, 2, 16
        q, k, v, do = (torch.randn(B, S, H, D).to(torch.bfloat16) for _ in range(4))
        seg = kv = None
        if packed:
            seg = torch.zeros(B, S, dtype=torch.int32)
            seg[:, S // 3:] = 1
            seg[:, (2 * S) // 3 + 5:] = 2
            kv = torch.ones(B, S, dtype=torch.uint8)
            kv[:, 3:9] = 0
        lay = SeqLayout(layout_kind, world, S)
        idx = lay.global_index(rank)
        ql, kl, vl = (t[:, idx].clone().requires_grad_(True) for t in (q, k, v))
        out = ring_attention(ql, kl, vl, causal=causal, segment_ids=seg, key_valid=kv, layout=lay,
                             block_ops=OracleBlockOps, comm=TorchRingComm(None, schedule=schedule))
        out.backward(do[:, idx])
        q_out.put((rank, idx.numpy(), out.detach().float().numpy(), ql.grad.float().numpy(),
                   kl.grad.float().numpy(), vl.grad.float().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,layout_kind,causal,packed,schedule,B", [
    (2, "contiguous", True, False, "ring", 1),
    (2, "zigzag", True, True, "ring", 1),
    (2, "contiguous", False, True, "ring", 1),
    (4, "zigzag", True, False, "ring", 1),
    # the mesh schedule: direct fetch of the visible K/V segments, partial dK/dV returned to the owner
    (2, "zigzag", True, True, "mesh", 1),
    (4, "zigzag", True, True, "mesh", 1),
    (4, "contiguous", True, False, "mesh", 1),
    (3, "contiguous", False, True, "mesh", 1),
    (2, "zigzag", True, False, "mesh", 2),          # batch 2: strided segment views of the outputs
    (2, "zigzag", True, False, "ring", 2),
])
def test_ring_equals_single_device(world, layout_kind, causal, packed, schedule, B):
    from oracle import attention_ref as R
    ctx = mp.get_context("spawn")
    qout = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, layout_kind, causal, packed, schedule, qout, B))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [qout.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    S, H, D = 64 * world, 2, 16
    q, k, v, do = (torch.randn(B, S, H, D).to(torch.bfloat16).float().numpy() for _ in range(4))
    seg = kv = None
    if packed:
        seg = np.zeros((B, S), np.int32)
        seg[:, S // 3:] = 1
        seg[:, (2 * S) // 3 + 5:] = 2
        kv = np.ones((B, S), np.uint8)
        kv[:, 3:9] = 0
    kw = dict(causal=causal, seg_q=seg, seg_k=seg, key_valid=kv)
    ro, _ = R.dense_attention(q, k, v, **kw)
    rq, rk, rv = R.dense_attention_bwd(q, k, v, do, **kw)
    out = np.zeros_like(ro); dq = np.zeros_like(rq); dk = np.zeros_like(rk); dv = np.zeros_like(rv)
    for _, idx, o, gq, gk, gv in res:
        out[:, idx], dq[:, idx], dk[:, idx], dv[:, idx] = o, gq, gk, gv
    # the stand-in computes in f64 and rounds results to bf16 once
    for name, a, b in (("out", out, ro), ("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-9)
        assert err < 1e-2, (name, err)
