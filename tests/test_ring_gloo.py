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
        if layout_kind == "balanced":      # a table: 4 chunks of 16 rows per rank, weighed by the packed documents' pairs
            from lwm_amd.ring import balanced_layout
            lay = balanced_layout(world, S, [S // 3, (2 * S) // 3 + 5 - S // 3, S - (2 * S) // 3 - 5] if packed else None,
                                  chunks_per_rank=4, align=16)
            assert lay.kind == "table" and len(lay.owner) == 4 * world
        else:
            lay = SeqLayout(layout_kind, world, S)
        idx = lay.global_index(rank)
        ql, kl, vl = (t[:, idx].clone().requires_grad_(True) for t in (q, k, v))
        launches = {"fwd": 0, "dq": 0, "dkdv": 0}

        class Counting(OracleBlockOps):
            @staticmethod
            def fwd(*a, **kw):
                launches["fwd"] += 1
                return OracleBlockOps.fwd(*a, **kw)

            @classmethod
            def bwd_dq(cls, *a, **kw):
                launches["dq"] += 1
                return super().bwd_dq(*a, **kw)

            @classmethod
            def bwd_dkdv(cls, *a, **kw):
                launches["dkdv"] += 1
                return super().bwd_dkdv(*a, **kw)

        out = ring_attention(ql, kl, vl, causal=causal, segment_ids=seg, key_valid=kv, layout=lay,
                             block_ops=Counting, comm=TorchRingComm(None, schedule=schedule))
        out.backward(do[:, idx])
        if schedule == "mesh" and causal and B == 1:
            # the gathered form: the local block + everything that arrived -- two launches per kernel, whatever n is
            assert all(v_ <= 2 for v_ in launches.values()), launches
        elif world > 2 or layout_kind != "contiguous":
            assert launches["fwd"] > 2, launches
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
    # an ownership table (balanced_layout): up to four runs of positions per rank, through both schedules
    (4, "balanced", True, True, "mesh", 1),
    (2, "balanced", True, False, "ring", 1),
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


def _surface_worker(rank, world, port, q_out, layout_env):
    """the operator surface above the driver: set_sp_group's ownership rule, sp_shard / sp_positions, `ringattention`
    following the bound rule without being told, the loss normalisation summed over the ring"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if layout_env:
        os.environ["LWM_SP_LAYOUT"] = layout_env
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lwm_amd import ringattention as RA
        from lwm_amd.llama_ops import _row_weights
        from lwm_amd.ring import SeqLayout, TorchRingComm, ring_attention
        from tests._standin import OracleBlockOps
        RA.set_sp_group(dist.group.WORLD)
        S, H, D = 64 * world, 2, 16
        kind = RA.sp_layout("sp", S // world)
        pos = RA.sp_positions(S // world)
        full = torch.arange(3 * S).reshape(3, S)
        shard = RA.sp_shard(full)
        # valid targets differ per rank: the loss weights must still be (valid / count over the WHOLE sequence) / B
        valid = torch.zeros(2, S)
        valid[0, : S // 4] = 1
        valid[1, 5:] = 1
        _, w = _row_weights(RA.sp_shard(valid), 2, S // world, "cpu")
        # ringattention() with NO layout argument must treat the local rows as sp_shard cut them
        torch.manual_seed(0)
        q, k, v = (torch.randn(1, S, H, D).to(torch.bfloat16) for _ in range(3))
        import lwm_amd.ring as ring_mod
        real = ring_mod.ring_attention
        seen = {}

        def spy(q_, k_, v_, **kw):
            seen["layout"] = kw.get("layout")
            return real(q_, k_, v_, causal=kw["causal"], segment_ids=kw["segment_ids"], key_valid=kw["key_valid"],
                        layout=kw["layout"], block_ops=OracleBlockOps, comm=TorchRingComm(None, schedule="mesh"))

        RA.ring_attention = spy
        out = RA.ringattention(RA.sp_shard(q), RA.sp_shard(k), RA.sp_shard(v), None, None, axis_name="sp",
                               blockwise_kwargs=dict(causal_block_size=1))
        q_out.put((rank, kind, pos.numpy(), shard.numpy(), w.numpy(), seen["layout"], out.float().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,layout_env", [(2, ""), (4, ""), (2, "contiguous")])
def test_operator_surface_follows_the_bound_ownership_rule(world, layout_env):
    from oracle import attention_ref as R
    from lwm_amd.ring import SeqLayout
    ctx = mp.get_context("spawn")
    qout = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_surface_worker, args=(r, world, port, qout, layout_env)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([qout.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    S, H, D = 64 * world, 2, 16
    want = layout_env or "zigzag"            # the default for more than one rank
    lay = SeqLayout(want, world, S)
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, S, H, D).to(torch.bfloat16).float().numpy() for _ in range(3))
    ro, _ = R.dense_attention(q, k, v, causal=True)
    valid = np.zeros((2, S), np.float32)
    valid[0, : S // 4] = 1
    valid[1, 5:] = 1
    out = np.zeros_like(ro)
    wsum = np.zeros(2)
    covered = []
    for rank, kind, pos, shard, w, seen_layout, o in res:
        idx = lay.global_index(rank).numpy()
        assert kind == want and seen_layout == want
        assert np.array_equal(pos, idx) and np.array_equal(shard, np.arange(3 * S).reshape(3, S)[:, idx])
        assert np.allclose(w, valid[:, idx] / (valid.sum(-1, keepdims=True) * 2))      # counts of the WHOLE rows
        wsum += w.sum(-1)
        out[:, idx] = o
        covered.append(idx)
    assert np.array_equal(np.sort(np.concatenate(covered)), np.arange(S))
    assert np.allclose(wsum, 0.5)            # the ranks' shares add up to the per-sequence mean / B
    assert np.abs(out - ro).max() / np.abs(ro).max() < 1e-2
