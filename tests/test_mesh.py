"""--mesh_dim parsing (lwm/train.py:35, tux.get_jax_mesh formats) and the sp process groups it implies."""
import os
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from lwm_amd import mesh as M


def test_parse_mesh_dim_formats():
    assert M.parse_mesh_dim("1,-1,1,1", 8) == dict(dp=1, fsdp=8, tp=1, sp=1)          # train.py default
    assert M.parse_mesh_dim("!1,1,1,8", 8) == dict(dp=1, fsdp=1, tp=1, sp=8)
    assert M.parse_mesh_dim("dp:2,sp:4,tp:1,fsdp:1", 8) == dict(dp=2, fsdp=1, tp=1, sp=4)
    assert M.parse_mesh_dim("1,1,2,-1", 8)["sp"] == 4
    for bad, n in (("1,1,1", 8), ("1,1,1,3", 8), ("-1,-1,1,1", 8), ("dp:1,sp:8", 8), ("1,3,1,-1", 8), ("0,1,1,8", 8)):
        with pytest.raises(ValueError):
            M.parse_mesh_dim(bad, n)


def test_sp_is_the_fastest_axis():
    mesh = M.parse_mesh_dim("2,1,2,2", 8)
    assert M.coords(mesh, 5) == dict(dp=1, fsdp=0, tp=0, sp=1)
    assert M.axis_ranks(mesh, 5, "sp") == [4, 5] and M.axis_ranks(mesh, 5, "tp") == [5, 7]
    assert M.axis_ranks(mesh, 5, "dp") == [1, 5]
    groups = {tuple(M.axis_ranks(mesh, r, "sp")) for r in range(8)}
    assert groups == {(0, 1), (2, 3), (4, 5), (6, 7)}            # consecutive ranks ring together


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch
        g = M.sp_group(M.parse_mesh_dim("1,2,1,2", world))
        t = torch.tensor([float(rank)])
        dist.all_reduce(t, group=g)
        q.put((rank, dist.get_world_size(g), t.item()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sp_groups_on_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in ps:
        p.start()
    res = dict((r, (n, v)) for r, n, v in (q.get(timeout=120) for _ in range(4)))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: (2, 1.0), 1: (2, 1.0), 2: (2, 5.0), 3: (2, 5.0)}


def test_blockwise_feedforward_remat_is_transparent():
    """blockwise_feedforward (lwm/llama.py:729-734): chunked == whole, and pre_remat (recompute the module in
    the backward, lwm/llama.py:673-678) changes neither values nor gradients."""
    import torch
    from lwm_amd.ringattention import blockwise_feedforward
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.SiLU(), torch.nn.Linear(64, 16))
    x = torch.randn(2, 96, 16, requires_grad=True)
    outs = {}
    for name, kw in (("whole", dict(chunk_size=None, pre_remat=False)), ("chunked", dict(chunk_size=32, pre_remat=False)),
                     ("remat", dict(chunk_size=32, pre_remat=True))):
        for p in mlp.parameters():
            p.grad = None
        x.grad = None
        y = blockwise_feedforward(mlp, x, **kw)
        y.square().sum().backward()
        outs[name] = (y.detach().clone(), x.grad.clone(), mlp[0].weight.grad.clone())
    for name in ("chunked", "remat"):
        for a, b in zip(outs[name], outs["whole"]):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5), name
    with pytest.raises(ValueError):
        blockwise_feedforward(mlp, x, chunk_size=40)
