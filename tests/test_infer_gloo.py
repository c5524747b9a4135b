"""ringattention_inference and the sharded KV-cache update on CPU: world_size 2
and 4 gloo processes run the product drivers (lwm_amd/ring.py: ring_inference,
cache_update) with the per-block kernels replaced by the oracle stand-in."""
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


def _data(world):
    g = torch.Generator().manual_seed(0)
    B, H, D = 2, 2, 16
    max_len = 24 * world
    P = 8 * world                       # prompt rows written by the sharded prefill update
    k_new = torch.randn(B, P, H, D, generator=g).to(torch.bfloat16)
    v_new = torch.randn(B, P, H, D, generator=g).to(torch.bfloat16)
    k_dec = torch.randn(B, 1, H, D, generator=g).to(torch.bfloat16)
    v_dec = torch.randn(B, 1, H, D, generator=g).to(torch.bfloat16)
    q_dec = torch.randn(B, 1, H, D, generator=g).to(torch.bfloat16)
    q_pre = torch.randn(B, P, H, D, generator=g).to(torch.bfloat16)
    am = torch.ones(B, max_len, dtype=torch.bool)
    am[:, 2:4] = False
    return B, H, D, max_len, P, k_new, v_new, k_dec, v_dec, q_dec, q_pre, am


def _worker(rank, world, port, start, q_out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lwm_amd.ring import TorchRingComm
        from lwm_amd.ringattention import concatenate_to_cache, ringattention_inference
        from tests._standin import OracleBlockOps
        B, H, D, max_len, P, k_new, v_new, k_dec, v_dec, q_dec, q_pre, am = _data(world)
        c, p = max_len // world, P // world
        comm = TorchRingComm(None)
        ck = torch.zeros(B, c, H, D, dtype=torch.bfloat16)
        cv = torch.zeros(B, c, H, D, dtype=torch.bfloat16)
        kw = dict(block_ops=OracleBlockOps, comm=comm)
        # prefill: every rank contributes its P/n rows, landing at global [start, start+P)
        idx = concatenate_to_cache(ck, cv, k_new[:, rank * p:(rank + 1) * p], v_new[:, rank * p:(rank + 1) * p],
                                   start, **kw)
        # decode step: replicated single row, only the owner writes
        idx = concatenate_to_cache(ck, cv, k_dec, v_dec, idx, **kw)
        K = max_len
        mask_dec = (torch.arange(K)[None, None, None, :] <= (idx - 1)) & am[:, None, None, :]
        out_dec = ringattention_inference(q_dec, ck, cv, mask_dec, **kw)
        # short prefill-style block: queries sharded over the group, arbitrary mask
        g = torch.Generator().manual_seed(5)
        mask_pre = (torch.rand(B, 1, P, K, generator=g) > 0.3)
        out_pre = ringattention_inference(q_pre[:, rank * p:(rank + 1) * p], ck, cv,
                                          mask_pre[:, :, rank * p:(rank + 1) * p], **kw)
        q_out.put((rank, idx, ck.float().numpy(), cv.float().numpy(), out_dec.float().numpy(),
                   out_pre.float().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,start", [(2, 0), (2, 5), (4, 17)])
def test_cache_and_inference_match_single_device(world, start):
    from oracle import attention_ref as R
    ctx = mp.get_context("spawn")
    qout = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, start, qout)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([qout.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    B, H, D, max_len, P, k_new, v_new, k_dec, v_dec, q_dec, q_pre, am = _data(world)
    # single-device cache
    ck = torch.zeros(B, max_len, H, D)
    cv = torch.zeros(B, max_len, H, D)
    ck[:, start:start + P], cv[:, start:start + P] = k_new.float(), v_new.float()
    ck[:, start + P], cv[:, start + P] = k_dec[:, 0].float(), v_dec[:, 0].float()
    assert all(r[1] == start + P + 1 for r in res)
    got_k = np.concatenate([r[2] for r in res], axis=1)
    got_v = np.concatenate([r[3] for r in res], axis=1)
    assert np.array_equal(got_k, ck.numpy()) and np.array_equal(got_v, cv.numpy())
    mask_dec = R.decode_mask(B, 1, max_len, start + P, am.numpy())
    ro, _ = R.dense_attention(q_dec.float().numpy(), ck.numpy(), cv.numpy(), causal=False, dense_mask=mask_dec)
    for r in res:                       # decode output is replicated
        assert np.abs(r[4] - ro).max() / np.abs(ro).max() < 1e-2
    g = torch.Generator().manual_seed(5)
    mask_pre = (torch.rand(B, 1, P, max_len, generator=g) > 0.3)[:, 0].numpy()
    rp, _ = R.dense_attention(q_pre.float().numpy(), ck.numpy(), cv.numpy(), causal=False, dense_mask=mask_pre)
    got_pre = np.concatenate([r[5] for r in res], axis=1)
    assert np.abs(got_pre - rp).max() / np.abs(rp).max() < 1e-2
    # and the f32 ring restatement of ringattention_inference agrees with the dense oracle
    rr = R.ring_inference(q_dec.float().numpy(), ck.numpy(), cv.numpy(), mask_dec, ring=world)
    assert np.abs(rr - ro).max() < 1e-5
