"""The contract of the sequence-parallel operator surface where a silent wrong answer used to be possible:
which positions a rank's rows ARE when the caller brings its own position_ids, the ownership rule of a ProcessGroup that
was never bound, the loss normalisation on replicated tokens, and the first-contact watchdog of the C ring driver.
world_size-2 gloo processes on CPU."""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q_out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("LWM_SP_LAYOUT", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lwm_amd import ringattention as RA
        from lwm_amd.llama import LLaMAForCausalLM
        from lwm_amd.llama_ops import _row_weights
        res = {}
        ids = torch.zeros(1, 8, dtype=torch.int64)
        own = torch.arange(8, dtype=torch.int32)[None] + 8 * rank
        # 1. defaulted rule + the caller's own position_ids: refused (nobody said which positions the rows are)
        RA.set_sp_group(dist.group.WORLD)
        assert not RA.sp_layout_is_explicit("sp") and RA.sp_layout("sp", 8) == "zigzag"
        try:
            LLaMAForCausalLM._ring_position_ids(ids, own, None)
            res["refused"] = False
        except ValueError as e:
            res["refused"] = "layout" in str(e)
        # ... accepted once the rule is named, per call or on the axis; derived positions follow the named rule
        n, pos = LLaMAForCausalLM._ring_position_ids(ids, own, None, layout="contiguous")
        res["per_call"] = n == world and torch.equal(pos, own)
        RA.set_sp_group(dist.group.WORLD, layout="contiguous")
        assert RA.sp_layout_is_explicit("sp")
        res["named"] = torch.equal(LLaMAForCausalLM._ring_position_ids(ids, own, None)[1], own)
        res["derived_contiguous"] = LLaMAForCausalLM._ring_position_ids(ids, None, None)[1][0].tolist()
        RA.set_sp_group(dist.group.WORLD, layout="zigzag")
        res["derived_zigzag"] = LLaMAForCausalLM._ring_position_ids(ids, None, None)[1][0].tolist()
        # 2. the bound group handed over as an OBJECT follows the bound rule; a group nobody bound gets the reference's
        res["bound_object"] = RA.sp_layout(dist.group.WORLD, 8)
        other = dist.new_group(list(range(world)))
        res["unbound_object"] = RA.sp_layout(other, 8)
        # 3. loss weights: sharded rows sum the valid count over the ring, replicated tokens must not
        valid = torch.ones(1, 8)
        _, w_sh = _row_weights(valid, 1, 8, "cpu", sp_sharded=True)
        _, w_rep = _row_weights(valid, 1, 8, "cpu", sp_sharded=False)
        res["w_sharded"], res["w_replicated"] = float(w_sh[0, 0]), float(w_rep[0, 0])
        q_out.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_positions_layout_and_loss_contract_over_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    qout = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, qout)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(qout.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        r = got[rank]
        assert r["refused"] is True and r["per_call"] and r["named"]
        assert r["derived_contiguous"] == list(range(8 * rank, 8 * rank + 8))               # lwm/llama.py:560-562
        assert r["derived_zigzag"] == list(range(4 * rank, 4 * rank + 4)) + list(range(4 * (3 - rank), 4 * (3 - rank) + 4))
        assert r["bound_object"] == "zigzag" and r["unbound_object"] == "contiguous"
        assert abs(r["w_sharded"] - 1.0 / 16) < 1e-9 and abs(r["w_replicated"] - 1.0 / 8) < 1e-9


def test_first_contact_watchdog_turns_a_hang_into_exit_75():
    """a guarded region that never finishes ends the process with the diagnosis, not a hang; a finished one disarms"""
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from lwm_amd.ring_c import first_contact\n"
            "with first_contact('a quick stage', 5):\n    pass\n"
            "with first_contact('the stuck stage', 0.5):\n    time.sleep(30)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 75, (p.returncode, p.stderr[-500:])
    assert "the stuck stage did not finish" in p.stderr and "LWM_RING_DRIVER=python" in p.stderr
    assert "a quick stage" not in p.stderr
    code0 = ("import sys; sys.path.insert(0, %r)\nfrom lwm_amd.ring_c import first_contact\nimport time\n"
             "with first_contact('off', 0):\n    time.sleep(0.2)\nprint('done')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code0], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "done" in p.stdout
