"""The C ring driver end to end between REAL processes: N python processes share the one GPU of a test box, bootstrap
over gloo (127.0.0.1), and exchange K/V blocks and f32 dK/dV partials through the library's IPC transport (peer
mailboxes mapped with hipIpcMemHandles, hipMemcpyAsync, hipStreamWriteValue32 / hipStreamWaitValue32) -- every piece of
the multi-process path except the xGMI wires: separate address spaces, interprocess visibility, the driver's events
against a transport that really is asynchronous.  Rank 0 checks the assembled result against the single-device driver
and the fp64 oracle (tests/_ipc_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(n, S, H, layout, schedule, packed, big, timeout=900):
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ipc_worker.py"), str(S), str(H), layout, schedule,
                                       str(int(packed)), str(int(big))], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.returncode for p in procs]
    if not all(c == 0 for c in codes) or "IPC_RING_OK" not in outs[0]:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        for r, o in enumerate(outs):
            with open(os.path.join(ROOT, "gpurun_out", f"ipc_fail_rank{r}.log"), "w") as f:
                f.write(o)
    assert all(c == 0 for c in codes), (codes, [o[-3000:] for o in outs])
    assert "IPC_RING_OK" in outs[0], outs[0][-2000:]
    return outs[0]


@pytest.mark.gpu
@pytest.mark.parametrize("n,layout,schedule,packed", [(2, "zigzag", "direct", True), (4, "contiguous", "ring", True)])
def test_ipc_ring_between_processes(n, layout, schedule, packed):
    _launch(n, 384 * 2 * n, 2, layout, schedule, packed, big=False)


@pytest.mark.gpu
def test_ipc_ring_ownership_table_between_processes():
    """an ownership table (4 chunks of 256 rows per rank, handed out by the packed documents' pair counts) between 4 real
    processes: the gathered form's fetch and partial return by chunk through real mailboxes, 8 messages per pair and group"""
    _launch(4, 4096, 2, "table", "direct", True, big=False)


@pytest.mark.gpu
def test_ipc_ring8_at_config3_shard_shape():
    """BASELINE configs[2]'s shard shape between 8 real processes: S = 131072, c = 16384 per rank, zigzag ownership, the
    direct schedule, packed documents; oracle windows at the end of every document (first, middle and last ranks)."""
    out = _launch(8, 131072, 2, "zigzag", "direct", True, big=True, timeout=1500)
    line = [l for l in out.splitlines() if "IPC_RING_OK" in l][-1]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ipc_ring8.txt"), "w") as f:
        f.write(line + "\n")
