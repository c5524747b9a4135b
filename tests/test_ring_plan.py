"""Pure-Python checks of the schedules' bookkeeping (lwm_amd/ring.py): which (query segment, key segment)
pairs are launched, which key segments travel under the mesh schedule, for every rank pair of every ring
size and both ownership layouts -- against a brute-force element-wise visibility table."""
import itertools

import numpy as np
import pytest

from lwm_amd.ring import SeqLayout, _fwd_plan, _needed_ksegs, pair_visible


def _elementwise_visible(qseg, kseg):
    _, ql, qg = qseg
    _, kl, kg = kseg
    qpos = np.arange(qg, qg + ql)[:, None]
    kpos = np.arange(kg, kg + kl)[None, :]
    return bool((kpos <= qpos).any())


@pytest.mark.parametrize("kind,n", [("contiguous", 1), ("contiguous", 3), ("contiguous", 8), ("zigzag", 2), ("zigzag", 4),
                                    ("zigzag", 8)])
def test_pairs_and_transfers_match_bruteforce(kind, n):
    S = 16 * 2 * n
    lay = SeqLayout(kind, n, S)
    # ownership is a partition of the sequence
    owned = np.concatenate([lay.global_index(r).numpy() for r in range(n)])
    assert sorted(owned.tolist()) == list(range(S))
    for r, s in itertools.product(range(n), range(n)):
        qs_, ks_ = lay.segments(r), lay.segments(s)
        for qs, ks in itertools.product(qs_, ks_):
            assert pair_visible(qs, ks, True) == _elementwise_visible(qs, ks)
            assert pair_visible(qs, ks, False) is True
        need = [ki for ki, ks in enumerate(ks_) if any(_elementwise_visible(qs, ks) for qs in qs_)]
        assert _needed_ksegs(lay, r, s, True) == need
        assert _needed_ksegs(lay, r, s, False) == list(range(len(ks_)))
    for r in range(n):
        plan = _fwd_plan(lay, r, n, True)
        launched = {(t, qi, ki) for t, qi, ki in plan}
        assert len(launched) == len(plan)                        # nothing launched twice
        expect = {(t, qi, ki) for t in range(n) for qi, qs in enumerate(lay.segments(r))
                  for ki, ks in enumerate(lay.segments((r - t) % n)) if _elementwise_visible(qs, ks)}
        assert launched == expect                                # every visible pair exactly once


def test_zigzag_balances_causal_work_and_mesh_ships_three_quarters():
    n, S = 8, 8 * 2 * 64
    work = {}
    for kind in ("contiguous", "zigzag"):
        lay = SeqLayout(kind, n, S)
        per_rank = []
        for r in range(n):
            idx = lay.global_index(r).numpy()
            per_rank.append(int((idx + 1).sum()))                # visible keys per query row, summed
        work[kind] = max(per_rank) / (sum(per_rank) / n)
    assert work["contiguous"] > 1.8 and work["zigzag"] < 1.01    # (n - 1/2)/(n/2) vs balanced
    lay = SeqLayout("zigzag", n, S)
    sent = sum(len(_needed_ksegs(lay, dst, r, True)) for r in range(n) for dst in range(n) if dst != r)
    assert sent == 0.75 * (2 * n * (n - 1))                      # of the 2 segments x (n-1) peers a ring moves


def test_ownership_tables_partition_balance_and_plan_like_the_product_library():
    """SeqLayout("table") / balanced_layout: a partition of the sequence with P chunks per rank; for ONE document the
    4-chunk table balances the causal triangle as zigzag does; for a packed batch it beats the best pairing of half-chunks;
    and the bytes the C driver plans for a table (lwm_ring_planned_bytes_table, pure geometry) are what the visibility rule
    of this module gives -- for a zigzag-shaped table exactly what the zigzag layout plans."""
    import ctypes as C
    from lwm_amd import _capi
    from lwm_amd._lib import lib
    from lwm_amd.ring import balanced_layout
    n, S = 8, 8 * 4 * 256
    docs = [S // 16, S // 4 + 300, S // 8 - 300, S // 2, S // 16]
    for lens in (None, docs):
        lay = balanced_layout(n, S, lens, chunks_per_rank=4)
        assert lay.kind == "table" and len(lay.owner) == 4 * n and all(lay.owner.count(r) == 4 for r in range(n))
        owned = np.concatenate([lay.global_index(r).numpy() for r in range(n)])
        assert sorted(owned.tolist()) == list(range(S))
        starts = np.cumsum([0] + (lens or [S])[:-1])
        w = np.arange(S) - np.repeat(starts, lens or [S]) + 1.0
        load = [w[lay.global_index(r).numpy()].sum() for r in range(n)]
        assert max(load) / np.mean(load) < (1.02 if lens is None else 1.06), load
        if lens is not None:
            zz = SeqLayout("zigzag", n, S)
            zload = [w[zz.global_index(r).numpy()].sum() for r in range(n)]
            assert max(zload) / np.mean(zload) > 1.3
        # segments: one per chunk, ascending positions, local rows in that order
        for r in range(n):
            segs = lay.segments(r)
            assert [s_[0] for s_ in segs] == [i * (S // (4 * n)) for i in range(4)] and all(a[2] < b[2] for a, b in zip(segs, segs[1:]))
    L = lib()
    c, H, D = S // n, 2, 128
    zig = [j if j < n else 2 * n - 1 - j for j in range(2 * n)]
    tab = (C.c_int32 * len(zig))(*zig)
    for r in range(n):
        for bwd in (0, 1):
            assert L.lwm_ring_planned_bytes_table(tab, len(zig), n, r, c, H, D, bwd) == \
                L.lwm_ring_planned_bytes(_capi.RING_LAYOUT["zigzag"], _capi.RING_SCHEDULE["direct"], n, r, 1, c, H, D, 1, bwd)
    lay = balanced_layout(n, S, docs, chunks_per_rank=4)
    tab = (C.c_int32 * len(lay.owner))(*lay.owner)
    cs = S // len(lay.owner)
    for r in range(n):
        last = {q: max(j for j, o in enumerate(lay.owner) if o == q) for q in range(n)}
        fetch = sum(2 * cs * H * D * 2 for j, o in enumerate(lay.owner) if o == r for q in range(n) if q != r and j < last[q])
        back = sum(2 * cs * H * D * 4 for j, o in enumerate(lay.owner) if o != r and j < last[r])
        assert L.lwm_ring_planned_bytes_table(tab, len(lay.owner), n, r, c, H, D, 0) == fetch
        assert L.lwm_ring_planned_bytes_table(tab, len(lay.owner), n, r, c, H, D, 1) == fetch + back
    bad = (C.c_int32 * 16)(*([0] * 16))
    assert L.lwm_ring_planned_bytes_table(bad, 16, n, 0, c, H, D, 0) == -1
