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
