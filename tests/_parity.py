"""Shared parity check for the attention kernels (bf16 operands, f32 accumulation) against the fp64
oracle.  Stated tolerances (north_star: "within a stated fp tolerance"):

  global : max|err| <= 8e-3 * max|ref|   and cosine >= 0.9999
  per row: for every (b, s, h) row of D values,
           max|err_row| <= 2.5e-2 * max(max|ref_row|, floor * max|ref|),  floor = 0.02 (out, dk, dv), 0.25 (dq)
           -- a wrong SMALL-magnitude row fails this one although it passes a max-normalised test
  lse    : max|err| <= 2e-3 (absolute, natural log)

Why dq has its own floor: ds = p * (dp - delta) with delta = rowsum(dO * O) taken from the SAVED output,
which is bf16 (the reference saves `out` cast to v.dtype too).  In rows that see
few keys (the first rows of a sequence or of a packed document) dp - delta cancels almost completely:
the exact dq is ~0 while the bf16 rounding of O leaves |dO|.|O|.2^-9 in delta -- an error that scales
with the operands, not with the row's own (vanishing) gradient.  Measured on MI355X (round 2,
profiles/r02_parity_stats.json): worst row error / max(row scale, 2 % of global) = 0.0068 (out),
0.0096 (dk), 0.0060 (dv) but 0.144 (dq, first row of a document); against the 25 % floor dq is 0.012.

Every call records what it measured; the session writes the worst figures per quantity to
gpurun_out/parity_stats.json (tests/conftest.py) so the bounds above can be compared with what the
hardware actually produced."""
import numpy as np

TOL, ROW_TOL, ROW_FLOOR, ROW_FLOOR_DQ, COS = 8e-3, 2.5e-2, 0.02, 0.25, 0.9999
STATS = []   # (name, global_rel_err, row_rel_err, cosine)


def check(name, got, ref, tol=TOL, row_tol=ROW_TOL):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    gmax = max(np.abs(ref).max(), 1e-9)
    diff = np.abs(got - ref)
    err = diff.max() / gmax
    cos = (got * ref).sum() / max(np.sqrt((got ** 2).sum() * (ref ** 2).sum()), 1e-30)
    floor = ROW_FLOOR_DQ if name.startswith("dq") else ROW_FLOOR
    row = (diff.max(axis=-1) / np.maximum(np.abs(ref).max(axis=-1), floor * gmax)).max()
    STATS.append((name, float(err), float(row), float(cos)))
    assert err <= tol, f"{name}: max|err| / max|ref| = {err:.3e} > {tol}"
    assert row <= row_tol, f"{name}: worst row error relative to its own scale = {row:.3e} > {row_tol}"
    assert cos >= COS, f"{name}: cosine {cos}"


def summary():
    out = {}
    for name, err, row, cos in STATS:
        key = name.split(" ")[0].split("(")[0]
        s = out.setdefault(key, {"n": 0, "max_global_rel_err": 0.0, "max_row_rel_err": 0.0, "min_cosine": 1.0})
        s["n"] += 1
        s["max_global_rel_err"] = max(s["max_global_rel_err"], err)
        s["max_row_rel_err"] = max(s["max_row_rel_err"], row)
        s["min_cosine"] = min(s["min_cosine"], cos)
    return {"bounds": {"global": TOL, "row": ROW_TOL, "row_floor": ROW_FLOOR, "row_floor_dq": ROW_FLOOR_DQ, "cosine": COS},
            "measured": out}
