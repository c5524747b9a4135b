"""Per-CU timelines from scripts/micro/conv_prof's stamps: where the matrix pipe of a CU has no workgroup in its main loop.
usage: python scripts/conv_prof_report.py gpurun_out/conv_prof/*.bin"""
import sys
import numpy as np

for path in sys.argv[1:]:
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    t = a[:, :5]
    hw, xcc = a[:, 5], a[:, 6] & 0xF
    cu = ((xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))   # (xcc, se, sh, cu)
    P, M, E, S = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2]), (t[:, 4] - t[:, 3])
    print(f"== {path}: {len(a)} workgroups on {len(np.unique(cu))} CUs")
    for n, v in (("prologue (entry -> patch landed)", P), ("main loop", M), ("epilogue to stores issued", E), ("stores acknowledged", S)):
        print(f"   {n:34s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  median {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
    tot = cov1 = cov2 = gap = 0
    for c in np.unique(cu):
        w = t[cu == c]
        lo, hi = w[:, 0].min(), w[:, 4].max()
        ev = sorted([(x, 1) for x in w[:, 1]] + [(x, -1) for x in w[:, 2]])
        n = 0; last = lo; c0 = c1 = c2 = 0
        for x, d in ev:
            dt = x - last
            if n == 0: c0 += dt
            elif n == 1: c1 += dt
            else: c2 += dt
            n += d; last = x
        c0 += hi - last
        tot += hi - lo; cov1 += c1; cov2 += c2; gap += c0
    print(f"   per CU: time with 0 workgroups in the main loop {gap / tot:.3f}, with 1 {cov1 / tot:.3f}, with 2+ {cov2 / tot:.3f}")
    # a lone main loop runs at the full matrix rate; its length when alone vs when shared tells the in-loop efficiency
    print(f"   main-loop MFMA cycles per wave (ideal, at 64 cycles each): c128 73728 / c256 147456;  observed sum over the CU's slots / span: {M.sum() / tot:.3f} main loops in flight on average")


def rates(path):
    """least squares: work of one main loop = r_alone * (cycles alone on the CU) + r_shared * (cycles with another main loop)"""
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    t = a[:, :5]
    hw, xcc = a[:, 5], a[:, 6] & 0xF
    cu = ((xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))
    X = []
    gaps = []
    for c in np.unique(cu):
        w = t[cu == c]
        w = w[np.argsort(w[:, 0])]
        for i in range(len(w)):
            s, e = w[i, 1], w[i, 2]
            ov = 0
            for j in range(max(0, i - 3), min(len(w), i + 4)):
                if j != i:
                    ov += max(0, min(e, w[j, 2]) - max(s, w[j, 1]))
            X.append((e - s - ov, ov))
        # slot turnaround: a workgroup's entry minus the latest exit before it
        ends = np.sort(w[:, 4])
        for i in range(len(w)):
            k = np.searchsorted(ends, w[i, 0]) - 1
            if k >= 0:
                gaps.append(w[i, 0] - ends[k])
    X = np.array(X, dtype=np.float64)
    sol, *_ = np.linalg.lstsq(X, np.ones(len(X)), rcond=None)
    print(f"   {path}: a main loop's work = {1 / sol[0]:.0f} cycles alone, {1 / sol[1]:.0f} cycles shared;  alone {X[:, 0].mean():.0f} + shared {X[:, 1].mean():.0f} cycles on average")
    gaps = np.array(gaps)
    print(f"   entry minus the latest earlier exit on the CU: median {np.median(gaps):.0f}, p90 {np.percentile(gaps, 90):.0f} cycles")


for path in sys.argv[1:]:
    rates(path)
