"""Randomised check of the slot map behind the pairing by distance: random grids, slot counts, source counts (more sources
than slots: several rounds), max_batch, single-source calls on thread numbers in between -- a grid with option pair_sources
= 1 against one with 0: receiver traveltimes, per-thread fields, iteration counts and change histories must be identical.
usage: fuzz_pairing.py <seconds> [seed]"""
import sys, time, os
os.environ.setdefault('TTCR_FSM_PAIR', '1')   # the pair layout is the default of big batches only: asked for here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_ok = 0
while time.time() < t_end:
    nn = tuple(int(v) for v in rng.integers(18, 70, 3))
    dt = np.float32 if rng.random() < 0.7 else np.float64
    nthr = int(rng.integers(2, 12))
    ns = int(rng.integers(1, 3 * nthr))
    dx = 0.5
    axes = [np.arange(n) * dx for n in nn]
    s = rng.uniform(0.3, 1.0, nn)
    src = np.column_stack([rng.uniform(a[1], a[-2], ns) for a in axes])
    rcv = np.column_stack([rng.uniform(a[1], a[-2], ns) for a in axes])
    mb = int(rng.integers(1, nthr + 1)) if rng.random() < 0.3 else 0
    res = []
    for pair in (1, 0):
        g = ttcr_amd.Grid3d(*axes, n_threads=nthr, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=dt)
        g.set_slowness(s)
        g.set_option('pair_sources', pair)
        if mb: g.set_option('max_batch', mb)
        out = []
        tt = g.raytrace(src, rcv)
        out.append(tt)
        k = int(rng.integers(0, nthr)) if pair else k_keep   # a single source on a thread number, same choices for both grids
        if pair: k_keep, j_keep = k, int(rng.integers(0, ns))
        out.append(g.raytrace(src[j_keep:j_keep + 1], rcv[j_keep:j_keep + 1], thread_no=k_keep))
        ns2 = max(1, ns // 2)
        out.append(g.raytrace(src[:ns2][::-1], rcv[:ns2][::-1]))   # a second call with fewer sources, other order
        for t in range(nthr):
            out.append(g.get_grid_traveltimes(t).copy())
            out.append(np.array([g.get_niter(t)]))
            out.append(np.asarray(g.get_changes(t)[0]))
        res.append(out)
        del g
    for a, b in zip(*res):
        # (the change sums come from double atomics in launch order: equal to rounding, everything else bit for bit)
        same = np.allclose(a, b, rtol=1e-9, atol=0) if a.dtype == np.float64 and a.ndim == 1 and a.size and a.size < 64 and dt == np.float32 else np.array_equal(a, b)
        assert a.shape == b.shape and (same or np.allclose(a, b, rtol=1e-9, atol=0) and a.ndim == 1 and a.size < 64), (nn, dt, nthr, ns, mb)
    n_ok += 1
print(f"fuzz_pairing: {n_ok} random configurations; pairing by distance on / off: receivers, per-thread fields, iteration counts, change histories identical")
