"""Randomised driver-equivalence stress: random grid shapes / dtypes / source sets / WENO on-off, solved with the
overlapping-sweeps driver (mode 2, default) with and without exact skipping, the per-sweep persistent driver (mode 1)
with skipping and the launch-per-tile driver (mode 0, first-order stage only); all fields must be bit-identical.  usage: fuzz_modes.py <seconds> [seed]"""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_ok = 0
while time.time() < t_end:
    dim = 3 if rng.random() < 0.75 else 2
    dt = np.float32 if rng.random() < 0.6 else np.float64
    weno = int(rng.random() < 0.35)
    ns = int(rng.integers(1, 6))
    nthr = int(rng.integers(1, ns + 1))
    if dim == 3:
        nn = tuple(int(v) for v in rng.integers(18, 120, 3))
        if np.prod(nn) > 900000: continue
    else:
        nn = tuple(int(v) for v in rng.integers(40, 700, 2))
    dx = float(rng.choice([0.25, 0.5, 1.0]))
    axes = [np.arange(n) * dx + float(np.round(rng.uniform(-5, 5) * 8) / 8) for n in nn]   # exactly representable in fp32
    rough = rng.random() < 0.5
    s = rng.uniform(0.3, 1.0, nn) if rough else 1.0 / (1.0 + 0.05 * np.add.outer(np.zeros(nn[:-1]), axes[-1] - axes[-1][0]))
    src = np.column_stack([rng.uniform(a[1], a[-2], ns) for a in axes])
    if rng.random() < 0.3: src[0] = [a[int(rng.integers(1, a.size - 1))] for a in axes]   # on a node
    rcv = np.column_stack([rng.uniform(a[1], a[-2], ns) for a in axes])
    os.environ['TTCR_FSM_PAIR'] = '1' if rng.random() < 0.5 else '0'   # (default: pairs for big batches only)
    fields = {}
    # (driver, exact skipping): the whole-iteration launch with the skipping scheduler is the reference of the comparison
    for mode, skip in ((2, 1), (2, 0), (1, 1), (0, 0)):
        kw = dict(n_threads=nthr, cell_slowness=0, method='FSM', weno=weno, dtype=dt, maxit=8 if weno else 50)
        g = ttcr_amd.Grid3d(*axes, tt_from_rp=0, **kw) if dim == 3 else ttcr_amd.Grid2d(*axes, **kw)
        g.set_option('mode', mode)
        g.set_option('skip', skip)
        tt = g.raytrace(src, rcv, slowness=s)
        fields[(mode, skip)] = (tt, [g.get_grid_traveltimes(k).copy() for k in range(nthr)], [(g.get_niter(k), g.get_niterw(k)) for k in range(nthr)])
        del g
    ref = fields[(2, 1)]
    for mode in ((2, 0), (1, 1), (0, 0)):
        assert fields[mode][2] == ref[2], (nn, dt, weno, ns, nthr, mode, fields[mode][2], ref[2])
        assert np.array_equal(fields[mode][0], ref[0]), (nn, dt, weno, ns, nthr, mode)
        for a, b in zip(fields[mode][1], ref[1]):
            assert np.array_equal(a, b), (nn, dt, weno, ns, nthr, mode, float(np.max(np.abs(a - b))))
    n_ok += 1
print(f"fuzz: {n_ok} random configurations; whole-iteration launch with / without exact skipping, per-sweep launches with skipping, tile launches: bit-identical", flush=True)
