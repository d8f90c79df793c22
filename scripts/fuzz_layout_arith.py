"""Randomised check of the round-6 host logic: grids in the pairing window (TTCR_FSM_PAIR_UNITS lowered so that small grids sit in it) whose field
layout follows the model from call to call (option pair_layout = -1) against the same grid with the layout pinned to pairs -- random models
(smooth / rough, changed between calls), calls that restart every slot and calls that restart one, both arithmetic modes: receiver traveltimes,
per-thread fields, iteration counts must be identical between the two grids (same arithmetic), and the tolerance-grade fields within 1e-5 s RMS of
the default mode's.  Every few cases the parallel form of the reference's stopping sum (stopping_rule = 1) against its one-chain form (= 2) on a
rough model run to convergence: same iteration counts, same fields.   usage: fuzz_layout_arith.py <seconds> [seed]"""
import sys, time, os
os.environ.setdefault('TTCR_FSM_PAIR_UNITS', '60')
os.environ.setdefault('TTCR_FSM_LAYOUT_LO', '0.75')
os.environ.setdefault('TTCR_FSM_LAYOUT_HI', '0.9')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_ok = n_switch = n_stop = 0
while time.time() < t_end:
    nn = tuple(int(v) for v in rng.integers(40, 100, 3))
    S = int(rng.integers(4, 9))
    dx = 0.5
    axes = [np.arange(n) * dx for n in nn]
    def model(kind):
        if kind == 0: return np.full(nn, float(rng.uniform(0.2, 1.0)), dtype=np.float32)
        if kind == 1: return np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * axes[2])).astype(np.float32), nn))
        return rng.uniform(0.25, 1.0, nn).astype(np.float32)
    grids = []
    for lay in (-1, 1):
        g = ttcr_amd.Grid3d(*axes, n_threads=S, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
        g.set_option('pair_layout', lay); g.set_option('skip', 1)
        grids.append(g)
    kernels = []
    for call in range(int(rng.integers(3, 7))):
        if call == 0 or rng.random() < 0.4:
            m = model(int(rng.integers(0, 3)))
            for g in grids: g.set_slowness(m)
        ar = int(rng.random() < 0.4)
        for g in grids: g.set_option('arith', ar)
        full = rng.random() < 0.7
        if full:
            src = np.column_stack([rng.uniform(0, a[-1], S) for a in axes])
            rcv = np.column_stack([rng.uniform(0, a[-1], S) for a in axes])
            out = [g.raytrace(src, rcv) for g in grids]
        else:
            th = int(rng.integers(0, S))
            src = np.array([[rng.uniform(0, a[-1]) for a in axes]]); rcv = np.array([[rng.uniform(0, a[-1]) for a in axes]])
            out = [g.raytrace(src, rcv, thread_no=th) for g in grids]
        kernels.append(grids[0].last_kernel())
        assert np.array_equal(out[0], out[1]), ('receivers', nn, S, call)
        for t in range(S):
            assert grids[0].get_niter(t) == grids[1].get_niter(t), ('niter', nn, S, call, t)
            assert np.array_equal(grids[0].get_grid_traveltimes(t), grids[1].get_grid_traveltimes(t)), ('field', nn, S, call, t, grids[0].last_kernel(), grids[1].last_kernel())
        if ar and full:   # the tolerance-grade fields against the default mode on the pinned grid
            f1 = [np.array(grids[1].get_grid_traveltimes(t), dtype=np.float64) for t in range(S)]
            grids[1].set_option('arith', 0); grids[1].raytrace(src, rcv)
            for t in range(S):
                d = f1[t] - grids[1].get_grid_traveltimes(t)
                assert np.sqrt(np.mean(d * d)) <= 1e-5, ('rms', nn, S, call, t, float(np.sqrt(np.mean(d * d))))
            grids[0].set_option('arith', 0); grids[0].raytrace(src, rcv)
    n_switch += len({(',1,2,true' in k) for k in kernels}) == 2
    n_ok += 1
    if n_ok % 3 == 0:   # the reference's stopping sum, parallel against one chain (grids above 2^24 nodes take snapshots by prediction; small ones always)
        m = rng.uniform(0.25, 1.0, nn).astype(np.float32)
        src = np.column_stack([rng.uniform(0, a[-1], 2) for a in axes]); rcv = np.zeros((2, 3))
        res = []
        for rule in (1, 2):
            g = ttcr_amd.Grid3d(*axes, n_threads=2, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
            g.set_slowness(m); g.set_option('stopping_rule', rule); g.raytrace(src, rcv)
            res.append(([g.get_niter(t) for t in range(2)], [np.array(g.get_grid_traveltimes(t)) for t in range(2)], g.stopping_stats()))
        assert res[0][0] == res[1][0] and all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1])), ('stopping rule', nn, res[0][0], res[1][0])
        n_stop += res[0][2]['reference_sums'] > 0
print(f"fuzz_layout_arith: {n_ok} cases identical ({n_switch} of them changed their layout at least once; {n_stop} stopping-rule cases decided by the reference's sum)")
