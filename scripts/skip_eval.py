"""Exact skipping of no-op chunks (option skip): time to solution and fraction of node updates evaluated, skip off / on,
on solves run to convergence (gradient model of the bench, and a heterogeneous model).
usage: skip_eval.py n S weno(0|1) [model: grad|het]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases

n = int(sys.argv[1]); S = int(sys.argv[2]); weno = int(sys.argv[3])
model = sys.argv[4] if len(sys.argv) > 4 else 'grad'
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
if model == 'grad':
    s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n))).astype(np.float32)
else:
    rng = np.random.default_rng(5)
    c = rng.uniform(0.4, 1.0, (n // 16 + 2,) * 3)
    s = np.repeat(np.repeat(np.repeat(c, 16, 0), 16, 1), 16, 2)[:n, :n, :n].astype(np.float32).copy()
src = cases.mt_sources(max(S, 1))[:S]
rcv = np.zeros((S, 3))
fields = {}
for skip in (0, 1):
    g = ttcr_amd.Grid3d(x, x, x, n_threads=S, cell_slowness=0, method='FSM', tt_from_rp=0, weno=weno, dtype=np.float32)
    g.set_slowness(s)
    g.set_option('skip', skip)
    best = None
    for r in range(2):
        g.raytrace(src, rcv)
        tm = g.timing()
        if best is None or tm['sweep_ms'] < best['sweep_ms']:
            best = tm
    it = [(g.get_niter(i), g.get_niterw(i)) for i in range(S)]
    fields[skip] = [g.get_grid_traveltimes(i) for i in range(min(S, 2))]
    print(f"n={n} S={S} weno={weno} {model} skip={skip}: sweeps {best['sweep_ms']:.2f} ms, iterations {it[:4]}, "
          f"evaluated {best['evaluated_updates'] / max(best['node_updates'], 1):.3f} of the node updates", flush=True)
for a, b in zip(fields[0], fields[1]):
    assert np.array_equal(a, b), "skip changed the result"
print("fields identical", flush=True)
