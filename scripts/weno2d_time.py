"""2-D default path (weno=1) timing: n x n nodes, ns sources"""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd
n = int(sys.argv[1]); ns = int(sys.argv[2])
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, :], (n, n)))
rng = np.random.default_rng(3)
srcs = np.round(rng.uniform(2, 18, (ns, 2)) / dx) * dx
rc = np.column_stack([np.linspace(1, 19, 50), np.full(50, 19.0)])
g = ttcr_amd.Grid2d(x, x, n_threads=ns, cell_slowness=0, method='FSM', tt_from_rp=0, weno=1, dtype=np.float32)
g.set_slowness(s)
best = 1e9
for _ in range(2):
    g.raytrace(np.repeat(srcs, len(rc), axis=0), np.tile(rc, (ns, 1)))
    best = min(best, g.timing()['sweep_ms'])
it = sum(g.get_niter(i) + g.get_niterw(i) for i in range(ns))
print(f"weno2d {n}^2 x{ns}: sweeps {best:.1f} ms, iterations summed {it}, {n*n*it/best/1e3:.0f} Mnodes/s/iter", flush=True)
