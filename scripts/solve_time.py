"""Sweep time of a fixed-iteration first-order solve on the gradient model: n^3 nodes, S sources (one call).
usage: solve_time.py n S [iters] [reps]   (TTCR_AMD_LIB selects a tuning build, TTCR_FSM_PROF=1 its phase timers)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases

n = int(sys.argv[1]); S = int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n))).astype(np.float32)
g = ttcr_amd.Grid3d(x, x, x, n_threads=S, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
g.set_slowness(s)
g.set_option('fixed_iters', iters)
src = cases.mt_sources(max(S, 1))[:S]
rcv = np.zeros((S, 3))
best = 1e9
for r in range(reps):
    g.raytrace(src, rcv)
    tm = g.timing()
    best = min(best, tm['sweep_ms'])
    print(f"rep {r}: sweep_ms {tm['sweep_ms']:.3f} total_ms {tm['total_ms']:.3f} evaluated {tm['evaluated_updates'] / max(tm['node_updates'], 1):.3f}", flush=True)
per_it = best / iters
print(f"n={n} S={S} iters={iters}: best {best:.3f} ms = {per_it:.3f} ms/sweep-iteration = {n**3*S/per_it/1e3:.0f} Mnodes/s/iter, "
      f"roofline frac {104.0*n**3*S/(per_it*1e-3)/8e12:.4f}", flush=True)
