#!/usr/bin/env python
"""lone 512^3 source: time of the first and of the second sweep-iteration with / without exact skipping
python scripts/lone_skip.py [n=512] [nsrc=1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, ttcr_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x)).astype(np.float32), (n, n, n)))
src = cases.mt_sources(max(nsrc, 1))[:nsrc]
rcv = np.array([[0.0, 0.0, 0.0]])
ITERS = tuple(int(v) for v in os.environ.get('ITERS', '1,2,3').split(','))
for skip in (0, 1):
    g = ttcr_amd.Grid3d(x, x, x, n_threads=nsrc, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
    g.set_slowness(s)
    g.set_option("skip", skip)
    res = {}
    for it in ITERS:
        g.set_option("fixed_iters", it)
        best = None
        for _ in range(3):
            g.raytrace(src, np.tile(rcv, (nsrc, 1)))
            t = g.timing()
            if best is None or t["sweep_ms"] < best[0]:
                best = (t["sweep_ms"], t["evaluated_updates"] / (8.0 * n ** 3 * nsrc * it), t["solve_ms"] if "solve_ms" in t else 0)
        res[it] = best
    print(f"n={n} sources={nsrc} skip={skip} [{g.last_kernel()}]: " + "  ".join(f"{it} it: {v[0]:.2f} ms (evaluated {v[1]:.3f})" for it, v in res.items()), flush=True)
