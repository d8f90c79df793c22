#!/usr/bin/env python
"""time of a batch of sources per sweep-iteration (two fixed iterations): python scripts/lone_time.py [n=512] [reps=4] [sources=1]
(environment: TTCR_AMD_LIB, TTCR_FSM_SKIP, TTCR_FSM_PAIR, TTCR_FSM_WGS)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, ttcr_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nsrc = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
g = ttcr_amd.Grid3d(x, x, x, n_threads=nsrc, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x)).astype(np.float32), (n, n, n)))
g.set_slowness(s)
g.set_option("fixed_iters", 2)
src = cases.mt_sources(max(nsrc, 1))[:nsrc]
rcv = np.array([[0.0, 0.0, 0.0]])
best = None
for _ in range(reps):
    g.raytrace(np.repeat(src, 1, axis=0), np.tile(rcv, (nsrc, 1)))
    ms = g.timing()["sweep_ms"] / 2
    best = ms if best is None else min(best, ms)
print(f"n={n} sources={nsrc} lib={os.path.basename(os.environ.get('TTCR_AMD_LIB', 'libttcr_amd.so'))} skip={os.environ.get('TTCR_FSM_SKIP', 'auto')} pair={os.environ.get('TTCR_FSM_PAIR','default')} "
      f"[{g.last_kernel()}]: {best:.3f} ms per sweep-iteration ({104.0 * n ** 3 * nsrc / (best * 1e-3) / 8e12:.3f} of the roofline)", flush=True)
