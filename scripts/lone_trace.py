#!/usr/bin/env python
"""unit trace of a lone source (trace build: TTCR_AMD_LIB=variants/trace.so TTCR_FSM_PROF=1 TTCR_FSM_PROF_TRACE=file):
python scripts/lone_trace.py n skip iters [nsrc]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, ttcr_amd
n, skip, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nsrc = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x)).astype(np.float32), (n, n, n)))
src = cases.mt_sources(max(nsrc, 1))[:nsrc]
print("source", src[0][:3] / dx)
rcv = np.array([[0.0, 0.0, 0.0]])
g = ttcr_amd.Grid3d(x, x, x, n_threads=nsrc, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
g.set_slowness(s)
g.set_option("skip", skip)
g.set_option("fixed_iters", iters)
for _ in range(2):
    g.raytrace(src, np.tile(rcv, (nsrc, 1)))
t = g.timing()
print(f"n={n} skip={skip} iters={iters} [{g.last_kernel()}] sweep_ms {t['sweep_ms']:.2f} evaluated {t['evaluated_updates'] / (8.0 * n ** 3 * nsrc * iters):.3f}")
