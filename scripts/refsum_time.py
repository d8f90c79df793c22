#!/usr/bin/env python
"""time of the reference's change sum of a 512^3 fp32 field pair: parallel form against the one-chain kernel (scripts/refsum_time.py [n])"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ttcr_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = np.arange(n) * 0.1
g = ttcr_amd.Grid3d(x, x, x, n_threads=1, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
rng = np.random.default_rng(1)
N = n ** 3
base = rng.uniform(1.0, 2.0, N).astype(np.float32)
d = (rng.uniform(0, 2e-5, N) * (rng.uniform(0, 1, N) < 0.3)).astype(np.float32)
old = (base + d).astype(np.float32)
for par in (True, False, True):
    t = time.perf_counter(); v = g.reference_change(old, base, parallel=par); dt = time.perf_counter() - t
    print(f"n={n} parallel={par}: change {v!r} in {dt*1e3:.1f} ms (host -> device copies of 2 x {N*4/1e6:.0f} MB included)", g.stopping_stats(), flush=True)
