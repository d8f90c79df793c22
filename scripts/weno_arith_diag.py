#!/usr/bin/env python
"""where the tolerance-grade WENO stage differs from the exact one (the exact GPU mode is the oracle bit for bit): python scripts/weno_arith_diag.py [n=128] [nsrc=4]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, ttcr_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x)).astype(np.float32), (n, n, n)))
srcs = cases.mt_sources(64)[:nsrc]
rcv = np.zeros((nsrc, 3))
g = ttcr_amd.Grid3d(x, x, x, n_threads=nsrc, cell_slowness=0, method="FSM", tt_from_rp=0, weno=1, dtype=np.float32)
g.set_slowness(s)
g.raytrace(srcs, rcv)
ref = [np.array(g.get_grid_traveltimes(i), dtype=np.float64) for i in range(nsrc)]
itr = [(g.get_niter(i), g.get_niterw(i)) for i in range(nsrc)]
g.set_option("arith", 1)
g.raytrace(srcs, rcv)
X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
for i in range(nsrc):
    f = np.array(g.get_grid_traveltimes(i), dtype=np.float64)
    d = f - ref[i]
    r = np.sqrt((X - srcs[i][0]) ** 2 + (Y - srcs[i][1]) ** 2 + (Z - srcs[i][2]) ** 2)
    k = np.unravel_index(np.argmax(np.abs(d)), d.shape)
    big = np.abs(d) > 1e-5
    print(f"source {i} at {np.round(srcs[i] / dx, 1)}: niter {itr[i]} / {(g.get_niter(i), g.get_niterw(i))} rms {np.sqrt(np.mean(d * d)):.3e} max {np.abs(d).max():.3e} at node {k} "
          f"(distance {r[k] / dx:.1f} nodes, T {ref[i][k]:.4f}); nodes with |d| > 1e-5: {big.sum()} ({100.0 * big.mean():.3f} %), their mean distance {r[big].mean() / dx if big.any() else 0:.1f} nodes; "
          f"rms beyond 8 nodes of the source {np.sqrt(np.mean(d[r > 8 * dx] ** 2)):.3e}", flush=True)
    if big.any():
        idx = np.argwhere(big)
        print("   bounding box of those nodes:", idx.min(0), idx.max(0), " on a grid face:", int(np.sum((idx == 0).any(1) | (idx == n - 1).any(1))))
