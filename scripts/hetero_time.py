#!/usr/bin/env python
"""the heterogeneous leg of bench.py alone: 8 sources on 512^3, random 16^3-block model, to convergence; stopping_rule 0 / 1
(TTCR_FSM_HOST_PROF=1: host phases on stderr)    usage: hetero_time.py [n=512] [sources=8]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, ttcr_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
rng = np.random.default_rng(5)
nb = (n + 15) // 16
b = rng.uniform(0.25, 1.0, (nb, nb, nb)).astype(np.float32)
s = np.repeat(np.repeat(np.repeat(b, 16, 0), 16, 1), 16, 2)[:n, :n, :n]
src = cases.mt_sources(64)[:ns]
rcv = cases.rcv_lattice3d()
sr, rr = np.repeat(src, rcv.shape[0], axis=0), np.tile(rcv, (ns, 1))
for rule in (0, 1):
    g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
    g.set_slowness(s)
    g.set_option("stopping_rule", rule)
    g.raytrace(sr, rr)
    t = time.perf_counter(); g.raytrace(sr, rr); wall = (time.perf_counter() - t) * 1e3
    print(f"stopping_rule={rule}: {wall:.1f} ms per step, sweeps {g.timing()['sweep_ms']:.1f} ms, niter {[g.get_niter(i) for i in range(ns)]}, {g.stopping_stats()}", flush=True)
    del g
