#!/usr/bin/env python
"""tolerance-grade arithmetic (option arith = 1) against the default (bit-identical to the reference): fields to convergence on
the bench model, then the time per sweep-iteration of both.  python scripts/arith_check.py [n_cmp=256] [n_time=512] [sources=1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, ttcr_amd
ncmp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ntime = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nsrc = int(sys.argv[3]) if len(sys.argv) > 3 else 1


def make(n, rough=False):
    dx = 20.0 / (n - 1)
    x = np.arange(n) * dx
    g = ttcr_amd.Grid3d(x, x, x, n_threads=nsrc, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
    if rough:
        rng = np.random.default_rng(5)
        s = rng.uniform(0.2, 1.0, (n, n, n)).astype(np.float32)
    else:
        s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x)).astype(np.float32), (n, n, n)))
    g.set_slowness(s)
    return g


src = cases.mt_sources(max(nsrc, 1))[:nsrc]
rcv = np.tile(np.array([[0.0, 0.0, 0.0]]), (nsrc, 1))
if ncmp:
    for rough in (False, True):
        g = make(ncmp, rough)
        fields, its = [], []
        for ar in (0, 1):
            g.set_option("arith", ar)
            g.raytrace(src, rcv)
            fields.append(np.array(g.get_grid_traveltimes(0)))
            its.append(g.get_niter(0))
        d = fields[1].astype(np.float64) - fields[0].astype(np.float64)
        print(f"n={ncmp} rough={rough} niter exact/tolerance {its}: rms {np.sqrt(np.mean(d * d)):.3e} s, max |d| {np.max(np.abs(d)):.3e} s, "
              f"identical {np.mean(fields[0] == fields[1]) * 100:.2f} %, max T {fields[0].max():.3f} [{g.last_kernel()}]", flush=True)
        del g
if ntime:
    g = make(ntime)
    g.set_option("fixed_iters", 2)
    for ar in (0, 1, 0, 1):
        g.set_option("arith", ar)
        best = None
        for _ in range(4):
            g.raytrace(src, rcv)
            ms = g.timing()["sweep_ms"] / 2
            best = ms if best is None else min(best, ms)
        print(f"n={ntime} sources={nsrc} arith={ar} [{g.last_kernel()}]: {best:.3f} ms per sweep-iteration "
              f"({104.0 * ntime ** 3 * nsrc / (best * 1e-3) / 8e12:.3f} of the roofline)", flush=True)
