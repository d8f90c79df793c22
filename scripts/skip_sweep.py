"""Sweep time with exact skipping off / on over batch sizes (gradient model of the bench, run to convergence).
usage: skip_sweep.py n S1,S2,... [weno]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = int(sys.argv[1]); SS = [int(v) for v in sys.argv[2].split(',')]; weno = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n))).astype(np.float32)
for S in SS:
    src = cases.mt_sources(max(S, 1))[:S]; rcv = np.zeros((S, 3)); out = []
    for skip in (0, 1):
        g = ttcr_amd.Grid3d(x, x, x, n_threads=S, cell_slowness=0, method='FSM', tt_from_rp=0, weno=weno, dtype=np.float32)
        g.set_slowness(s); g.set_option('skip', skip)
        best = None
        for r in range(3):
            g.raytrace(src, rcv); tm = g.timing()
            if best is None or tm['sweep_ms'] < best['sweep_ms']: best = tm
        out.append((best['sweep_ms'], best['evaluated_updates'] / max(best['node_updates'], 1)))
        del g
    print(f"n={n} S={S} weno={weno}: skip off {out[0][0]:9.2f} ms | on {out[1][0]:9.2f} ms (evaluated {out[1][1]:.3f}) | ratio {out[0][0]/out[1][0]:.3f}", flush=True)
