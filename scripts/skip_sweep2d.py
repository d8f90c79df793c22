"""2-D: sweep time with exact skipping off / on.  usage: skip_sweep2d.py n S1,S2,..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = int(sys.argv[1]); SS = [int(v) for v in sys.argv[2].split(',')]
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s2 = np.ascontiguousarray(np.broadcast_to((1 / (1 + 0.1 * x))[None, :], (n, n)), dtype=np.float32)
for S in SS:
    srcs = cases.mt_sources(max(S, 1), ndim=2)[:S]; rcv = np.zeros((S, 2)); out = []
    for skip in (0, 1):
        g = ttcr_amd.Grid2d(x, x, n_threads=S, cell_slowness=0, method='FSM', weno=0, dtype=np.float32)
        g.set_slowness(s2); g.set_option('skip', skip)
        best = None
        for r in range(3):
            g.raytrace(srcs, rcv); tm = g.timing()
            if best is None or tm['sweep_ms'] < best['sweep_ms']: best = tm
        out.append((best['sweep_ms'], best['evaluated_updates'] / max(best['node_updates'], 1)))
        del g
    print(f"2-D n={n} S={S}: skip off {out[0][0]:9.2f} ms | on {out[1][0]:9.2f} ms (evaluated {out[1][1]:.3f}) | ratio {out[0][0]/out[1][0]:.3f}", flush=True)
