"""BASELINE config 5 (Grid2d 4096^2 nodes, 16 sources, fp32) + a single source, for 2-D tuning"""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dx = 20.0/(n-1); x = np.arange(n)*dx
s2 = np.ascontiguousarray(np.broadcast_to((1/(1+0.1*x))[None, :], (n, n)), dtype=np.float32)
rc = np.stack([np.zeros(21), np.linspace(0, 20, 21)], axis=1)
NSL = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [16, 1]
for ns in NSL:
    g = ttcr_amd.Grid2d(x, x, n_threads=ns, cell_slowness=0, method='FSM', weno=0, dtype=np.float32)
    g.set_slowness(s2)
    srcs = cases.mt_sources(max(ns, 16), ndim=2)[:ns]
    best = 1e9
    for _ in range(3):
        g.raytrace(np.repeat(srcs, len(rc), axis=0), np.tile(rc, (ns, 1)))
        best = min(best, g.timing()['sweep_ms'])
    it = sum(g.get_niter(i) for i in range(ns))
    print(f"{n}^2 x{ns}: sweeps {best:.2f} ms, iterations {it}, {n*n*it/best/1e3:.0f} Mnodes/s/iter ({56*n*n*it/best/1e6:.0f} GB/s algorithmic)", flush=True)
    del g
