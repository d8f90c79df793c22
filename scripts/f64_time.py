"""fp64 (the reference's default precision): 256^3 / 384^3 gradient, 1 and 8 sources"""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
for n in (256, 384):
    dx = 20.0 / (n - 1); x = np.arange(n) * dx
    s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n)))
    rc = cases.rcv_lattice3d()
    for ns in (1, 8):
        g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float64)
        g.set_slowness(s)
        srcs = cases.mt_sources(8)[:ns]
        best = 1e9
        for _ in range(3):
            g.raytrace(np.repeat(srcs, len(rc), axis=0), np.tile(rc, (ns, 1)))
            best = min(best, g.timing()['sweep_ms'])
        it = sum(g.get_niter(i) for i in range(ns))
        print(f"fp64 {n}^3 x{ns}: sweeps {best:.2f} ms, iterations {it}, {n**3*it/best/1e3:.0f} Mnodes/s/iter ({208*n**3*it/best/1e6:.0f} GB/s algorithmic at 208 B/node)", flush=True)
        del g
