"""return_rays with several sources per call: wall time of raytrace(..., return_rays=True) against its sweep time and against tt_from_rp
usage: rays_batch_time.py n nsrc [weno]"""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = int(sys.argv[1]); ns = int(sys.argv[2]); weno = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n)), dtype=np.float32)
v = np.linspace(2.0, 18.0, 21)
X, Y = np.meshgrid(v, v, indexing='ij')
rc = np.stack([X.ravel(), Y.ravel(), np.full(441, 2.0)], axis=1)
srcs = np.delete(4.0 + 0.6 * cases.mt_sources(64), 12, axis=0)[:ns]
src = np.repeat(srcs, len(rc), axis=0); rcv = np.tile(rc, (ns, 1))
for rays in (False, True):
    g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=0, method='FSM', tt_from_rp=1, weno=weno, dtype=np.float32)
    g.set_slowness(s)
    for rep in range(3):
        t = time.perf_counter(); out = g.raytrace(src, rcv, return_rays=rays); el = time.perf_counter() - t
    npts = sum(len(r) for r in out[1]) if rays else 0
    print(f"{n}^3 x{ns} x{len(rc)} receivers, return_rays={rays}: wall {el*1e3:.1f} ms, sweeps {g.timing()['sweep_ms']:.1f} ms, ray points {npts}", flush=True)
