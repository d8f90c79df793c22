"""PCIe-inclusive variant of the bench step: the slowness model handed over as a HOST array (numpy) before every
batch of solves, and one full traveltime field read back."""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n)), dtype=np.float32)
rc = cases.rcv_lattice3d(); srcs = cases.mt_sources(64)[:ns]
src = np.repeat(srcs, len(rc), axis=0); rcv = np.tile(rc, (ns, 1))
g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
g.set_slowness(s); g.raytrace(src, rcv)
for rep in range(3):
    t0 = time.perf_counter(); g.set_slowness(s); t1 = time.perf_counter(); g.raytrace(src, rcv); t2 = time.perf_counter()
    T = g.get_grid_traveltimes(0); t3 = time.perf_counter()
    it = sum(g.get_niter(i) for i in range(ns))
    print(f"{n}^3 x{ns}: set_slowness(host array) {1e3*(t1-t0):.1f} ms, raytrace {1e3*(t2-t1):.1f} ms, one field to host {1e3*(t3-t2):.1f} ms; "
          f"upload + solves: {n**3*it/(t2-t0)/1e6:.0f} Mnodes/s per sweep-iteration", flush=True)
