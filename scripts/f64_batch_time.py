"""fp64 first-order / WENO solves of small batches: python scripts/f64_batch_time.py n weno sources..."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = int(sys.argv[1]); weno = int(sys.argv[2])
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n)))
rc = cases.rcv_lattice3d()
for ns in (int(v) for v in sys.argv[3:]):
    g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=0, method='FSM', tt_from_rp=0, weno=weno, dtype=np.float64)
    g.set_slowness(s)
    srcs = cases.mt_sources(8)[:ns]
    best = 1e9
    for _ in range(3):
        g.raytrace(np.repeat(srcs, len(rc), axis=0), np.tile(rc, (ns, 1)))
        best = min(best, g.timing()['sweep_ms'])
    print(f"fp64 {n}^3 weno {weno} x{ns} [{g.last_kernel()}]: sweeps {best:.2f} ms", flush=True)
    del g
