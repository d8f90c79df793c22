"""WENO stage: evaluated fraction, launches and time (weno=1; sources from the bench set)
usage: weno_eval.py n nsrc"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = int(sys.argv[1]); ns = int(sys.argv[2])
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n)))
rc = cases.rcv_lattice3d(); srcs = cases.mt_sources(64)[:ns]
g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=0, method='FSM', tt_from_rp=0, weno=1, dtype=np.float32)
g.set_slowness(s)
for k, v in [a.split('=') for a in sys.argv[3:]]:
    g.set_option(k, float(v))
for _ in range(2):
    g.raytrace(np.repeat(srcs, len(rc), axis=0), np.tile(rc, (ns, 1)))
    tm = g.timing()
it1 = [g.get_niter(i) for i in range(ns)]; itw = [g.get_niterw(i) for i in range(ns)]
tot = sum(it1) + sum(itw)
print(f"weno {n}^3 x{ns}: sweeps {tm['sweep_ms']:.1f} ms, launches {tm['kernel_launches']}, niter {it1} niterw {itw}, "
      f"evaluated {tm['evaluated_updates'] / (n ** 3 * 8 * tot):.3f} of all updates, kernel {g.last_kernel()}", flush=True)
ch = g.get_changes(0)
print("changes slot 0:", [f"{c:.3g}" for c in ch[0]], "|", [f"{c:.3g}" for c in (ch[1] if len(ch) > 1 else [])], flush=True)
