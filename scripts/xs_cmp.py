"""mode 1 (launch per sweep) vs mode 2 (one launch per iteration, overlapping sweeps): bit equality + speed"""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
counts = [int(c) for c in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 8, 16, 64]
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n))).astype(np.float32)
rcv = np.array([[0., 0, 0]])
for ns in counts:
    src = cases.mt_sources(ns)
    res = {}
    for mode in (1, 2):
        g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
        g.set_option('mode', mode)
        g.set_slowness(s)
        best = 1e9
        for rep in range(3):
            g.raytrace(src, np.repeat(rcv, ns, 0))
            tm = g.timing()
            best = min(best, tm['sweep_ms'])
        chk = [g.get_grid_traveltimes(i) for i in sorted({0, ns // 2, ns - 1})]
        res[mode] = (best, chk, g.get_niter())
        del g
    eq = all(np.array_equal(a, b) for a, b in zip(res[1][1], res[2][1]))
    upd = n ** 3 * 8 * 2 * ns
    print(f"n={n} sources={ns}: mode1 {res[1][0]:.1f} ms ({upd/res[1][0]/1e3:.0f} Mn/s)  mode2 {res[2][0]:.1f} ms ({upd/res[2][0]/1e3:.0f} Mn/s)  "
          f"speedup {res[1][0]/res[2][0]:.2f}  equal={eq} niter={res[1][2]},{res[2][2]}", flush=True)
