"""is a stalled walk (step limit) the same through the batched kernel launch and the per-source recording launch?"""
import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = 256; ns = 16
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n)), dtype=np.float32)
v = np.linspace(2.0, 18.0, 21)
X, Y = np.meshgrid(v, v, indexing='ij')
rc = np.stack([X.ravel(), Y.ravel(), np.full(441, 2.0)], axis=1)
srcs = 4.0 + 0.6 * cases.mt_sources(64)[:ns]
g = ttcr_amd.Grid3d(x, x, x, n_threads=1, cell_slowness=0, method='FSM', tt_from_rp=1, weno=1, dtype=np.float32)
g.set_slowness(s)
for k, p in enumerate(srcs):
    res = []
    for rays in (False, True):
        try:
            g.raytrace(np.repeat(p[None, :], 441, axis=0), rc, return_rays=rays)
            res.append("ok")
        except RuntimeError as e:
            res.append(str(e).split("\n")[0][-70:])
    print(k, res, flush=True)
