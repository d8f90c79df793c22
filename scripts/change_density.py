"""How many terms of the stopping rule's sum are non-zero in the late iterations of the heterogeneous leg (8 sources at 512^3, random
blocks), and how they cluster in node order: fraction of nodes, of 1024-node runs and of 4096-node tiles that changed at all."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import cases  # noqa: E402
import ttcr_amd  # noqa: E402

n = 512
dx = 20.0 / (n - 1)
x = np.arange(n, dtype=np.float64) * dx
g = ttcr_amd.Grid3d(x, x, x, n_threads=8, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32, device=0)
rng = np.random.default_rng(5)
nb = (n + 15) // 16
b = torch.from_numpy(rng.uniform(0.25, 1.0, (nb, nb, nb)).astype(np.float32)).cuda()
s = b.repeat_interleave(16, 0).repeat_interleave(16, 1).repeat_interleave(16, 2)[:n, :n, :n]
s = s.permute(2, 1, 0).contiguous().reshape(-1)
torch.cuda.synchronize()
g.set_slowness_device(s.data_ptr(), s.numel())
src = cases.mt_sources(64)[:8]
rcv = cases.rcv_lattice3d()
sr, rr = np.repeat(src, rcv.shape[0], axis=0), np.tile(rcv, (8, 1))
prev = None
for k in range(5, 12):
    g.set_option("fixed_iters", k)
    g.raytrace(sr, rr)
    cur = [np.asarray(g.get_grid_traveltimes(i)).reshape(-1).copy() for i in (0, 5)]
    if prev is not None:
        for q, (a, c) in enumerate(zip(prev, cur)):
            d = np.abs(a - c)
            nz = d != 0
            m = nz.size // 4096 * 4096
            print("iteration %2d source %d: sum %.4e  non-zero %.4f of the nodes, %.4f of the 1024-runs, %.4f of the 4096-tiles, %.4f of the 16-runs" % (
                k, (0, 5)[q], d.astype(np.float64).sum(), nz.mean(), nz[:m].reshape(-1, 1024).any(axis=1).mean(),
                nz[:m].reshape(-1, 4096).any(axis=1).mean(), nz[:m].reshape(-1, 16).any(axis=1).mean()), flush=True)
    prev = cur
