"""tests/golden/edge_receiver_case.npz (receiver on a boundary edge, two-point source, WENO, return_rays, fp64):
HIP path against the oracle"""
import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd
from oracle import oracle
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'edge_receiver_case.npz'))
s, src, t0, rcv = d['s'], d['src'], d['t0'], d['rcv']; dx = float(d['dx']); org = tuple(d['org']); nc = tuple(int(v) for v in d['nc'])
axes = [o + np.arange(n + 1) * dx for o, n in zip(org, nc)]
o = oracle.solve3d(np.float64, nc, dx, org, s.flatten("F"), src, t0, return_rays=True, cell_slowness=False, rcv=rcv, weno=True)
g = ttcr_amd.Grid3d(*axes, cell_slowness=False, method="FSM", tt_from_rp=0, weno=1, dtype=np.float64)
tt, rays = g.raytrace(np.hstack([t0[:, None], src]), rcv, slowness=s, aggregate_src=True, return_rays=True)
print("oracle", o["tt_rcv"], "hip", tt, "equal", np.array_equal(tt, o["tt_rcv"]),
      "rays equal", all(np.array_equal(a, b) for a, b in zip(rays, o["rays"])))
