"""compute_M with several events: one batched call (ttcr_fsm_raytrace_multi_m) against event-by-event calls.  usage: m_batch_time.py n n_events"""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = int(sys.argv[1]); ne = int(sys.argv[2])
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n)), dtype=np.float32)
v = np.linspace(2.0, 18.0, 7)
X, Y = np.meshgrid(v, v, indexing='ij')
rc = np.stack([X.ravel(), Y.ravel(), np.full(49, 2.0)], axis=1)
srcs = np.delete(4.0 + 0.6 * cases.mt_sources(64), 12, axis=0)[:ne]
src = np.repeat(srcs, len(rc), axis=0); rcv = np.tile(rc, (ne, 1))
g = ttcr_amd.Grid3d(x, x, x, n_threads=ne, cell_slowness=0, method='FSM', tt_from_rp=0, weno=1, dtype=np.float32)
g.set_slowness(s)
for rep in range(2):
    t = time.perf_counter(); tt, M = g.raytrace(src, rcv, compute_M=True); el = time.perf_counter() - t
t = time.perf_counter()
for e in range(ne):
    tt1, M1 = g.raytrace(src[e * 49:(e + 1) * 49], rcv[e * 49:(e + 1) * 49], compute_M=True)
    assert np.array_equal(tt1, tt[e * 49:(e + 1) * 49]) and (M1[0] != M[e]).nnz == 0
el1 = time.perf_counter() - t
print(f"{n}^3, {ne} events x 49 receivers, compute_M: one call {el*1e3:.0f} ms, event by event {el1*1e3:.0f} ms (same traveltimes and matrices)", flush=True)
