"""BASELINE.json configs 1, 2, 4, 5 on one MI355X (config 3 is bench.py). Prints one line each."""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases

def run(g, src, rcv, reps=3):
    best = None
    for _ in range(reps):
        t = time.perf_counter(); tt = g.raytrace(src, rcv); el = time.perf_counter() - t
        best = el if best is None else min(best, el)
    return tt, best, g.timing()

# C1: 64^3 cells constant slowness, source at the centre node, analytic check t = s*r (fp64, reference default dtype)
n = 65; x = np.arange(n, dtype=float)
g = ttcr_amd.Grid3d(x, x, x, cell_slowness=1, method='FSM', tt_from_rp=0, weno=0)
g.set_slowness(np.full((64, 64, 64), 1 / 3.))
tt, el, tm = run(g, np.array([[32., 32, 32]]), np.array([[0., 0, 0]]))
T = g.get_grid_traveltimes(); i, j, k = np.meshgrid(x, x, x, indexing='ij'); r = np.sqrt((i-32)**2+(j-32)**2+(k-32)**2)
m = r > 0; err = np.mean(np.abs(T[m] - r[m]/3) / (r[m]/3))
print(f"C1 64^3 cells constant fp64: {el*1e3:.2f} ms, niter {g.get_niter()}, mean rel err vs analytic {err:.4f}")

# C2: 256^3 nodes gradient, 1 source, fp32
n = 256; dx = 20.0/(n-1); x = np.arange(n)*dx
s = np.ascontiguousarray(np.broadcast_to((1/(1+0.1*x))[None, None, :], (n, n, n)), dtype=np.float32)
g = ttcr_amd.Grid3d(x, x, x, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
g.set_slowness(s)
for name, p in (("corner", [[0., 0, 0]]), ("centre", [[10., 10, 10]])):
    tt, el, tm = run(g, np.array(p), cases.rcv_lattice3d())
    print(f"C2 256^3 gradient fp32 1 source ({name}): {el*1e3:.2f} ms wall, sweeps {tm['sweep_ms']:.2f} ms, niter {g.get_niter()}, "
          f"{n**3*g.get_niter()/tm['sweep_ms']/1e3:.0f} Mnodes/s per sweep-iteration, {104*n**3*g.get_niter()/tm['sweep_ms']/1e6:.0f} GB/s algorithmic")
del g
# C4: 256^3 CELLS layers model, cell_slowness=True, 8 sources
nc = 256; dx = 20.0/nc; x = np.arange(nc+1)*dx
sc = np.ascontiguousarray(np.broadcast_to((1/(1+0.1*(np.floor(np.arange(nc)*dx)+0.5)))[None, None, :], (nc, nc, nc)), dtype=np.float32)
g = ttcr_amd.Grid3d(x, x, x, n_threads=8, cell_slowness=1, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
g.set_slowness(sc)
srcs = cases.mt_sources(64)[:8]; rc = cases.rcv_lattice3d()
tt, el, tm = run(g, np.repeat(srcs, len(rc), axis=0), np.tile(rc, (8, 1)))
it = sum(g.get_niter(i) for i in range(8))
print(f"C4 256^3 cells (257^3 nodes) layers fp32 8 sources: {el*1e3:.1f} ms wall, sweeps {tm['sweep_ms']:.1f} ms, iterations {it}, "
      f"{257**3*it/tm['sweep_ms']/1e3:.0f} Mnodes/s per sweep-iteration, {8/el:.1f} sources/s")
del g
# C5: Grid2d 4096 x 4096 nodes gradient, 16 sources
n = 4096; dx = 20.0/(n-1); x = np.arange(n)*dx
s2 = np.ascontiguousarray(np.broadcast_to((1/(1+0.1*x))[None, :], (n, n)), dtype=np.float32)
g = ttcr_amd.Grid2d(x, x, n_threads=16, cell_slowness=0, method='FSM', weno=0, dtype=np.float32)
g.set_slowness(s2)
srcs = cases.mt_sources(16, ndim=2); rc = np.stack([np.zeros(21), np.linspace(0, 20, 21)], axis=1)
tt, el, tm = run(g, np.repeat(srcs, len(rc), axis=0), np.tile(rc, (16, 1)), reps=2)
it = sum(g.get_niter(i) for i in range(16))
print(f"C5 Grid2d 4096^2 nodes gradient fp32 16 sources: {el*1e3:.1f} ms wall, sweeps {tm['sweep_ms']:.1f} ms, iterations {it}, "
      f"{n*n*it/tm['sweep_ms']/1e3:.0f} Mnodes/s per sweep-iteration ({56*n*n*it/tm['sweep_ms']/1e6:.0f} GB/s algorithmic), {16/el:.1f} sources/s")
del g
# beyond the BASELINE configs: the 2-D solver with 1 / 64 sources, the default (WENO) 3-D path with 1 / 8 sources
for ns in (1, 64):
    g = ttcr_amd.Grid2d(x, x, n_threads=ns, cell_slowness=0, method='FSM', weno=0, dtype=np.float32)
    g.set_slowness(s2)
    srcs = cases.mt_sources(64, ndim=2)[:ns]
    tt, el, tm = run(g, np.repeat(srcs, len(rc), axis=0), np.tile(rc, (ns, 1)), reps=2)
    it = sum(g.get_niter(i) for i in range(ns))
    print(f"2-D 4096^2 nodes gradient fp32 {ns} source(s): sweeps {tm['sweep_ms']:.1f} ms, iterations {it}, "
          f"{n*n*it/tm['sweep_ms']/1e3:.0f} Mnodes/s per sweep-iteration ({56*n*n*it/tm['sweep_ms']/1e6:.0f} GB/s algorithmic = "
          f"{56*n*n*it/tm['sweep_ms']/1e6/8000:.3f} of 8 TB/s)")
    del g
n = 256; dx = 20.0/(n-1); x = np.arange(n)*dx
s = np.ascontiguousarray(np.broadcast_to((1/(1+0.1*x))[None, None, :], (n, n, n)), dtype=np.float32)
rc = cases.rcv_lattice3d()
for ns in (1, 8):
    g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=0, method='FSM', tt_from_rp=0, weno=1, dtype=np.float32)
    g.set_slowness(s)
    srcs = cases.mt_sources(64)[:ns]
    tt, el, tm = run(g, np.repeat(srcs, len(rc), axis=0), np.tile(rc, (ns, 1)), reps=2)
    it = sum(g.get_niter(i) + g.get_niterw(i) for i in range(ns))
    print(f"WENO (ttcrpy default) 256^3 gradient fp32 {ns} source(s): sweeps {tm['sweep_ms']:.1f} ms, iterations (first-order + WENO, summed) {it}, "
          f"{n**3*it/tm['sweep_ms']/1e3:.0f} Mnodes/s per sweep-iteration, {ns/el:.2f} sources/s")
    del g
