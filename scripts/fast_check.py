#!/usr/bin/env python
"""a tuning build of the tolerance-grade four-wave kernels (TTCR_AMD_LIB_B) against the library (TTCR_AMD_LIB_A or the default): the same
arithmetic and the same partial order must give bit-identical fields.  python scripts/fast_check.py [n_cases=10]"""
import os, sys, subprocess, pickle
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ttcr_amd
    ncases = int(sys.argv[2])
    rng = np.random.default_rng(7)
    out = []
    for case in range(ncases):
        nx, ny, nz = (int(rng.integers(5, 150)) for _ in range(3)) if case else (128, 128, 128)
        nsrc = int(rng.integers(1, 4))
        dx = float(rng.uniform(0.05, 2.0))
        x, y, z = np.arange(nx) * dx, np.arange(ny) * dx, np.arange(nz) * dx
        s = rng.uniform(0.2, 1.0, (nx, ny, nz)).astype(np.float32) if case % 2 == 0 else np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * z)).astype(np.float32), (nx, ny, nz)))
        src = np.column_stack([rng.uniform(0, x[-1], nsrc), rng.uniform(0, y[-1], nsrc), rng.uniform(0, z[-1], nsrc)])
        g = ttcr_amd.Grid3d(x, y, z, n_threads=nsrc, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
        g.set_slowness(s)
        g.set_option("arith", 1)
        g.set_option("slab", 0)
        g.raytrace(src, np.zeros((nsrc, 3)))
        out.append(([np.array(g.get_grid_traveltimes(t)) for t in range(nsrc)], [g.get_niter(t) for t in range(nsrc)], (nx, ny, nz), g.last_kernel()))
    pickle.dump(out, open(sys.argv[3], "wb"))
    sys.exit(0)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
res = []
for tag, lib in (("A", os.environ.get("TTCR_AMD_LIB_A", "")), ("B", os.environ["TTCR_AMD_LIB_B"])):
    env = dict(os.environ)
    if lib:
        env["TTCR_AMD_LIB"] = lib
    else:
        env.pop("TTCR_AMD_LIB", None)
    f = f"/tmp/fast_check_{tag}.pkl"
    subprocess.check_call([sys.executable, __file__, "--child", str(ncases), f], env=env)
    res.append(pickle.load(open(f, "rb")))
bad = 0
for (fa, ia, shp, ka), (fb, ib, _, kb) in zip(*res):
    ok = ia == ib and all(np.array_equal(a, b) for a, b in zip(fa, fb))
    print(shp, len(fa), "sources", ka, "niter", ia, ib, "identical" if ok else "DIFFERENT")
    bad += 0 if ok else 1
print("fast_check:", "all identical" if bad == 0 else f"{bad} MISMATCHES")
