#!/usr/bin/env python
"""one source run to convergence, exact skipping off / on, both arithmetic modes, smooth and rough models: ms per solve (sweeps)
python scripts/lone_skip_models.py [n=512] [nsrc=1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, ttcr_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
rng = np.random.default_rng(5)
c = rng.uniform(0.4, 1.0, (n // 16 + 2,) * 3)
models = {"gradient": np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x)).astype(np.float32), (n, n, n))),
          "rough (16^3 blocks)": np.repeat(np.repeat(np.repeat(c, 16, 0), 16, 1), 16, 2)[:n, :n, :n].astype(np.float32).copy()}
src = cases.mt_sources(max(nsrc, 1))[:nsrc]
rcv = np.zeros((nsrc, 3))
g = ttcr_amd.Grid3d(x, x, x, n_threads=nsrc, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
for name, m in models.items():
    g.set_slowness(m)
    for ar in (0, 1):
        g.set_option("arith", ar)
        for sk in (0, 1):
            g.set_option("skip", sk)
            best = None
            for _ in range(3):
                g.raytrace(src, rcv)
                t = g.timing()
                if best is None or t["sweep_ms"] < best["sweep_ms"]: best = t
            print(f"n={n} sources={nsrc} {name}: arith={ar} skip={sk}: sweeps {best['sweep_ms']:.2f} ms, niter {sorted({g.get_niter(i) for i in range(nsrc)})}, "
                  f"evaluated {best['evaluated_updates'] / max(best['node_updates'], 1):.3f}, total {best['total_ms']:.2f} ms  [{g.last_kernel()}]", flush=True)
