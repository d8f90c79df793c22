"""bench.py's `heterogeneous` leg alone (8 sources at 512^3, random blocks; reference stopping rule against the fp64 sum alone).
TTCR_FSM_HOST_PROF=1 adds the host-side phases of every iteration on stderr.  Usage: python scripts/hetero_time.py [n] [history]
history = 1: the grid first runs bench.py's `eight_sources` steps on the gradient model, as in the bench."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
import cases  # noqa: E402
import ttcr_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 20.0 / (n - 1)
x = np.arange(n, dtype=np.float64) * dx
if len(sys.argv) > 2 and int(sys.argv[2]):
    s_dev = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(bench.gradient_slowness_f32(n, dx), (n, n, n))).reshape(-1)).cuda()
    out8, g = bench.small_batch_leg(n, dx, x, s_dev, 0, 8, 3, None)
    print("eight_sources", out8["ms_per_step_wall"], file=sys.stderr)
else:
    g = ttcr_amd.Grid3d(x, x, x, n_threads=8, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32, device=0)
out = bench.heterogeneous_leg(g, n)
print(json.dumps(out))
src = cases.mt_sources(64)[:8]
rcv = cases.rcv_lattice3d()
sr, rr = np.repeat(src, rcv.shape[0], axis=0), np.tile(rcv, (8, 1))
for rule in (0, 0, 0, 1, 1, 0, 0):
    g.set_option("stopping_rule", rule)
    t = time.perf_counter()
    g.raytrace(sr, rr)
    print("rule %d: %.2f ms  kernel %s" % (rule, (time.perf_counter() - t) * 1e3, g.last_kernel()), file=sys.stderr)
