"""bench.py's `heterogeneous` leg alone (8 sources at 512^3, random blocks; reference stopping rule against the fp64 sum alone).
TTCR_FSM_HOST_PROF=1 adds the host-side phases of every iteration on stderr.  Usage: python scripts/hetero_time.py [n]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
import ttcr_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 20.0 / (n - 1)
x = np.arange(n, dtype=np.float64) * dx
g = ttcr_amd.Grid3d(x, x, x, n_threads=8, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32, device=0)
out = bench.heterogeneous_leg(g, n)
print(json.dumps(out))
