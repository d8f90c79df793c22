"""Slab sweep kernel (ttcr_amd/csrc/fsm_slab_kernels.h, option "slab" = 1): bit-identical to the default kernel.
Reference semantics: Grid3Drn::sweep / update_node, ttcr/Grid3Drn.h:2816-2959 (the default kernel is pinned to the oracle by
test_parity_gpu.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", ["2x4", "1x4", "4x2"])
def test_slab_kernel_matches_default_kernel(shape):
    env = dict(os.environ, TTCR_FSM_SLAB_SHAPE=shape)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "slab_check.py"), "--cases", "14", "--no-time", "--seed", "5"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "cases with differences: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
