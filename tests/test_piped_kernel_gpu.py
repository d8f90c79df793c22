"""Pipelined sweep kernel (ttcr_amd/csrc/fsm_piped_kernels.h, option "piped" = 1: four march wavefronts + two staging wavefronts per
patch, two LDS tiles): bit-identical to the default kernel -- fields, iteration counts, change history.
Reference semantics: Grid3Drn::sweep / update_node, ttcr/Grid3Drn.h:2816-2959 (the default kernel is pinned to the oracle by
test_parity_gpu.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [9, 21])   # (seed 9: 41-node columns, the shapes that found the 16-byte store hazard of gfx950)
def test_piped_kernel_matches_default_kernel(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "piped_check.py"), "--cases", "12", "--no-time", "--seed", str(seed)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "cases with differences: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
