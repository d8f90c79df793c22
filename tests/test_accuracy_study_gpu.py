"""Replay, on the GPU, of the fast-sweeping rows of the reference's own accuracy study (tests/accuracy_grid3d.cpp, results in
tests/accuracy_grid3d.csv) and of its float / "fine" FSM test (tests/test_grid3d.cpp:452-488: float grids, fine models,
weno3, < 1 % against the analytic fields):

  study 1 (accuracy_grid3d.cpp:203-252): one source at the origin, 441 receivers, layers (cell slowness -> Grid3Drcfs) and
      gradient (node slowness -> Grid3Drnfs) models, medium (41^3 nodes) and fine (161^3 nodes), double and float, weno3;
      error = mean relative misfit against the analytic field at the nearest point (get_rel_error);
  study 2 (accuracy_grid3d.cpp:258-345): constant velocity 3, the 100 sources of mt19937_64(12345) in ONE multi-source
      call, error = mean relative misfit against s * distance over all source-receiver pairs.

Expected values: tests/golden/accuracy_study.json, computed in the build container by the compiled, unmodified reference
(the two constant / fine rows by the CPU restatement pinned to it; tests/golden/make_accuracy_study.py).  The HIP path
reproduces the reference bit for bit, so the same error expression gives the same number, to rounding of the mean.  The
four medium rows of study 1 also reproduce the PUBLISHED CSV to the six digits it prints; the fine and constant rows of
the CSV do not come out of the reference sources as they lie in /root/reference (the reference itself gives e.g.
0.000589 for gradient / fine / double where the CSV says 0.00280), which the fixture records row by row.
The medium layers / gradient models, the source, the receivers and the analytic fields are the reference's data files
under tests/files; the fine and constant models are regenerated from the formulas of its generator scripts."""
import json
import os

import numpy as np
import pytest

import accuracy_study as S
import cases
from test_io_formats import F, rel_error
from ttcr_amd import io

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "accuracy_study.json")) as _f:
    EXPECTED = json.load(_f)


def _check(err, precision, name, resolution):
    e = EXPECTED[f"{precision},{name},{resolution}"]
    assert abs(err - e["error"]) <= 1e-12 * e["error"], (err, e)
    assert bool(S.six_digits(err, e["published"])) == e["published_reproduced"], (err, e)
    if resolution == "medium" and name != "constant":
        assert e["published_reproduced"]               # these four rows are the CSV's, digit for digit


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("resolution", ["medium", "fine"])
@pytest.mark.parametrize("name", ["layers", "gradient"])
def test_convergence_study_rows(precision, resolution, name):
    import ttcr_amd

    dt = S.DTYPE[precision]
    x, s, cell = S.model(name, resolution)
    if resolution == "medium":   # the reference's own model file holds exactly what the formula gives
        m = io.model_from_vtr(F(name + "_medium.vtr"))
        np.testing.assert_array_equal(m["x"], x)
        np.testing.assert_array_equal(m["slowness"], s.flatten("F"))
    src, t0 = io.read_src(F("src.dat"))
    rcv = io.read_rcv(F("rcv.dat"))
    g = ttcr_amd.Grid3d(x, x, x, cell_slowness=cell, method="FSM", tt_from_rp=0, weno=1, eps=1e-5, maxit=50, dtype=dt)
    tt = g.raytrace(np.hstack([t0[:, None], src]), rcv, slowness=s)
    ref = "sol_analytique_couches_tt.vtr" if name == "layers" else "sol_analytique_gradient_tt.vtr"
    err = rel_error(F(ref), rcv, np.asarray(tt, dtype=np.float64), 3)
    assert err < 0.01                                   # tests/test_grid3d.cpp:466,199
    _check(err, precision, name, resolution)


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("resolution", ["medium", "fine"])
def test_constant_velocity_study_rows(precision, resolution):
    import ttcr_amd

    dt = S.DTYPE[precision]
    x, s, _ = S.model("constant", resolution)
    srcs = cases.mt_sources(100)                        # make_sources(100, 12345), doubles
    rcv = io.read_rcv(F("rcv.dat"))
    g = ttcr_amd.Grid3d(x, x, x, n_threads=16, cell_slowness=0, method="FSM", tt_from_rp=0, weno=1, eps=1e-5, maxit=50,
                        dtype=dt)
    nr = rcv.shape[0]
    tt = g.raytrace(np.repeat(srcs, nr, axis=0), np.tile(rcv, (len(srcs), 1)), slowness=s)
    _check(S.constant_error(dt, srcs, rcv, tt), precision, "constant", resolution)
