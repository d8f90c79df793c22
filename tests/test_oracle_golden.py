"""The CPU restatement (oracle/fsm_oracle.c) against the committed golden vectors, which
were produced by the unmodified compiled reference (tests/golden/make_golden.py).
Bit-exact: full traveltime field, iteration count and receiver interpolation."""
import numpy as np
import pytest

import cases

ALL = [(c, dt) for c in cases.cases3d() + cases.cases2d() for dt in (np.float32, np.float64)]


def solve(O, c, dt, slowness, weno=False):
    if c["dim"] == 3:
        return O.solve3d(dt, c["ncells"], c["dx"], c["origin"], slowness, c["src"], c["t0"],
                         cell_slowness=c["cell_slowness"], translate=c["translate"], rcv=c["rcv"], weno=weno)
    return O.solve2d(dt, c["ncells"], c["dx"], c["dz"], c["origin"], slowness, c["src"], c["t0"],
                     cell_slowness=c["cell_slowness"], rcv=c["rcv"], weno=weno)


@pytest.mark.parametrize("c,dt", ALL, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in ALL])
def test_oracle_matches_golden(oracle, golden, c, dt):
    key = f"{c['name']}/{np.dtype(dt).name}"
    # inputs come from the fixture file, not from numpy's RNG
    np.testing.assert_array_equal(golden[f"{c['name']}/slowness"], c["slowness"])
    r = solve(oracle, c, dt, golden[f"{c['name']}/slowness"])
    assert r["niter"] == int(golden[key + "/niter"])
    np.testing.assert_array_equal(r["tt"], golden[key + "/tt"])
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + "/tt_rcv"])


WENO = [(c, dt) for c, dt in ALL if cases.weno_ok(c)]


@pytest.mark.parametrize("c,dt", WENO, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in WENO])
def test_oracle_weno_matches_golden(oracle, golden, c, dt):
    """two-stage solve (first order, then WENO3 sweeps): field, both iteration counts, receivers"""
    key = f"{c['name']}/{np.dtype(dt).name}"
    r = solve(oracle, c, dt, golden[f"{c['name']}/slowness"], weno=True)
    assert r["niter"] == int(golden[key + "/weno_niter"])
    assert r["niterw"] == int(golden[key + "/weno_niterw"])
    np.testing.assert_array_equal(r["tt"], golden[key + "/weno_tt"])
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + "/weno_tt_rcv"])


ROT = [(c, dt) for c, dt in ALL if cases.rot_ok(c)]


@pytest.mark.parametrize("c,dt", ROT, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in ROT])
def test_oracle_rotated_template_matches_golden(oracle, golden, c, dt):
    """rotated_template=True: sweep45 (stencil rotated by pi/4) after every first-order sweep"""
    key = f"{c['name']}/{np.dtype(dt).name}"
    r = oracle.solve2d(dt, c["ncells"], c["dx"], c["dz"], c["origin"], golden[f"{c['name']}/slowness"], c["src"],
                       c["t0"], cell_slowness=c["cell_slowness"], rcv=c["rcv"], rotated=True)
    assert r["niter"] == int(golden[key + "/rot_niter"])
    np.testing.assert_array_equal(r["tt"], golden[key + "/rot_tt"])
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + "/rot_tt_rcv"])
    assert not np.array_equal(r["tt"], golden[key + "/tt"])  # the stage really changes the field


RP = [(c, dt) for c, dt in ALL if cases.rp_ok(c)]


@pytest.mark.parametrize("c,dt", RP, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in RP])
@pytest.mark.parametrize("tag,iv", [("rp", False), ("rpv", True)])
def test_oracle_tt_from_raypath_matches_golden(oracle, golden, c, dt, tag, iv):
    """weno=True + tt_from_rp=True (ttcrpy's 3-D defaults): traveltimes integrated along the ray"""
    key = f"{c['name']}/{np.dtype(dt).name}"
    kw = dict(cell_slowness=c["cell_slowness"], translate=c["translate"], rcv=c["rcv"], weno=True, tt_from_rp=True,
              interp_vel=iv)
    if int(golden[key + f"/{tag}_error"]):
        with pytest.raises(RuntimeError, match="going outside grid"):
            oracle.solve3d(dt, c["ncells"], c["dx"], c["origin"], golden[f"{c['name']}/slowness"], c["src"], c["t0"], **kw)
        return
    r = oracle.solve3d(dt, c["ncells"], c["dx"], c["origin"], golden[f"{c['name']}/slowness"], c["src"], c["t0"], **kw)
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + f"/{tag}_tt_rcv"])


@pytest.mark.parametrize("c,dt", RP, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in RP])
def test_oracle_raypaths_match_golden(oracle, golden, c, dt):
    """return_rays: Grid3Drn::getRaypath(Tx,t0,Rx,r_data,tt,threadNo) -- every point of every ray, bit-exact"""
    key = f"{c['name']}/{np.dtype(dt).name}"
    kw = dict(cell_slowness=c["cell_slowness"], translate=c["translate"], rcv=c["rcv"], weno=True, return_rays=True)
    if int(golden[key + "/rays_error"]):
        with pytest.raises(RuntimeError, match="going outside grid"):
            oracle.solve3d(dt, c["ncells"], c["dx"], c["origin"], golden[f"{c['name']}/slowness"], c["src"], c["t0"], **kw)
        return
    r = oracle.solve3d(dt, c["ncells"], c["dx"], c["origin"], golden[f"{c['name']}/slowness"], c["src"], c["t0"], **kw)
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + "/rays_tt_rcv"])
    off, pts = golden[key + "/rays_off"], golden[key + "/rays_pts"]
    assert len(r["rays"]) == off.size - 1
    for n, ray in enumerate(r["rays"]):
        np.testing.assert_array_equal(ray, pts[off[n]:off[n + 1]])
        np.testing.assert_array_equal(ray[0], np.asarray(c["rcv"][n], dtype=dt))   # starts on the receiver


RP2 = [(c, dt) for c, dt in ALL if cases.rp2_ok(c)]


@pytest.mark.parametrize("c,dt", RP2, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in RP2])
def test_oracle_raypaths_2d_match_golden(oracle, golden, c, dt):
    """2-D: Grid2Drn::getTraveltimeFromRaypath (tt_from_rp) and Grid2Drn::getRaypath (return_rays), node and cell grids"""
    key = f"{c['name']}/{np.dtype(dt).name}"
    kw = dict(cell_slowness=c["cell_slowness"], rcv=c["rcv"], weno=cases.weno_ok(c))
    args = (dt, c["ncells"], c["dx"], c["dz"], c["origin"], golden[f"{c['name']}/slowness"], c["src"], c["t0"])
    assert int(golden[key + "/rp2_error"]) == 0 and int(golden[key + "/rays2_error"]) == 0
    r = oracle.solve2d(*args, tt_from_rp=True, **kw)
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + "/rp2_tt_rcv"])
    r = oracle.solve2d(*args, return_rays=True, **kw)
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + "/rays2_tt_rcv"])
    off, pts = golden[key + "/rays2_off"], golden[key + "/rays2_pts"]
    assert len(r["rays"]) == off.size - 1
    for n, ray in enumerate(r["rays"]):
        np.testing.assert_array_equal(ray, pts[off[n]:off[n + 1]])


def test_golden_covers_multi_iteration_cases(golden):
    # the stopping rule / sweep order is only exercised when iterations >= 2 do real work
    assert int(golden["random_24x20x28_node/float32/niter"]) >= 4
    assert int(golden["random2d_64x96/float32/niter"]) >= 5


def analytic_gradient(src, pts):
    """t = |acosh(1 + b^2 r^2 / (2 Va Vb)) / b|  (tests/files/sol_analytique_gradient.py:53-55)"""
    a, b = cases.A, cases.B
    va = a + b * src[2]
    vb = a + b * pts[:, 2]
    r2 = np.sum((pts - src) ** 2, axis=1)
    return np.abs(np.arccosh(1 + b * b * r2 / (2 * va * vb)) / b)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_reference_accuracy_bar_gradient41(oracle, dt):
    """The reference's own bar (tests/test_grid3d.cpp:181-199): mean relative error vs the analytic
    solution < 1 % at the 441 rcv.dat lattice points, gradient medium model, source at the origin,
    weno3=1 -- replayed here with the WENO stage (the first-order solver alone meets a looser 3 %)."""
    n = 41
    dx = 20.0 / (n - 1)
    rcv = cases.rcv_lattice3d()
    r = oracle.solve3d(dt, (n - 1,) * 3, dx, (0, 0, 0), cases.gradient3d((n,) * 3, dx), [[0, 0, 0]], rcv=rcv)
    ana = analytic_gradient(np.zeros(3), rcv)
    m = ana > 0
    err = np.mean(np.abs(r["tt_rcv"][m] - ana[m]) / ana[m])
    assert err < 0.03
    rw = oracle.solve3d(dt, (n - 1,) * 3, dx, (0, 0, 0), cases.gradient3d((n,) * 3, dx), [[0, 0, 0]], rcv=rcv, weno=True)
    errw = np.mean(np.abs(rw["tt_rcv"][m] - ana[m]) / ana[m])
    assert errw < 0.01 and errw < err


def test_constant_c1_analytic(oracle):
    """BASELINE config C1: 64^3 cells constant slowness, source at the centre node: t = s * r
    (tests/accuracy_grid3d.cpp:313-328)."""
    n = 65
    s0 = 1.0 / 3.0
    r = oracle.solve3d(np.float64, (64,) * 3, 1.0, (0, 0, 0), np.full(n ** 3, s0), [[32.0, 32.0, 32.0]])
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    dist = np.sqrt((i - 32.0) ** 2 + (j - 32.0) ** 2 + (k - 32.0) ** 2).ravel()
    m = dist > 0
    err = np.mean(np.abs(r["tt"][m] - s0 * dist[m]) / (s0 * dist[m]))
    assert err < 0.05
