"""GPU parity (run with -m gpu on the MI355X box): the HIP path, called through the C ABI
(ttcr_amd.Grid3d/Grid2d -> libttcr_amd.so), against
  (a) the committed golden vectors produced by the unmodified compiled reference, and
  (b) the CPU oracle on the same inputs.
Bar: BIT-EXACT traveltime fields, receiver values and iteration counts, float32 and float64
(north_star only asks for 1e-5 s RMS; the kernels mirror the reference's rounding exactly)."""
import os

import numpy as np
import pytest

import cases
from gpu_util import run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[(1, 0), (1, 1), (0, 0), (2, 0), (2, 1)],
                ids=["persistent", "persistent+skip", "wavefront-launches", "overlapped-sweeps", "overlapped-sweeps+skip"],
                autouse=True)
def sweep_mode(request, monkeypatch):
    """every parity test runs with each sweep driver (ttcr_fsm_set_option "mode" / "skip")"""
    monkeypatch.setenv("TTCR_FSM_MODE", str(request.param[0]))
    monkeypatch.setenv("TTCR_FSM_SKIP", str(request.param[1]))
    return request.param


ALL = [(c, dt) for c in cases.cases3d() + cases.cases2d() for dt in (np.float32, np.float64)]
IDS = [f"{c['name']}-{np.dtype(dt).name}" for c, dt in ALL]


@pytest.mark.parametrize("c,dt", ALL, ids=IDS)
def test_hip_matches_golden_bit_exact(golden, c, dt):
    key = f"{c['name']}/{np.dtype(dt).name}"
    c = dict(c, slowness=golden[f"{c['name']}/slowness"])
    r = run_case(c, dt)
    ref = golden[key + "/tt"]
    rms = float(np.sqrt(np.mean((r["tt"].astype(np.float64) - ref.astype(np.float64)) ** 2)))
    assert rms <= 1e-5, f"RMS {rms} vs reference exceeds north_star tolerance 1e-5 s"
    assert r["niter"] == int(golden[key + "/niter"])
    np.testing.assert_array_equal(r["tt"], ref)
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + "/tt_rcv"])


WENO = [(c, dt) for c, dt in ALL if cases.weno_ok(c)]


@pytest.mark.parametrize("c,dt", WENO, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in WENO])
def test_hip_weno_matches_golden_bit_exact(golden, c, dt):
    """weno=True (the ttcrpy default): first-order sweeps then WENO3 sweeps, both iteration counts"""
    key = f"{c['name']}/{np.dtype(dt).name}"
    c = dict(c, slowness=golden[f"{c['name']}/slowness"])
    r = run_case(c, dt, weno=1)
    assert r["niter"] == int(golden[key + "/weno_niter"])
    assert r["niterw"] == int(golden[key + "/weno_niterw"])
    np.testing.assert_array_equal(r["tt"], golden[key + "/weno_tt"])
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + "/weno_tt_rcv"])


ROT = [(c, dt) for c, dt in ALL if cases.rot_ok(c)]


@pytest.mark.parametrize("c,dt", ROT, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in ROT])
def test_hip_rotated_template_matches_golden_bit_exact(golden, c, dt):
    """rotated_template=True: Grid2Drn::sweep45 after every first-order sweep (ttcr/Grid2Drnfs.h:277-286)"""
    key = f"{c['name']}/{np.dtype(dt).name}"
    c = dict(c, slowness=golden[f"{c['name']}/slowness"])
    r = run_case(c, dt, rotated=1)
    assert r["niter"] == int(golden[key + "/rot_niter"])
    np.testing.assert_array_equal(r["tt"], golden[key + "/rot_tt"])
    np.testing.assert_array_equal(r["tt_rcv"], golden[key + "/rot_tt_rcv"])


@pytest.mark.parametrize("dt,shape", [(np.float32, (1201, 150)), (np.float64, (600, 210)), (np.float32, (254, 700)),
                                      (np.float64, (511, 300))],
                         ids=["f32-2strips", "f64-2strips", "f32-1strip-256", "f64-2strips-edge"])
@pytest.mark.parametrize("kernel", ["rows", "strips"])
def test_hip_rotated_template_strips_vs_oracle(oracle, monkeypatch, dt, shape, kernel):
    """larger grids, random medium, off-node source, two slots solved concurrently: bit-exact vs the CPU oracle,
    with the row-parallel sweep45 kernel and with the strip kernel kept for grids too wide for its LDS rows
    (strips of 1022 / 510 columns: the shapes cross a strip boundary)"""
    import ttcr_amd

    monkeypatch.setenv("TTCR_FSM_SWEEP45", kernel)

    nx, nz = shape
    rng = np.random.default_rng(77)
    s = rng.uniform(0.25, 1.0, shape)
    x, z = np.arange(nx) * 0.25, 1.0 + np.arange(nz) * 0.25
    g = ttcr_amd.Grid2d(x, z, n_threads=2, cell_slowness=0, method="FSM", weno=0, rotated_template=1, dtype=dt)
    srcs = np.array([[x[-1] * 0.71, 1.0 + 3.3], [x[3], z[5]]])
    rcv = np.array([[0.0, 1.0], [x[-1], z[-1]], [x[7] + 0.1, z[9] + 0.05]])
    g.raytrace(np.repeat(srcs, 3, axis=0), np.tile(rcv, (2, 1)), slowness=s)
    for k in range(2):
        r = oracle.solve2d(dt, (nx - 1, nz - 1), g.dx, g.dz, (x[0], z[0]), s.astype(dt).ravel(), [srcs[k]], rotated=True)
        assert g.get_niter(k) == r["niter"]
        np.testing.assert_array_equal(g.get_grid_traveltimes(k).ravel(), r["tt"])


def test_hip_rotated_template_ignored_like_the_reference(golden):
    """with weno=True or dx != dz the reference never calls sweep45: the flag changes nothing"""
    for name, weno in (("random2d_xz", 0), ("grad2d_65_off", 1)):
        c = next(c for c in cases.cases2d() if c["name"] == name)
        c = dict(c, slowness=golden[f"{name}/slowness"])
        a, b = run_case(c, np.float32, weno=weno, rotated=1), run_case(c, np.float32, weno=weno, rotated=0)
        np.testing.assert_array_equal(a["tt"], b["tt"])
        assert (a["niter"], a["niterw"]) == (b["niter"], b["niterw"])


RP = [(c, dt) for c, dt in ALL if cases.rp_ok(c)]


@pytest.mark.parametrize("c,dt", RP, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in RP])
@pytest.mark.parametrize("tag,iv", [("rp", 0), ("rpv", 1)])
def test_hip_tt_from_raypath_matches_golden(golden, c, dt, tag, iv):
    """ttcrpy's 3-D defaults (weno=1, tt_from_rp=1): receiver traveltimes integrated along the ray
    traced back through the field, bit-exact; same error when the reference's ray leaves the grid"""
    import ttcr_amd
    from gpu_util import source_array

    key = f"{c['name']}/{np.dtype(dt).name}"
    nc, o = c["ncells"], c["origin"]
    x, y, z = (o[a] + np.arange(nc[a] + 1) * c["dx"] for a in range(3))
    g = ttcr_amd.Grid3d(x, y, z, cell_slowness=c["cell_slowness"], method="FSM", tt_from_rp=1, weno=1, interp_vel=iv,
                        translate_grid=c["translate"], dtype=dt)
    s = np.asarray(golden[f"{c['name']}/slowness"]).reshape(g.shape, order="F")
    if int(golden[key + f"/{tag}_error"]):
        with pytest.raises(RuntimeError, match="going outside grid"):
            g.raytrace(source_array(c), c["rcv"], slowness=s, aggregate_src=True)
        return
    tt = g.raytrace(source_array(c), c["rcv"], slowness=s, aggregate_src=True)
    np.testing.assert_array_equal(tt, golden[key + f"/{tag}_tt_rcv"])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_hip_matches_oracle_on_fresh_inputs(oracle, dt):
    """inputs that are NOT in the golden file: random slowness, random off-node source"""
    rng = np.random.default_rng(99)
    nn = (37, 29, 45)
    nc = tuple(v - 1 for v in nn)
    c = dict(name="fresh", dim=3, ncells=nc, dx=0.37, origin=(1.0, -2.0, 0.5), cell_slowness=False,
             slowness=rng.uniform(0.2, 1.5, nn[0] * nn[1] * nn[2]), translate=False,
             src=np.array([[1.0 + 5.123, -2.0 + 3.77, 0.5 + 9.01]]), t0=np.array([0.25]),
             rcv=np.array([[1.0 + 2.0, -2.0 + 2.0, 0.5 + 2.0], [1.5, -1.5, 1.0]]))
    r = run_case(c, dt)
    # the Python layer derives dx from the node coordinates, dx = x[1]-x[0] in the grid dtype
    # (rgrid.pyx:170-172); hand the oracle the very same number
    o = oracle.solve3d(dt, nc, r["grid"].dx, c["origin"], c["slowness"], c["src"], c["t0"], rcv=c["rcv"])
    assert r["niter"] == o["niter"]
    np.testing.assert_array_equal(r["tt"], o["tt"])
    np.testing.assert_array_equal(r["tt_rcv"], o["tt_rcv"])


@pytest.mark.parametrize("kind", ["gradient129", "random97"])
def test_hip_matches_oracle_medium_grids(oracle, kind):
    """Hundreds of patches, several chunks per patch: exercises the inter-patch hand-off and the
    wave-uniform fast paths of the local solver at scale (float32, bit-exact vs the oracle)."""
    rng = np.random.default_rng(5)
    if kind == "gradient129":
        nn = (129, 129, 129)
        dx = 20.0 / 128
        s = cases.gradient3d(nn, dx)
        src = np.array([[7.3, 11.2, 5.9]])
    else:
        nn = (97, 83, 91)
        dx = 0.25
        s = rng.uniform(0.2, 1.2, nn[0] * nn[1] * nn[2])
        src = np.array([[3.3, 17.1, 9.02]])
    nc = tuple(v - 1 for v in nn)
    c = dict(name=kind, dim=3, ncells=nc, dx=dx, origin=(0.0, 0.0, 0.0), cell_slowness=False, slowness=s,
             translate=False, src=src, t0=np.array([0.0]), rcv=np.array([[1.0, 2.0, 3.0]]))
    r = run_case(c, np.float32)
    o = oracle.solve3d(np.float32, nc, r["grid"].dx, c["origin"], s, src, rcv=c["rcv"])
    assert r["niter"] == o["niter"]
    np.testing.assert_array_equal(r["tt"], o["tt"])


@pytest.mark.parametrize("kind", ["gradient3d_49x41x45", "layers2d_300x210", "gradient2d_xz_290x180"])
def test_hip_weno_matches_oracle_multi_patch(oracle, kind):
    """WENO stage across several patches per axis (2-column halos exchanged between workgroups)."""
    if kind.startswith("gradient3d"):
        nn = (49, 41, 45)
        dx = 0.4
        nc = tuple(v - 1 for v in nn)
        s = cases.gradient3d(nn, dx)
        c = dict(name=kind, dim=3, ncells=nc, dx=dx, origin=(0.0, 0.0, 0.0), cell_slowness=False, slowness=s,
                 translate=False, src=np.array([[7.3, 11.2, 5.9]]), t0=np.array([0.0]), rcv=np.array([[1.0, 2.0, 3.0]]))
        r = run_case(c, np.float32, weno=1)
        o = oracle.solve3d(np.float32, nc, r["grid"].dx, c["origin"], s, c["src"], rcv=c["rcv"], weno=True)
    else:
        xz = "xz" in kind
        nn = (290, 180) if xz else (300, 210)
        dx, dz = (0.1, 0.15) if xz else (0.1, 0.1)
        nc = tuple(v - 1 for v in nn)
        s = cases.gradient2d(nn, dz) if xz else np.tile(1.0 / (1.0 + 0.1 * np.floor(np.arange(nn[1]) * dz)), nn[0])
        c = dict(name=kind, dim=2, ncells=nc, dx=dx, dz=dz, origin=(0.0, 0.0), cell_slowness=False, slowness=s,
                 translate=False, src=np.array([[7.31, 11.27]]), t0=np.array([0.0]), rcv=np.array([[1.0, 2.0]]))
        r = run_case(c, np.float32, weno=1)
        o = oracle.solve2d(np.float32, nc, r["grid"].dx, r["grid"].dz, c["origin"], s, c["src"], rcv=c["rcv"], weno=True)
    assert (r["niter"], r["niterw"]) == (o["niter"], o["niterw"])
    np.testing.assert_array_equal(r["tt"], o["tt"])


@pytest.mark.parametrize("dim", [3, 2])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_hip_weno_operands_outside_the_tame_range(oracle, dim, dt):
    """Slowness contrasts of 1e12: the second differences of the WENO stencils leave the range in which the fp32 kernel writes its
    divisions out (weno_axes, fsm_kernels.h) and the wavefronts concerned take the compiler's IEEE divisions -- same bits."""
    if dim == 3:
        nn = (37, 33, 35)
        s = np.ones(nn); s[10:20, 8:30, 5:25] = 3e12; s[25:30, :, :] = 7e9
        c = dict(name="untame3d", dim=3, ncells=tuple(v - 1 for v in nn), dx=0.5, origin=(0.0, 0.0, 0.0), cell_slowness=False,
                 slowness=s.flatten("F"), translate=False, src=np.array([[2.3, 4.1, 3.9]]), t0=np.array([0.0]), rcv=np.array([[1.0, 2.0, 3.0]]))
        r = run_case(c, dt, weno=1)
        o = oracle.solve3d(dt, c["ncells"], r["grid"].dx, c["origin"], c["slowness"], c["src"], rcv=c["rcv"], weno=True)
    else:
        nn = (150, 90)
        s = np.ones(nn); s[40:70, 20:60] = 3e12; s[100:110, :] = 7e9
        c = dict(name="untame2d", dim=2, ncells=tuple(v - 1 for v in nn), dx=0.5, dz=0.5, origin=(0.0, 0.0), cell_slowness=False,
                 slowness=s.ravel(), translate=False, src=np.array([[2.3, 4.1]]), t0=np.array([0.0]), rcv=np.array([[1.0, 2.0]]))
        r = run_case(c, dt, weno=1)
        o = oracle.solve2d(dt, c["ncells"], r["grid"].dx, r["grid"].dz, c["origin"], c["slowness"], c["src"], rcv=c["rcv"], weno=True)
    assert (r["niter"], r["niterw"]) == (o["niter"], o["niterw"])
    assert o["niterw"] > 1 and np.nanmax(o["tt"]) > 1e12
    np.testing.assert_array_equal(r["tt"], o["tt"])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("pair", ["0", "1"], ids=["unpaired", "pairs"])
def test_hip_weno_wide_batch(oracle, monkeypatch, dt, pair):
    """16 sources in 16 slots swept together, one field per slot (the default of weno grids: 8-level chunks) and as
    8 slot groups of the pair layout (the WENO stage then runs its short-chunk kernel, 4 levels per chunk); every
    field and both iteration counts must equal the oracle's"""
    import ttcr_amd

    monkeypatch.setenv("TTCR_FSM_PAIR", pair)

    rng = np.random.default_rng(23)
    nn = (37, 41, 33)
    x, y, z = (np.arange(n) * 0.5 for n in nn)
    X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
    s = (1.0 / (1.0 + 0.05 * Z)) * (1.0 + 0.2 * np.sin(0.4 * X) * np.cos(0.3 * Y))
    srcs = np.column_stack([rng.uniform(0.5, 17.5, 16), rng.uniform(0.5, 19.5, 16), rng.uniform(0.5, 15.5, 16)])
    srcs[3] = [4.0, 6.5, 8.0]   # one source on a node
    rcv1 = np.array([[1.0, 2.0, 3.0], [18.0, 20.0, 16.0]])
    g = ttcr_amd.Grid3d(x, y, z, n_threads=16, cell_slowness=0, method="FSM", tt_from_rp=0, weno=1, maxit=15, dtype=dt)
    tt = g.raytrace(np.repeat(srcs, 2, axis=0), np.tile(rcv1, (16, 1)), slowness=s)
    for k in (0, 3, 7, 8, 15):
        o = oracle.solve3d(dt, tuple(n - 1 for n in nn), g.dx, (0, 0, 0), s.flatten("F"), [srcs[k]], rcv=rcv1,
                           weno=True, maxit=15)
        np.testing.assert_array_equal(g.get_grid_traveltimes(k).flatten("F"), o["tt"])
        assert (g.get_niter(k), g.get_niterw(k)) == (o["niter"], o["niterw"])
        np.testing.assert_array_equal(tt[2 * k:2 * k + 2], o["tt_rcv"])


@pytest.mark.parametrize("weno", [0, 1], ids=["first-order", "weno3"])
@pytest.mark.parametrize("n_threads", [2, 3])
@pytest.mark.parametrize("pair", ["1", "0"], ids=["pairs", "unpaired"])
def test_hip_multi_source_batches(oracle, monkeypatch, weno, n_threads, pair):
    """Several sources solved concurrently (slots = the reference's threads; two sources of a slot
    group are marched together from an interleaved field -- the default of the first-order 3-D solver -- or every slot
    has its own field -- the default with weno=1): every slot's field, iteration counts and receiver values must
    equal the single-source solves of the oracle."""
    import ttcr_amd

    monkeypatch.setenv("TTCR_FSM_PAIR", pair)

    rng = np.random.default_rng(17)
    nn = (41, 37, 29)
    dx = 0.5
    x, y, z = (np.arange(n) * dx for n in nn)
    s = rng.uniform(0.3, 1.0, nn)
    srcs = np.array([[3.3, 4.1, 5.7], [10.0, 9.0, 2.0], [19.9, 0.1, 13.9], [0.0, 18.0, 14.0], [7.7, 7.7, 7.7]])
    rcv1 = np.array([[1.0, 2.0, 3.0], [20.0, 18.0, 14.0], [5.5, 0.0, 7.25]])
    g = ttcr_amd.Grid3d(x, y, z, n_threads=n_threads, cell_slowness=0, method="FSM", tt_from_rp=0, weno=weno,
                        maxit=12, dtype=np.float32)
    source = np.repeat(srcs, len(rcv1), axis=0)
    rcv = np.tile(rcv1, (len(srcs), 1))
    tt = g.raytrace(source, rcv, slowness=s)
    want = []
    outs = []
    for p in srcs:
        o = oracle.solve3d(np.float32, tuple(n - 1 for n in nn), g.dx, (0, 0, 0), s.flatten("F"), [p], rcv=rcv1,
                           weno=bool(weno), maxit=12)
        outs.append(o)
        want.append(o["tt_rcv"])
    np.testing.assert_array_equal(tt, np.concatenate(want))
    # block distribution of the 5 sources over the slots (get_blk_size): slot b holds the LAST source of its block
    from ttcr_amd.dist import blk_sizes
    sizes = blk_sizes(len(srcs), n_threads)
    last = np.cumsum(sizes) - 1
    for slot, n in enumerate(last):
        np.testing.assert_array_equal(g.get_grid_traveltimes(slot).flatten("F"), outs[n]["tt"])
        assert g.get_niter(slot) == outs[n]["niter"] and g.get_niterw(slot) == outs[n]["niterw"]


@pytest.mark.parametrize("pair", ["0", "1"], ids=["unpaired", "pairs"])
def test_hip_multi_source_2d(oracle, monkeypatch, pair):
    """2-D slots have their own fields by default (the one-wave patches gain nothing from pairs); the pair layout
    stays selectable (TTCR_FSM_PAIR=1) and exact"""
    import ttcr_amd

    monkeypatch.setenv("TTCR_FSM_PAIR", pair)

    rng = np.random.default_rng(23)
    nn = (150, 70)
    x, z = np.arange(nn[0]) * 0.2, np.arange(nn[1]) * 0.2
    s = rng.uniform(0.3, 1.0, nn)
    srcs = np.array([[3.3, 4.1], [10.0, 9.0], [29.0, 0.1], [0.0, 13.0]])
    rcv1 = np.array([[1.0, 2.0], [29.8, 13.8]])
    g = ttcr_amd.Grid2d(x, z, n_threads=4, cell_slowness=0, method="FSM", weno=1, maxit=15, dtype=np.float64)
    tt = g.raytrace(np.repeat(srcs, 2, axis=0), np.tile(rcv1, (4, 1)), slowness=s)
    for n, p in enumerate(srcs):
        o = oracle.solve2d(np.float64, (nn[0] - 1, nn[1] - 1), g.dx, g.dz, (0, 0), s.ravel(), [p], rcv=rcv1, weno=True,
                           maxit=15)
        np.testing.assert_array_equal(tt[2 * n:2 * n + 2], o["tt_rcv"])
        np.testing.assert_array_equal(g.get_grid_traveltimes(n).ravel(), o["tt"])
        assert (g.get_niter(n), g.get_niterw(n)) == (o["niter"], o["niterw"])


@pytest.mark.parametrize("c,dt", RP, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in RP])
def test_hip_raypaths_match_golden(golden, c, dt):
    """raytrace(..., return_rays=True) -> (tt, rays): the raypaths of Grid3Drn::getRaypath, point for point,
    and the traveltimes integrated along them; same error when the reference's ray leaves the grid"""
    import ttcr_amd
    from gpu_util import source_array

    key = f"{c['name']}/{np.dtype(dt).name}"
    nc, o = c["ncells"], c["origin"]
    x, y, z = (o[a] + np.arange(nc[a] + 1) * c["dx"] for a in range(3))
    g = ttcr_amd.Grid3d(x, y, z, cell_slowness=c["cell_slowness"], method="FSM", tt_from_rp=0, weno=1,
                        translate_grid=c["translate"], dtype=dt)
    s = np.asarray(golden[f"{c['name']}/slowness"]).reshape(g.shape, order="F")
    if int(golden[key + "/rays_error"]):
        with pytest.raises(RuntimeError, match="going outside grid"):
            g.raytrace(source_array(c), c["rcv"], slowness=s, aggregate_src=True, return_rays=True)
        return
    tt, rays = g.raytrace(source_array(c), c["rcv"], slowness=s, aggregate_src=True, return_rays=True)
    np.testing.assert_array_equal(tt, golden[key + "/rays_tt_rcv"])
    off, pts = golden[key + "/rays_off"], golden[key + "/rays_pts"]
    assert len(rays) == off.size - 1
    for n, ray in enumerate(rays):
        np.testing.assert_array_equal(ray, pts[off[n]:off[n + 1]].astype(np.float64))
    # the option does not stick: the next plain call interpolates again (tt_from_rp=0)
    tt2 = g.raytrace(source_array(c), c["rcv"], aggregate_src=True)
    np.testing.assert_array_equal(tt2, golden[key + "/weno_tt_rcv"])


def test_hip_raypaths_several_sources_and_to_vtk(tmp_path, oracle):
    """two events in one call (rows come back in receiver order), rays written by to_vtk and read back"""
    import ttcr_amd
    from ttcr_amd import io

    n = 25
    x = np.arange(n) * 0.5
    s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n)))
    g = ttcr_amd.Grid3d(x, x, x, n_threads=2, cell_slowness=0, method="FSM", tt_from_rp=1, weno=1)
    src = np.array([[1.0, 1.0, 1.0], [10.2, 3.3, 7.1]])
    rcv = np.array([[11.0, 11.0, 2.0], [3.0, 9.5, 8.0], [6.0, 6.0, 11.5]])
    srows = np.vstack([src[[0, 1, 0]], src[[1, 0, 1]]])
    rrows = np.vstack([rcv, rcv])
    tt, rays = g.raytrace(srows, rrows, slowness=s, return_rays=True)
    assert len(rays) == 6
    for k in range(6):
        np.testing.assert_array_equal(rays[k][0], rrows[k])
        np.testing.assert_array_equal(rays[k][-1], srows[k])
        r = oracle.solve3d(np.float64, (n - 1,) * 3, 0.5, (0, 0, 0), s.flatten("F"), [srows[k]], rcv=[rrows[k]], weno=True,
                           return_rays=True)
        np.testing.assert_array_equal(rays[k], r["rays"][0])
        assert tt[k] == r["tt_rcv"][0]
    g.to_vtk({"rays": rays, "Travel time": g.get_grid_traveltimes(0)}, str(tmp_path / "out"))
    back = io.read_vtp_lines(str(tmp_path / "out_rays.vtp"))
    assert len(back) == 6
    for a, b in zip(rays, back):
        np.testing.assert_allclose(a, b, rtol=1e-6)   # vtkPoints are Float32


RP2 = [(c, dt) for c, dt in ALL if cases.rp2_ok(c)]


@pytest.mark.parametrize("c,dt", RP2, ids=[f"{c['name']}-{np.dtype(dt).name}" for c, dt in RP2])
def test_hip_raypaths_2d_match_golden(golden, c, dt):
    """Grid2d(tt_from_rp=1) and raytrace(return_rays=True) in 2-D (node and cell grids, dx != dz): traveltimes and
    every ray point bit-exact vs the reference"""
    import ttcr_amd
    from gpu_util import source_array

    key = f"{c['name']}/{np.dtype(dt).name}"
    nc, o = c["ncells"], c["origin"]
    x, z = o[0] + np.arange(nc[0] + 1) * c["dx"], o[1] + np.arange(nc[1] + 1) * c["dz"]
    w = int(cases.weno_ok(c))
    g = ttcr_amd.Grid2d(x, z, cell_slowness=c["cell_slowness"], method="FSM", weno=w, tt_from_rp=1, dtype=dt)
    s = np.asarray(golden[f"{c['name']}/slowness"]).reshape(g.shape)
    tt = g.raytrace(source_array(c), c["rcv"], slowness=s, aggregate_src=True)
    np.testing.assert_array_equal(tt, golden[key + "/rp2_tt_rcv"])
    tt, rays = g.raytrace(source_array(c), c["rcv"], aggregate_src=True, return_rays=True)
    np.testing.assert_array_equal(tt, golden[key + "/rays2_tt_rcv"])
    off, pts = golden[key + "/rays2_off"], golden[key + "/rays2_pts"]
    assert len(rays) == off.size - 1
    for n, ray in enumerate(rays):
        assert ray.shape[1] == 2
        np.testing.assert_array_equal(ray, pts[off[n]:off[n + 1]].astype(np.float64))
    g.set_traveltime_from_raypath(False)
    tt0 = g.raytrace(source_array(c), c["rcv"], aggregate_src=True)
    np.testing.assert_array_equal(tt0, golden[key + ("/weno_tt_rcv" if w else "/tt_rcv")])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(5, 7, 9), (33, 34, 70), (64, 31, 32), (2, 2, 130)])
def test_hip_c_order_model_upload(dt, shape):
    """set_slowness / set_velocity hand the (nx, ny, nz) array over in C order and the device permutes it
    (ttcr_fsm_set_slowness_c_order): same solver state as the host-side flatten('F') of rgrid.pyx:562-565 through
    the plain entry point -- node slowness read back, and for a cell grid the solved field, bit for bit"""
    import ttcr_amd
    from ttcr_amd import _lib
    from ttcr_amd.rgrid import _ptr

    rng = np.random.default_rng(sum(shape))
    axes = [np.arange(n + 1) * 0.5 for n in shape]
    # node grid: what the solver holds is the array itself
    a = rng.uniform(0.2, 1.0, [n + 1 for n in shape])
    g = ttcr_amd.Grid3d(*axes, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=dt)
    g.set_slowness(a)
    np.testing.assert_array_equal(g.get_slowness(), a.astype(dt))
    g.set_slowness(a.ravel())                    # flat C order in
    np.testing.assert_array_equal(g.get_slowness(), a.astype(dt))
    g.set_velocity(1.0 / a)
    np.testing.assert_array_equal(g.get_slowness(), (1.0 / (1.0 / a)).astype(dt))
    # cell grid: against the plain entry point fed with the host-side permutation
    c = rng.uniform(0.2, 1.0, shape)
    src = np.array([[axes[0][1], axes[1][1], axes[2][2]]])
    rcv = np.array([[axes[0][-1], axes[1][-1], axes[2][-1]]])
    fields = []
    for plain in (False, True):
        gc = ttcr_amd.Grid3d(*axes, cell_slowness=1, method="FSM", tt_from_rp=0, weno=0, dtype=dt)
        if plain:
            s = np.ascontiguousarray(c.flatten("F"), dtype=dt)
            _lib.check(gc._lib.ttcr_fsm_set_slowness(gc._h, _ptr(s), s.size))
        else:
            gc.set_slowness(c)
        gc.raytrace(src, rcv)
        fields.append((gc.get_slowness(), gc.get_grid_traveltimes()))
    np.testing.assert_array_equal(fields[0][0], fields[1][0])
    np.testing.assert_array_equal(fields[0][1], fields[1][1])


@pytest.mark.parametrize("n_threads", [1, 2, 3])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_hip_raypath_traveltimes_of_a_batch(oracle, dt, n_threads):
    """tt_from_rp=1 (the ttcrpy default) with several sources per call: the receivers of all sources of a batch are
    walked in one launch (raypath_batch); every value must equal the per-source walk of the oracle.  Smooth medium
    (the reference's walk needs one); the second source sits on a receiver (t0 shortcut), sources carry origin times."""
    import ttcr_amd

    nn = (29, 25, 33)
    dx = 0.5
    x, y, z = (np.arange(n) * dx for n in nn)
    X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
    s = (1.0 / (1.0 + 0.08 * Z)) * (1.0 + 0.2 * np.exp(-((X - 6) ** 2 + (Y - 5) ** 2 + (Z - 8) ** 2) / 20.0))
    srcs = np.array([[0.25, 3.3, 4.1, 5.7], [0.0, 7.0, 6.0, 8.0], [0.5, 10.0, 9.0, 12.0], [0.0, 11.2, 3.3, 6.1], [0.125, 7.7, 7.7, 7.7]])
    rcv1 = np.array([[1.0, 2.0, 3.0], [14.0, 12.0, 16.0], [5.5, 0.0, 7.25], [7.0, 6.0, 8.0]])
    g = ttcr_amd.Grid3d(x, y, z, n_threads=n_threads, cell_slowness=0, method="FSM", tt_from_rp=1, weno=1, dtype=dt)
    tt = g.raytrace(np.repeat(srcs, len(rcv1), axis=0), np.tile(rcv1, (len(srcs), 1)), slowness=s)
    for n, p in enumerate(srcs):
        o = oracle.solve3d(dt, tuple(m - 1 for m in nn), dx, (0, 0, 0), s.flatten("F"), [p[1:]], [p[0]], rcv=rcv1, weno=True,
                           tt_from_rp=True)
        np.testing.assert_array_equal(tt[len(rcv1) * n:len(rcv1) * (n + 1)], o["tt_rcv"], err_msg=str(n))
    # 2-D twin (cell grid: the walk integrates the cell slowness)
    nc = (60, 44)
    x2, z2 = np.arange(nc[0] + 1) * 0.25, np.arange(nc[1] + 1) * 0.25
    zc = 0.5 * (z2[1:] + z2[:-1])
    s2 = np.broadcast_to(1.0 / (1.0 + 0.1 * zc), nc).copy()
    srcs2 = np.array([[0.5, 3.3, 4.1], [0.0, 10.0, 9.0], [0.25, 13.0, 2.1]])
    rcv2 = np.array([[1.0, 2.0], [14.8, 10.8], [10.0, 9.0]])
    g2 = ttcr_amd.Grid2d(x2, z2, n_threads=n_threads, cell_slowness=1, method="FSM", tt_from_rp=1, weno=1, dtype=dt)
    tt2 = g2.raytrace(np.repeat(srcs2, len(rcv2), axis=0), np.tile(rcv2, (len(srcs2), 1)), slowness=s2)
    for n, p in enumerate(srcs2):
        o = oracle.solve2d(dt, nc, 0.25, 0.25, (0, 0), s2.ravel(), [p[1:]], [p[0]], cell_slowness=True, rcv=rcv2, weno=True,
                           tt_from_rp=True)
        np.testing.assert_array_equal(tt2[len(rcv2) * n:len(rcv2) * (n + 1)], o["tt_rcv"], err_msg="2d %d" % n)


def test_hip_receiver_near_the_last_plane_of_an_axis(oracle):
    """tests/golden/edge_receiver_case.npz (found by a long fuzz run): a ray point 6.6e-5 cell below the z-max face gets
    the cell index BEYOND the grid from the reference's (T2)(small + (p - min)/d) and the reference reads one node
    past its slowness array there (its own result changes from run to run).  Oracle and kernel clamp that index to
    the last node; everything in range is untouched.  Field, rays and traveltimes must agree bit for bit."""
    import os

    import ttcr_amd

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "edge_receiver_case.npz"))
    s, src, t0, rcv = d["s"], d["src"], d["t0"], d["rcv"]
    dx, org, nc = float(d["dx"]), tuple(d["org"]), tuple(int(v) for v in d["nc"])
    axes = [o + np.arange(n + 1) * dx for o, n in zip(org, nc)]
    o = oracle.solve3d(np.float64, nc, dx, org, s.flatten("F"), src, t0, return_rays=True, cell_slowness=False, rcv=rcv,
                       weno=True)
    g = ttcr_amd.Grid3d(*axes, cell_slowness=False, method="FSM", tt_from_rp=0, weno=1, dtype=np.float64)
    tt, rays = g.raytrace(np.hstack([t0[:, None], src]), rcv, slowness=s, aggregate_src=True, return_rays=True)
    np.testing.assert_array_equal(g.get_grid_traveltimes().flatten("F"), o["tt"])
    np.testing.assert_array_equal(tt, o["tt_rcv"])
    for a, b in zip(rays, o["rays"]):
        np.testing.assert_array_equal(a, b)
    assert abs(tt[0] - 9.245927) < 2e-6      # plain trilinear trapezoid along the same ray


@pytest.mark.parametrize("n_threads", [1, 2, 3])
def test_hip_raypaths_more_sources_than_slots(oracle, n_threads):
    """return_rays with MORE sources than slots and unequal receiver counts: the solve order is round-major
    (n_threads=2, 3 sources: 0, 2, 1), the rays must still come back one per receiver row, in row order."""
    import ttcr_amd

    n = 21
    x = np.arange(n) * 0.5
    s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n)))
    g = ttcr_amd.Grid3d(x, x, x, n_threads=n_threads, cell_slowness=0, method="FSM", tt_from_rp=1, weno=1)
    src = np.array([[1.0, 1.0, 1.0], [9.2, 3.3, 7.1], [5.0, 8.5, 2.2], [2.4, 7.9, 6.1], [7.7, 2.3, 4.4]])
    rcv = np.array([[9.0, 9.0, 2.0], [3.0, 9.5, 8.0], [6.0, 6.0, 9.5], [0.5, 0.25, 7.75]])
    # source n gets 1 + (n % 4) receivers, rows shuffled so that the rows of a source are not contiguous
    rows = [(n_, k) for n_ in range(len(src)) for k in range(1 + n_ % 4)]
    rng = np.random.default_rng(17)
    rows = [rows[i] for i in rng.permutation(len(rows))]
    srows = np.array([src[a] for a, _ in rows])
    rrows = np.array([rcv[b] for _, b in rows])
    tt, rays = g.raytrace(srows, rrows, slowness=s, return_rays=True)
    assert len(rays) == len(rows)
    for k in range(len(rows)):
        r = oracle.solve3d(np.float64, (n - 1,) * 3, 0.5, (0, 0, 0), s.flatten("F"), [srows[k]], rcv=[rrows[k]], weno=True,
                           return_rays=True)
        np.testing.assert_array_equal(rays[k], r["rays"][0], err_msg=f"row {k} (source {rows[k][0]})")
        assert tt[k] == r["tt_rcv"][0]


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_hip_receivers_next_to_the_last_planes(oracle, dt):
    """getTraveltime at receivers a rounding error / a fraction of 1e-4 cell below the last plane of every axis, dx > 1
    and dx not a power of two: the reference's index (rounded quotient + small) reaches the last node there while its
    absolute on-plane test fails, and it reads node index+1.  Oracle and kernel clamp that read to the last node
    (values in range untouched); both must agree bit for bit, in 3-D and in 2-D (cell grid: getCellNo as well)."""
    import ttcr_amd

    rng = np.random.default_rng(23)
    nn, dx = (9, 8, 7), 2.3
    axes = [np.arange(m) * dx for m in nn]
    # the far planes: xmax as the grid computes it (ttcr/Grid3Drn.h:73), not beyond the last axis value in the grid dtype
    # (the ttcrpy-style pre-check compares with the axes)
    hi = np.minimum(np.array([dt(0) + dt(m - 1) * dt(dx) for m in nn], dtype=dt), np.array([dt(a[-1]) for a in axes], dtype=dt))
    s = rng.uniform(0.3, 1.0, nn)
    pts = [hi.copy()]
    for ax in range(3):
        for back in (1, 2, 40, 400):
            p = hi.copy()
            for _ in range(back):
                p[ax] = np.nextafter(p[ax], dt(0))
            pts.append(p.copy())
            q = p.copy()
            q[(ax + 1) % 3] = dt(0.37 * hi[(ax + 1) % 3])
            pts.append(q)
    p = hi.copy()
    p -= dt(1.5e-4)
    pts.append(p)
    rcv = np.array(pts, dtype=np.float64)
    src = np.array([[2.0, 3.0, 1.5]])
    g = ttcr_amd.Grid3d(*axes, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=dt)
    tt = g.raytrace(src, rcv, slowness=s)
    o = oracle.solve3d(dt, tuple(m - 1 for m in nn), dx, (0, 0, 0), s.flatten("F"), src, rcv=rcv)
    np.testing.assert_array_equal(g._flat_tt(0), o["tt"])
    np.testing.assert_array_equal(tt, o["tt_rcv"])
    assert np.all(np.isfinite(tt))
    # 2-D, node grid with interpolation and cell grid with raypath traveltimes (getCellNo at segment mid-points)
    nn2, dx2, dz2 = (12, 10), 2.3, 3.1
    x2, z2 = np.arange(nn2[0]) * dx2, np.arange(nn2[1]) * dz2
    hi2 = np.minimum(np.array([dt(0) + dt(nn2[0] - 1) * dt(dx2), dt(0) + dt(nn2[1] - 1) * dt(dz2)], dtype=dt),
                     np.array([dt(x2[-1]), dt(z2[-1])], dtype=dt))
    pts2 = [hi2.copy()]
    for ax in range(2):
        for back in (1, 3, 60, 500):
            p = hi2.copy()
            for _ in range(back):
                p[ax] = np.nextafter(p[ax], dt(0))
            pts2.append(p.copy())
            q = p.copy()
            q[1 - ax] = dt(0.41 * hi2[1 - ax])
            pts2.append(q)
    pts2.append(hi2 - dt(1.7e-4))
    rcv2 = np.array(pts2, dtype=np.float64)
    src2 = np.array([[3.0, 4.0]])
    s2 = rng.uniform(0.3, 1.0, nn2)
    g2 = ttcr_amd.Grid2d(x2, z2, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=dt)
    tt2 = g2.raytrace(src2, rcv2, slowness=s2)
    o2 = oracle.solve2d(dt, (nn2[0] - 1, nn2[1] - 1), dx2, dz2, (0, 0), s2.ravel(), src2, rcv=rcv2)
    np.testing.assert_array_equal(tt2, o2["tt_rcv"])
    sc = 1.0 / (1.0 + 0.05 * np.arange(nn2[1] - 1))
    sc2 = np.ascontiguousarray(np.broadcast_to(sc[None, :], (nn2[0] - 1, nn2[1] - 1)))
    g3 = ttcr_amd.Grid2d(x2, z2, cell_slowness=1, method="FSM", tt_from_rp=1, weno=0, dtype=dt)
    keep = []   # receivers whose walk ends at the source (the reference throws / does not return for the others)
    for k, r in enumerate(rcv2):
        try:
            oracle.solve2d(dt, (nn2[0] - 1, nn2[1] - 1), dx2, dz2, (0, 0), sc2.ravel(), src2, rcv=[r], cell_slowness=True, tt_from_rp=True)
            keep.append(k)
        except RuntimeError:
            pass
    assert len(keep) >= 4
    rcv3 = rcv2[keep]
    tt3 = g3.raytrace(src2, rcv3, slowness=sc2)
    o3 = oracle.solve2d(dt, (nn2[0] - 1, nn2[1] - 1), dx2, dz2, (0, 0), sc2.ravel(), src2, rcv=rcv3, cell_slowness=True,
                        tt_from_rp=True)
    np.testing.assert_array_equal(tt3, o3["tt_rcv"])


def test_hip_one_grid_from_several_host_threads(oracle):
    """ttcrpy's raytrace(..., thread_no=k) pattern: several host threads, one grid, one slot each.  Calls on one handle
    are serialised by the library (ttcr_amd.h); every thread must get its own source's field and receiver values."""
    from concurrent.futures import ThreadPoolExecutor

    import ttcr_amd

    n = 40
    x = np.arange(n) * 0.5
    rng = np.random.default_rng(31)
    s = rng.uniform(0.3, 1.0, (n, n, n))
    nthr = 6
    g = ttcr_amd.Grid3d(x, x, x, n_threads=nthr, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
    g.set_slowness(s)
    srcs = rng.uniform(0.5, 19.0, (nthr, 3))
    rcv = rng.uniform(0.0, 19.5, (7, 3))

    def work(k):
        out = []
        for _ in range(3):
            tt = g.raytrace(srcs[k:k + 1], rcv, thread_no=k)
            out.append((tt, g._flat_tt(k), g.get_niter(k)))
        return out

    with ThreadPoolExecutor(max_workers=nthr) as ex:
        res = list(ex.map(work, range(nthr)))
    for k in range(nthr):
        o = oracle.solve3d(np.float32, (n - 1,) * 3, 0.5, (0, 0, 0), s.astype(np.float32).flatten("F"), srcs[k:k + 1], rcv=rcv)
        for tt, field, niter in res[k]:
            np.testing.assert_array_equal(field, o["tt"])
            np.testing.assert_array_equal(tt, o["tt_rcv"])
            assert niter == o["niter"]


def test_hip_device_views_of_a_field(monkeypatch):
    """ttcr_fsm_get_tt_device: n_nodes contiguous values; ttcr_fsm_get_tt_device_view: the field where it lies + stride
    (2 where two slots share an interleaved field: first-order 3-D grids with n_threads >= 2).  The raw device pointers are consumed the way a zero-copy consumer would:
    handed to another grid as device-resident input (set_slowness_device) and read back from there."""
    import ttcr_amd

    n = 20
    x = np.arange(n) * 0.5
    rng = np.random.default_rng(37)
    s = rng.uniform(0.3, 1.0, (n, n, n))
    nn = n ** 3
    sink = ttcr_amd.Grid3d(x, x, x, cell_slowness=0, method="FSM", dtype=np.float32)                 # nn values
    sink2 = ttcr_amd.Grid3d(np.arange(2 * n) * 0.5, x, x, cell_slowness=0, method="FSM", dtype=np.float32)   # 2 nn values
    # (one slot; three slots of a first-order grid: interleaved pairs; three slots with weno=1: one field per slot)
    # (pair: the sources of a call paired by distance -- a slot's field may lie in another slot's storage -- or not)
    # (source pairs are the layout of big batches -- slots x patches > 6 144 --; on this small grid they are asked for)
    for nthr, weno, want_stride, pair in ((1, 0, 1, 1), (3, 0, 2, 1), (3, 0, 2, 0), (3, 1, 1, 1), (3, 0, 1, 1)):
        if want_stride == 2:
            monkeypatch.setenv("TTCR_FSM_PAIR", "1")
        else:
            monkeypatch.delenv("TTCR_FSM_PAIR", raising=False)
        g = ttcr_amd.Grid3d(x, x, x, n_threads=nthr, cell_slowness=0, method="FSM", tt_from_rp=0, weno=weno, dtype=np.float32)
        g.set_option("pair_sources", pair)
        srcs = rng.uniform(0.5, 9.0, (nthr, 3))
        if pair and nthr == 3 and want_stride == 2:
            srcs[2] = srcs[0] + 0.2   # sources 0 and 2 end up in one pair
        g.raytrace(srcs, np.zeros((nthr, 3)), slowness=s)
        fields = [g._flat_tt(k) for k in range(nthr)]
        for slot in range(nthr):
            ptr, stride = g.tt_device_view(slot)
            assert stride == want_stride
            sink.set_slowness_device(g.tt_device_ptr(slot), nn)            # contiguous copy of the slot's field
            np.testing.assert_array_equal(sink.get_slowness().flatten("F"), fields[slot])
            if stride == 1:
                sink.set_slowness_device(ptr, nn)                           # the view IS the contiguous field
                np.testing.assert_array_equal(sink.get_slowness().flatten("F"), fields[slot])
        if want_stride == 2 and pair:
            p0, _ = g.tt_device_view(0)
            p1, _ = g.tt_device_view(1)
            p2, _ = g.tt_device_view(2)
            assert abs(p2 - p0) == 4 and abs(p1 - min(p0, p2)) == 2 * nn * 4   # 0 and 2 interleaved, 1 in the next group
            sink2.set_slowness_device(min(p0, p2), 2 * nn)
            both = sink2.get_slowness().flatten("F").reshape(nn, 2)
            np.testing.assert_array_equal(both[:, 0 if p0 < p2 else 1], fields[0])
            np.testing.assert_array_equal(both[:, 1 if p0 < p2 else 0], fields[2])
        if want_stride == 2 and not pair:
            p0, _ = g.tt_device_view(0)
            p1, _ = g.tt_device_view(1)
            p2, _ = g.tt_device_view(2)
            assert p1 - p0 == 4 and p2 - p0 == 2 * nn * 4                   # T[group][node][2]
            sink2.set_slowness_device(p0, 2 * nn)                           # the whole first pair, as it lies
            pair = sink2.get_slowness().flatten("F").reshape(nn, 2)
            np.testing.assert_array_equal(pair[:, 0], fields[0])
            np.testing.assert_array_equal(pair[:, 1], fields[1])


def test_hip_concurrent_single_source_calls_are_combined_and_isolated(oracle):
    """The way Grid3D's multi-source overload reaches a backend (ttcr/Grid3D.h:810-853): nt host threads, each calling the
    single-source entry point with its own slot.  The library gathers calls that arrive together into one device batch;
    every caller must get its own traveltimes / iteration count, and a call with a point outside the grid must fail
    alone (TTCR_ERR_RUNTIME + the reference's text in ITS thread's last_error), the others unharmed."""
    import ctypes as C
    import threading

    import ttcr_amd
    from ttcr_amd import _lib

    L = _lib.load()
    n = 36
    x = np.arange(n) * 0.5
    rng = np.random.default_rng(43)
    s = rng.uniform(0.3, 1.0, (n, n, n)).astype(np.float32)
    nthr = 8
    g = ttcr_amd.Grid3d(x, x, x, n_threads=nthr, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
    g.set_slowness(s)
    # (the default window is 200 us; Python threads leave the barrier one GIL hand-over after the other, so the test
    # gives them time -- the leader stops waiting as soon as every slot has a request)
    g.set_option("combine_window_us", 50000)
    srcs = rng.uniform(0.5, 17.0, (nthr, 3)).astype(np.float32)
    rcv = rng.uniform(0.0, 17.5, (5, 3)).astype(np.float32)
    bad = 3   # this caller's last receiver lies outside the grid
    results = [None] * nthr
    barrier = threading.Barrier(nthr)

    def work(k):
        tx = np.ascontiguousarray(srcs[k:k + 1])
        t0 = np.zeros(1, dtype=np.float32)
        rx = rcv.copy()
        if k == bad:
            rx[-1] = [5.0, 5.0, 17.6]
        out = np.full(5, -1.0, dtype=np.float32)
        barrier.wait()   # arrive together
        st = L.ttcr_fsm_raytrace(g._h, k, 1, tx.ctypes.data_as(C.c_void_p), t0.ctypes.data_as(C.c_void_p), 5,
                                 rx.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        results[k] = (st, L.ttcr_fsm_last_error().decode() if st else "", out)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(nthr)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(nthr):
        st, msg, out = results[k]
        if k == bad:
            assert st == _lib.ERR_RUNTIME and msg == "Error: Point (5 5 17.6) outside grid.", (st, msg)
            continue
        assert st == 0, msg
        o = oracle.solve3d(np.float32, (n - 1,) * 3, 0.5, (0, 0, 0), s.flatten("F"), srcs[k:k + 1], rcv=rcv)
        np.testing.assert_array_equal(out, o["tt_rcv"])
        np.testing.assert_array_equal(g._flat_tt(k), o["tt"])
        assert g.get_niter(k) == o["niter"]
    # the calls really went to the device together: the last batch held more than one source
    assert g.timing()["n_sources"] > 1


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_hip_end_game_that_leaves_the_grid(oracle, dt):
    """Found by the fuzz (seed 12, configuration 11 251): 8 x 7 cells of 2.3 x 0.125, a source of two points 1.5 apart.  The end game
    of a ray runs once per source point within a cell diagonal and moves curr_pt WITHOUT a bounds check (ttcr/Grid2Drn.h:1596-1655):
    the second run starts from a point outside the grid, the mid-point of its segment has a negative cell coordinate, and the
    reference converts that to an unsigned index -- undefined behaviour: the compiled reference returns traveltimes read from outside
    its cell array for receivers 0 and 3 and crashes for receiver 1.  HIP and the restatement define it the same way (negative ->
    cell 0, idx_u32 / FSM_U32) and must agree: traveltimes from raypaths and rays."""
    import ttcr_amd

    c = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "endgame_outside_case.npz"))
    nc = tuple(int(v) for v in c["nc"])
    dx, dz, org = float(c["dx"]), float(c["dz"]), tuple(float(v) for v in c["org"])
    axes = [org[0] + np.arange(nc[0] + 1) * dx, org[1] + np.arange(nc[1] + 1) * dz]
    source = np.hstack([c["t0"][:, None], c["src"]])
    kw = dict(dtype=dt, ncells=nc, dx=dx, dz=dz, origin=org, slowness=c["slowness"].ravel(), src=c["src"], t0=c["t0"], cell_slowness=True,
              rcv=c["rcv"], weno=True)
    for opt in (dict(tt_from_rp=True), dict(return_rays=True)):
        o = oracle.solve2d(**kw, **opt)
        g = ttcr_amd.Grid2d(*axes, cell_slowness=True, method="FSM", tt_from_rp=int(opt.get("tt_from_rp", False)), weno=1, dtype=dt)
        out = g.raytrace(source, c["rcv"], slowness=c["slowness"], aggregate_src=True, return_rays=opt.get("return_rays", False))
        tt = out[0] if isinstance(out, tuple) else out
        np.testing.assert_array_equal(tt, o["tt_rcv"])
        if "return_rays" in opt:
            for a, b in zip(out[1], o["rays"]):
                np.testing.assert_array_equal(a, b.astype(np.float64))
            assert sum(len(r) >= 3 and np.array_equal(r[-1], c["src"][1].astype(dt)) and np.array_equal(r[-2], c["src"][0].astype(dt))
                       for r in o["rays"]) == 3   # (both points of the source at the end of three of the four rays)
