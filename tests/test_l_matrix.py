"""The ray-projection matrix L of `compute_L` for 2-D FSM grids with cell slowness (Grid2D::raytrace(..., [r_data,] l_data, threadNo),
ttcr/Grid2D.h:583-640 -> Grid2Drn::getRaypath(..., l_data, ...), ttcr/Grid2Drn.h:1852-2190; Python layer rgrid.pyx:4060-4143): the
oracle's restatement against golden vectors made with the compiled reference (tests/golden/l_golden.npz + make_l_golden.py) and
against the live reference (build container); the HIP path against the same vectors through the C ABI and through rgrid.py (-m gpu)."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["l_const", "l_layers", "l_rough_xz", "l_weno", "l_two_points"]


@pytest.fixture(scope="module")
def lg():
    return np.load(os.path.join(HERE, "golden", "l_golden.npz"))


def _meta(lg, name):
    m = lg[name + "/meta"]
    return dict(nc=(int(m[0]), int(m[1])), dx=float(m[2]), dz=float(m[3]), org=(float(m[4]), float(m[5])), weno=bool(m[6]))


def _same_up_to_ties(i, v, gi, gv):
    """entries equal as multisets per cell (std::sort leaves the order of the entries of ONE cell open)"""
    assert sorted(zip(i.tolist(), v.tolist())) == sorted(zip(gi.tolist(), gv.tolist()))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("rays", [False, True])
def test_oracle_l_matches_golden(oracle, lg, name, dt, rays):
    c = _meta(lg, name)
    key = f"{name}/{np.dtype(dt).name}/{'rays' if rays else 'norays'}"
    r = oracle.solve2d(dt, c["nc"], c["dx"], c["dz"], c["org"], lg[name + "/slowness"], lg[name + "/src"], t0=lg[name + "/t0"],
                       rcv=lg[name + "/rcv"], weno=c["weno"], cell_slowness=True, compute_L=True, return_rays=rays)
    np.testing.assert_array_equal(r["tt_rcv"], lg[key + "/tt_rcv"])
    off = lg[key + "/l_off"]
    assert len(r["l"]) == off.size - 1
    for n, (i, v) in enumerate(r["l"]):
        _same_up_to_ties(i, v, lg[key + "/l_i"][off[n]:off[n + 1]], lg[key + "/l_v"][off[n]:off[n + 1]])
    if rays:
        roff = lg[key + "/r_off"]
        for n, p in enumerate(r["rays"]):
            np.testing.assert_array_equal(p, lg[key + "/r_pts"][roff[n]:roff[n + 1]])
    # a receiver on the source: no entries, traveltime t0; the segment lengths of a ray add up to at least the straight distance
    rcv, src = lg[name + "/rcv"], lg[name + "/src"]
    on_src = int(np.nonzero(np.all(rcv == src[0], axis=1))[0][0])
    assert off[on_src + 1] == off[on_src] and r["tt_rcv"][on_src] == dt(lg[name + "/t0"][0])


def test_the_two_overloads_agree_on_the_entries(lg):
    """the overload without r_data prices the last hop with the entry's value (the sum of the two last segments when they share a
    cell, ttcr/Grid2Drn.h:2163-2181): never less than the other overload's traveltime; the entries are the same"""
    for name in CASES:
        a, b = lg[f"{name}/float64/norays/tt_rcv"], lg[f"{name}/float64/rays/tt_rcv"]
        assert np.all(a >= b)
        np.testing.assert_array_equal(lg[f"{name}/float64/norays/l_v"], lg[f"{name}/float64/rays/l_v"])
        np.testing.assert_array_equal(lg[f"{name}/float64/norays/l_i"], lg[f"{name}/float64/rays/l_i"])


def test_oracle_l_matches_live_reference(oracle):
    if not oracle.have_ref():
        pytest.skip("the compiled reference is not present (GPU box)")
    rng = np.random.default_rng(91)
    n_err = 0
    for trial in range(16):
        dt = np.float32 if trial % 2 else np.float64
        nc = (int(rng.integers(6, 36)), int(rng.integers(6, 36)))
        dx = float(rng.choice([0.5, 1.0, 1.7]))
        dz = dx if trial % 3 else float(rng.choice([0.6, 1.2]))
        s = rng.uniform(0.3, 1.0, nc) if trial % 4 else np.full(nc, 0.7)
        hi = np.array([nc[0] * dx, nc[1] * dz])
        src = rng.uniform(0.1, 0.9, (1, 2)) * hi
        rcv = rng.uniform(0.0, 1.0, (8, 2)) * hi
        for rays in (False, True):
            kw = dict(rcv=rcv, cell_slowness=True, compute_L=True, return_rays=rays, weno=bool(trial % 5 == 0))
            try:
                a = oracle.solve2d(dt, nc, dx, dz, (0, 0), s.ravel(), src, **kw)
            except RuntimeError as e:
                assert "going outside grid" in str(e)
                with pytest.raises(RuntimeError, match="going outside grid"):
                    oracle.ref_solve2d(dt, nc, dx, dz, (0, 0), s.ravel(), src, **kw)
                n_err += 1
                continue
            b = oracle.ref_solve2d(dt, nc, dx, dz, (0, 0), s.ravel(), src, **kw)
            np.testing.assert_array_equal(a["tt_rcv"], b["tt_rcv"])
            for (i1, v1), (i2, v2) in zip(a["l"], b["l"]):
                _same_up_to_ties(i1, v1, i2, v2)
            if rays:
                for p, q in zip(a["rays"], b["rays"]):
                    np.testing.assert_array_equal(p, q)
    assert n_err < 16


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_hip_l_matches_golden(lg, name, dt):
    import ttcr_amd
    from ttcr_amd import _lib

    c = _meta(lg, name)
    ncx, ncz = c["nc"]
    x = c["org"][0] + np.arange(ncx + 1) * c["dx"]
    z = c["org"][1] + np.arange(ncz + 1) * c["dz"]
    g = ttcr_amd.Grid2d(x, z, n_threads=2, cell_slowness=1, method="FSM", weno=int(c["weno"]), dtype=dt)
    g.set_slowness(lg[name + "/slowness"])
    L = _lib.load()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    tx = np.ascontiguousarray(lg[name + "/src"], dtype=dt); t0 = np.ascontiguousarray(lg[name + "/t0"], dtype=dt)
    rcv = lg[name + "/rcv"]
    rx = np.ascontiguousarray(rcv, dtype=dt)
    for rays in (False, True):
        key = f"{name}/{np.dtype(dt).name}/{'rays' if rays else 'norays'}"
        out = np.empty(rx.shape[0], dtype=dt)
        _lib.check(L.ttcr_fsm_raytrace_l(g._h, 1, tx.shape[0], p(tx), p(t0), rx.shape[0], p(rx), p(out), int(rays)))
        np.testing.assert_array_equal(out, lg[key + "/tt_rcv"])
        nrow, nnz = C.c_size_t(0), C.c_size_t(0)
        _lib.check(L.ttcr_fsm_slot_l_size(g._h, 1, C.byref(nrow), C.byref(nnz)))
        off = np.zeros(nrow.value + 1, dtype=np.int64); ii = np.empty(max(nnz.value, 1), dtype=np.int64); vv = np.empty(max(nnz.value, 1), dtype=dt)
        _lib.check(L.ttcr_fsm_get_slot_l(g._h, 1, p(off), p(ii), p(vv)))
        # entry for entry in the reference's order: both sort the same sequence with std::sort and the same comparator
        np.testing.assert_array_equal(off, lg[key + "/l_off"])
        np.testing.assert_array_equal(ii[:nnz.value], lg[key + "/l_i"])
        np.testing.assert_array_equal(vv[:nnz.value], lg[key + "/l_v"])
        if rays:
            nr, npnt = C.c_size_t(0), C.c_size_t(0)
            _lib.check(L.ttcr_fsm_slot_rays_size(g._h, 1, C.byref(nr), C.byref(npnt)))
            roff = np.zeros(nr.value + 1, dtype=np.int64); pts = np.empty((max(npnt.value, 1), 2), dtype=dt)
            _lib.check(L.ttcr_fsm_get_slot_rays(g._h, 1, p(roff), p(pts)))
            np.testing.assert_array_equal(roff, lg[key + "/r_off"])
            np.testing.assert_array_equal(pts[:npnt.value], lg[key + "/r_pts"])
    # the Python layer: (tt, L) and (tt, rays, L) like ttcrpy -- one CSR matrix, receivers x cells
    src = np.column_stack([lg[name + "/t0"], lg[name + "/src"]])
    multi = src.shape[0] > 1
    srows = src if multi else np.repeat(src, rcv.shape[0], axis=0)
    key = f"{name}/{np.dtype(dt).name}/norays"
    tt, Lm = g.raytrace(srows, rcv, compute_L=True, aggregate_src=multi)
    np.testing.assert_array_equal(tt, lg[key + "/tt_rcv"])
    assert Lm.shape == (rcv.shape[0], ncx * ncz)
    goff = lg[key + "/l_off"]
    for n in range(rcv.shape[0]):
        gi, gv = lg[key + "/l_i"][goff[n]:goff[n + 1]], lg[key + "/l_v"][goff[n]:goff[n + 1]]
        dense = np.zeros(ncx * ncz)
        np.add.at(dense, gi, gv.astype(np.float64))
        np.testing.assert_allclose(np.asarray(Lm.getrow(n).todense()).ravel(), dense, rtol=1e-15, atol=0)
    tt2, rays, L2 = g.raytrace(srows, rcv, compute_L=True, return_rays=True, aggregate_src=multi)
    np.testing.assert_array_equal(tt2, lg[f"{name}/{np.dtype(dt).name}/rays/tt_rcv"])
    assert (L2 != Lm).nnz == 0 and len(rays) == rcv.shape[0]
    # refused where ttcrpy refuses: grids with slowness at the nodes (rgrid.pyx:3889-3890)
    gn = ttcr_amd.Grid2d(x, z, cell_slowness=0, method="FSM", dtype=dt)
    with pytest.raises(NotImplementedError):
        gn.raytrace(srows[:2], rcv[:2], compute_L=True)


@pytest.mark.gpu
def test_hip_l_walk_that_leaves_the_grid_raises_like_the_reference():
    import ttcr_amd

    # the receiver make_l_golden.py had to leave out: the reference throws "going outside grid" for it
    rng = np.random.default_rng(4242)
    ncx, ncz, dx, dz = 21, 33, 1.25, 0.75
    x = 8.0 + np.arange(ncx + 1) * dx
    z = -4.0 + np.arange(ncz + 1) * dz
    g = ttcr_amd.Grid2d(x, z, cell_slowness=1, method="FSM", weno=0, dtype=np.float64)
    rng.uniform(0.3, 1.0, 1)   # (not the golden model: any rough model will do, the corner receiver is what matters)
    g.set_slowness(rng.uniform(0.3, 1.0, (ncx, ncz)))
    src = np.array([[20.4, 3.3]])
    rcv = np.array([[x[-1], z[-1]], [15.0, 0.0]])
    try:
        g.raytrace(np.repeat(src, 2, axis=0), rcv, compute_L=True)
    except RuntimeError as e:
        assert "going outside grid" in str(e)


@pytest.mark.gpu
@pytest.mark.parametrize("n_threads,device", [(1, 0), (3, 0), (4, [0, 0, 0])], ids=["1 slot", "3 slots", "4 slots on 3 replicas"])
@pytest.mark.parametrize("rays", [False, True], ids=["l_data", "r_data+l_data"])
def test_hip_l_several_events_in_one_call(oracle, n_threads, device, rays):
    """compute_L with several events goes to the device as ONE call (ttcr_fsm_raytrace_multi_l: batched solves, then the walks):
    the same traveltimes, rays and matrix rows as the restatement gives event by event."""
    import ttcr_amd

    rng = np.random.default_rng(31)
    dt = np.float64
    nc = (60, 44)
    dx, dz = 0.5, 0.25
    nn = (nc[0] + 1, nc[1] + 1)
    X, Z = np.meshgrid((np.arange(nc[0]) + 0.5) * dx, (np.arange(nc[1]) + 0.5) * dz, indexing="ij")
    s = 1.0 / (1.0 + 0.05 * Z) * (1.0 + 0.2 * np.exp(-((X - 12) ** 2 + (Z - 5) ** 2) / 10.0))
    hi = np.array(nc) * np.array([dx, dz])
    n_ev = 4
    ev_src = rng.uniform(1.5, hi - 1.5, (n_ev, 2))
    ev_t0 = rng.uniform(0, 0.5, n_ev).round(3)
    ev_rcv = [rng.uniform(0.6, hi - 0.6, (int(k), 2)) for k in rng.integers(2, 6, n_ev)]
    rows = [(e, k) for e in range(n_ev) for k in range(len(ev_rcv[e]))]
    order = rng.permutation(len(rows))
    src = np.array([[ev_t0[rows[i][0]], *ev_src[rows[i][0]]] for i in order])
    rcv = np.array([ev_rcv[rows[i][0]][rows[i][1]] for i in order])
    axes = [np.arange(nn[0]) * dx, np.arange(nn[1]) * dz]
    g = ttcr_amd.Grid2d(*axes, n_threads=n_threads, cell_slowness=1, method="FSM", tt_from_rp=0, weno=0, dtype=dt, device=device)
    g.set_slowness(s)
    out = g.raytrace(src, rcv, compute_L=True, return_rays=rays)
    tt, Lm = out[0], out[-1]
    assert Lm.shape == (len(rows), nc[0] * nc[1])
    # events in order of first appearance; the reference's Python layer stacks the per-event matrices and takes `tmp[itmp, :]` with
    # itmp = the receiver rows of the events one after the other (rgrid.pyx:4139-4143) -- row q of L is stacked row itmp[q]
    seen = []
    for i in order:
        if rows[i][0] not in seen:
            seen.append(rows[i][0])
    stack, itmp = [], []
    for e in seen:
        sel = [q for q, i in enumerate(order) if rows[i][0] == e]
        o = oracle.solve2d(dt, nc, dx, dz, (0.0, 0.0), s.ravel(), ev_src[e:e + 1], ev_t0[e:e + 1], cell_slowness=True, rcv=rcv[sel],
                           compute_L=True, return_rays=rays)
        np.testing.assert_array_equal(tt[sel], o["tt_rcv"])
        stack += list(o["l"])
        itmp += sel
        if rays:
            for r, q in enumerate(sel):
                np.testing.assert_array_equal(out[1][q], o["rays"][r].astype(np.float64))
    for q in range(len(rows)):
        cells, lens = stack[itmp[q]]
        row = Lm.getrow(q)
        _same_up_to_ties(row.indices.astype(np.int64), row.data.astype(dt), cells.astype(np.int64), lens)
