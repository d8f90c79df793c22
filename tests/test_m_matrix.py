"""The matrix M of `compute_M` (Grid3D::raytrace(..., m_data, threadNo), ttcr/Grid3D.h:743-772 -> Grid3Drn::getRaypath(...,
m_data, ...), ttcr/Grid3Drn.h:1503-1800; and the overload that keeps the rays as well, ttcr/Grid3D.h:646-680 -> Grid3Drn.h:2144-2470,
which ttcrpy calls for compute_M with return_rays and which gives ANOTHER matrix -- keys rm_*): the oracle's restatement against golden vectors made with the compiled reference
(tests/golden/m_golden.npz + make_m_golden.py) and against the live reference (build container); the HIP path against the same
vectors through the C ABI (-m gpu), entry for entry in the reference's push order, signed zeros included."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["m_grad", "m_rough", "m_translate", "m_weno", "m_two_points", "m_close_points"]


@pytest.fixture(scope="module")
def mg():
    return np.load(os.path.join(HERE, "golden", "m_golden.npz"))


def _meta(mg, name):
    m = mg[name + "/meta"]
    return dict(nc=tuple(int(v) for v in m[:3]), dx=float(m[3]), org=tuple(float(v) for v in m[4:7]), translate=bool(m[7]), weno=bool(m[8]))


def _same_entries(j, v, gj, gv):
    assert np.array_equal(j, gj)
    assert np.array_equal(v, gv) and np.array_equal(np.signbit(v), np.signbit(gv))   # (-0.0 and +0.0 are different entries)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_oracle_m_matches_golden(oracle, mg, name, dt):
    c = _meta(mg, name)
    key = f"{name}/{np.dtype(dt).name}"
    r = oracle.solve3d(dt, c["nc"], c["dx"], c["org"], mg[name + "/slowness"], mg[name + "/src"], t0=mg[name + "/t0"], rcv=mg[name + "/rcv"],
                       weno=c["weno"], translate=c["translate"], compute_m=True)
    np.testing.assert_array_equal(r["tt_rcv"], mg[key + "/tt_rcv"])
    off = mg[key + "/m_off"]
    assert len(r["m"]) == off.size - 1
    for n, (j, v) in enumerate(r["m"]):
        _same_entries(j, v, mg[key + "/m_j"][off[n]:off[n + 1]], mg[key + "/m_v"][off[n]:off[n + 1]])
    # the overload with r_data and m_data: its own traveltimes (0 on the source too), terms and rays
    r2 = oracle.solve3d(dt, c["nc"], c["dx"], c["org"], mg[name + "/slowness"], mg[name + "/src"], t0=mg[name + "/t0"], rcv=mg[name + "/rcv"],
                        weno=c["weno"], translate=c["translate"], compute_m=True, return_rays=True)
    np.testing.assert_array_equal(r2["tt_rcv"], mg[key + "/rm_tt_rcv"])
    off2, roff = mg[key + "/rm_off"], mg[key + "/rm_ray_off"]
    for n, (j, v) in enumerate(r2["m"]):
        _same_entries(j, v, mg[key + "/rm_j"][off2[n]:off2[n + 1]], mg[key + "/rm_v"][off2[n]:off2[n + 1]])
        np.testing.assert_array_equal(r2["rays"][n], mg[key + "/rm_ray_pts"][roff[n]:roff[n + 1]])
    assert np.count_nonzero(mg[key + "/rm_v"]) > 3 * np.count_nonzero(mg[key + "/m_v"])   # (every segment carries weight there)
    # what the walk of that overload is: a receiver on the source has no entries and traveltime 0; of a ray's entries only
    # those of the last hop(s) carry weight
    on_src = int(np.nonzero(np.all(mg[name + "/rcv"] == mg[name + "/src"][0], axis=1))[0][0])
    assert off[on_src + 1] == off[on_src] and r["tt_rcv"][on_src] == 0
    for n in range(off.size - 1):
        if n != on_src:
            assert 1 <= np.count_nonzero(mg[key + "/m_v"][off[n]:off[n + 1]]) <= 24 * mg[name + "/src"].shape[0]


def test_oracle_m_matches_live_reference(oracle):
    if not oracle.have_ref():
        pytest.skip("the compiled reference is not present (GPU box)")
    rng = np.random.default_rng(77)
    for dt in (np.float32, np.float64):
        for trial in range(6):
            nc = tuple(int(v) for v in rng.integers(9, 17, 3))
            dx = float(rng.choice([0.5, 1.0, 2.0]))
            nn = tuple(v + 1 for v in nc)
            z = np.arange(nn[2]) * dx
            s = np.repeat(1.0 / (1.0 + 0.05 * z), nn[0] * nn[1]) * rng.uniform(0.9, 1.1, nn[0] * nn[1] * nn[2])
            hi = np.array(nc) * dx
            src = rng.uniform(1.5 * dx, hi - 1.5 * dx, (1, 3))
            if trial >= 2:   # one or two more points of the same source within a cell of the first
                src = np.vstack([src] + [src[0] + rng.uniform(-0.6, 0.6, 3) * dx for _ in range(trial % 2 + 1)])
            rcv = rng.uniform(0.7 * dx, hi - 0.7 * dx, (6, 3))
            kw = dict(rcv=rcv, compute_m=True, weno=bool(trial % 2), return_rays=bool(trial % 3 == 0), t0=rng.uniform(0, 0.5, src.shape[0]).round(3))
            try:
                a = oracle.solve3d(dt, nc, dx, (0, 0, 0), s, src, **kw)
            except RuntimeError as e:   # a walk that leaves the grid: the reference throws as well
                if "did not reach the source" in str(e):
                    continue               # (a walk the reference would never finish: nothing to compare with)
                assert "going outside grid" in str(e)
                with pytest.raises(RuntimeError, match="going outside grid"):
                    oracle.ref_solve3d(dt, nc, dx, (0, 0, 0), s, src, **kw)
                continue
            b = oracle.ref_solve3d(dt, nc, dx, (0, 0, 0), s, src, **kw)
            np.testing.assert_array_equal(a["tt_rcv"], b["tt_rcv"])
            for (j1, v1), (j2, v2) in zip(a["m"], b["m"]):
                _same_entries(j1, v1, j2, v2)
            if kw["return_rays"]:
                for r1, r2 in zip(a["rays"], b["rays"]):
                    np.testing.assert_array_equal(r1, r2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_hip_m_matches_golden(mg, name, dt):
    import ttcr_amd
    from ttcr_amd import _lib

    c = _meta(mg, name)
    key = f"{name}/{np.dtype(dt).name}"
    nn = tuple(v + 1 for v in c["nc"])
    axes = [c["org"][a] + np.arange(nn[a]) * c["dx"] for a in range(3)]
    g = ttcr_amd.Grid3d(*axes, n_threads=2, cell_slowness=0, method="FSM", tt_from_rp=0, weno=int(c["weno"]), dtype=dt,
                        translate_grid=c["translate"])
    g.set_slowness(mg[name + "/slowness"].reshape(nn, order="F"))
    src = np.column_stack([mg[name + "/t0"], mg[name + "/src"]])
    rcv = mg[name + "/rcv"]
    # raw entries through the C ABI, slot 1
    L = _lib.load()
    tx = np.ascontiguousarray(mg[name + "/src"], dtype=dt); t0 = np.ascontiguousarray(mg[name + "/t0"], dtype=dt)
    rx = np.ascontiguousarray(rcv, dtype=dt); out = np.empty(rx.shape[0], dtype=dt)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(L.ttcr_fsm_raytrace_m(g._h, 1, tx.shape[0], p(tx), p(t0), rx.shape[0], p(rx), p(out)))
    np.testing.assert_array_equal(out, mg[key + "/tt_rcv"])
    nrow, nnz = C.c_size_t(0), C.c_size_t(0)
    _lib.check(L.ttcr_fsm_slot_m_size(g._h, 1, C.byref(nrow), C.byref(nnz)))
    off = np.zeros(nrow.value + 1, dtype=np.int64); jj = np.empty(max(nnz.value, 1), dtype=np.int64); vv = np.empty(max(nnz.value, 1), dtype=dt)
    _lib.check(L.ttcr_fsm_get_slot_m(g._h, 1, p(off), p(jj), p(vv)))
    np.testing.assert_array_equal(off, mg[key + "/m_off"])
    _same_entries(jj[:nnz.value], vv[:nnz.value], mg[key + "/m_j"], mg[key + "/m_v"])
    # the Python layer: (tt, M) and (tt, rays, M) like ttcrpy -- one CSR matrix (receivers x nodes) per event, columns ascending
    multi = src.shape[0] > 1   # a source of several points: the rows are the points (aggregate_src), every receiver belongs to it
    srows = src if multi else np.repeat(src, rcv.shape[0], axis=0)
    tt, M = g.raytrace(srows, rcv, compute_M=True, aggregate_src=multi)
    np.testing.assert_array_equal(tt, mg[key + "/tt_rcv"])
    assert len(M) == 1 and M[0].shape == (rcv.shape[0], nn[0] * nn[1] * nn[2])
    goff = mg[key + "/m_off"]
    for n in range(rcv.shape[0]):
        row = M[0].getrow(n)
        gj, gv = mg[key + "/m_j"][goff[n]:goff[n + 1]], mg[key + "/m_v"][goff[n]:goff[n + 1]]
        o = np.argsort(gj, kind="stable")
        np.testing.assert_array_equal(row.indices, gj[o])
        np.testing.assert_array_equal(row.data, gv[o].astype(np.float64))
    # with the rays: the overload with r_data and m_data -- its own matrix, traveltimes and the rays of the r_data overload
    _lib.check(L.ttcr_fsm_raytrace_rm(g._h, 0, tx.shape[0], p(tx), p(t0), rx.shape[0], p(rx), p(out)))
    np.testing.assert_array_equal(out, mg[key + "/rm_tt_rcv"])
    _lib.check(L.ttcr_fsm_slot_m_size(g._h, 0, C.byref(nrow), C.byref(nnz)))
    off = np.zeros(nrow.value + 1, dtype=np.int64); jj = np.empty(max(nnz.value, 1), dtype=np.int64); vv = np.empty(max(nnz.value, 1), dtype=dt)
    _lib.check(L.ttcr_fsm_get_slot_m(g._h, 0, p(off), p(jj), p(vv)))
    np.testing.assert_array_equal(off, mg[key + "/rm_off"])
    _same_entries(jj[:nnz.value], vv[:nnz.value], mg[key + "/rm_j"], mg[key + "/rm_v"])
    tt2, rays, M2 = g.raytrace(srows, rcv, compute_M=True, return_rays=True, aggregate_src=multi)
    np.testing.assert_array_equal(tt2, mg[key + "/rm_tt_rcv"])
    assert len(rays) == rcv.shape[0] and len(M2) == 1
    goff, roff = mg[key + "/rm_off"], mg[key + "/rm_ray_off"]
    for n in range(rcv.shape[0]):
        np.testing.assert_array_equal(rays[n], mg[key + "/rm_ray_pts"][roff[n]:roff[n + 1]].astype(np.float64))
        row = M2[0].getrow(n)
        gj, gv = mg[key + "/rm_j"][goff[n]:goff[n + 1]], mg[key + "/rm_v"][goff[n]:goff[n + 1]]
        o = np.argsort(gj, kind="stable")
        np.testing.assert_array_equal(row.indices, gj[o])
        np.testing.assert_array_equal(row.data, gv[o].astype(np.float64))
    # refused where the reference's Python layer refuses, and where this backend does not follow it
    gc = ttcr_amd.Grid3d(*axes, cell_slowness=1, method="FSM", dtype=dt)
    with pytest.raises(NotImplementedError):
        gc.raytrace(np.repeat(src[:1], 2, axis=0), rcv[:2], compute_M=True)
    with pytest.raises(NotImplementedError):
        g.raytrace(np.repeat(src[:1], 2, axis=0), rcv[:2], compute_L=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("both", [False, True], ids=["m_data", "r_data+m_data"])
def test_hip_m_matches_oracle_medium_grid(oracle, dt, both):
    """Both overloads on a grid of hundreds of patches (97 x 83 x 91 nodes, rough model), a source of three points within a cell
    of each other, 48 receivers spread over the grid: HIP (walk kernel + host assembly) against the restatement, entry for entry."""
    import ttcr_amd
    from ttcr_amd import _lib

    rng = np.random.default_rng(9)
    nn = (97, 83, 91)
    nc = tuple(v - 1 for v in nn)
    dx = 0.25
    s = rng.uniform(0.5, 1.0, nn[0] * nn[1] * nn[2])
    hi = np.array(nc) * dx
    src = np.array([[11.3, 9.1, 13.2]])
    src = np.vstack([src, src[0] + np.array([0.11, -0.07, 0.09]), src[0] + np.array([-0.13, 0.05, 0.02])])
    t0 = np.array([0.25, 0.1, 0.3])
    rcv = rng.uniform(0.6 * dx, hi - 0.6 * dx, (48, 3))
    try:
        o = oracle.solve3d(dt, nc, dx, (0.0, 0.0, 0.0), s, src, t0=t0, rcv=rcv, compute_m=True, return_rays=both)
    except RuntimeError as e:   # (a walk that leaves the grid: choose other receivers -- none with this seed)
        pytest.fail(str(e))
    axes = [np.arange(n) * dx for n in nn]
    g = ttcr_amd.Grid3d(*axes, n_threads=1, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=dt)
    g.set_slowness(s.reshape(nn, order="F"))
    L = _lib.load()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    tx = np.ascontiguousarray(src, dtype=dt); tt0 = np.ascontiguousarray(t0, dtype=dt); rx = np.ascontiguousarray(rcv, dtype=dt)
    out = np.empty(rx.shape[0], dtype=dt)
    _lib.check((L.ttcr_fsm_raytrace_rm if both else L.ttcr_fsm_raytrace_m)(g._h, 0, 3, p(tx), p(tt0), rx.shape[0], p(rx), p(out)))
    np.testing.assert_array_equal(out, o["tt_rcv"])
    nrow, nnz = C.c_size_t(0), C.c_size_t(0)
    _lib.check(L.ttcr_fsm_slot_m_size(g._h, 0, C.byref(nrow), C.byref(nnz)))
    off = np.zeros(nrow.value + 1, dtype=np.int64); jj = np.empty(max(nnz.value, 1), dtype=np.int64); vv = np.empty(max(nnz.value, 1), dtype=dt)
    _lib.check(L.ttcr_fsm_get_slot_m(g._h, 0, p(off), p(jj), p(vv)))
    assert nrow.value == rx.shape[0]
    for n, (j, v) in enumerate(o["m"]):
        _same_entries(jj[off[n]:off[n + 1]], vv[off[n]:off[n + 1]], j, v)
    if both:
        nr, npnt = C.c_size_t(0), C.c_size_t(0)
        _lib.check(L.ttcr_fsm_slot_rays_size(g._h, 0, C.byref(nr), C.byref(npnt)))
        roff = np.zeros(nr.value + 1, dtype=np.int64); pts = np.empty((max(npnt.value, 1), 3), dtype=dt)
        _lib.check(L.ttcr_fsm_get_slot_rays(g._h, 0, p(roff), p(pts)))
        for n, ray in enumerate(o["rays"]):
            np.testing.assert_array_equal(pts[roff[n]:roff[n + 1]], ray)
        # more than one source point at the end of most rays (the end game served several of them)
        assert sum(len(r) >= 3 and any(np.array_equal(r[-2], q.astype(dt)) for q in src) for r in o["rays"]) >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("n_threads,device", [(1, 0), (3, 0), (4, [0, 0, 0]), (5, [0, 0])], ids=["1 slot", "3 slots", "4 slots on 3 replicas", "5 slots on 2 replicas"])
@pytest.mark.parametrize("rays", [False, True], ids=["m_data", "r_data+m_data"])
def test_hip_m_several_events_in_one_call(oracle, n_threads, device, rays):
    """compute_M with several events goes to the device as ONE call (ttcr_fsm_raytrace_multi_m: batched solves, then the walks):
    the same traveltimes, matrices and rays as event by event -- the restatement's, per event."""
    import ttcr_amd

    rng = np.random.default_rng(21)
    dt = np.float32
    nn = (41, 37, 33)
    nc = tuple(v - 1 for v in nn)
    dx = 0.5
    s = rng.uniform(0.5, 1.0, nn[0] * nn[1] * nn[2])
    hi = np.array(nc) * dx
    n_ev = 5
    ev_src = rng.uniform(1.5 * dx, hi - 1.5 * dx, (n_ev, 3))
    ev_t0 = rng.uniform(0, 0.5, n_ev).round(3)
    ev_rcv = [rng.uniform(0.7 * dx, hi - 0.7 * dx, (int(k), 3)) for k in rng.integers(2, 7, n_ev)]
    # ttcrpy-style rows: (event id, t0, x, y, z) per receiver row -- rows of the events interleaved
    rows = [(e, k) for e in range(n_ev) for k in range(len(ev_rcv[e]))]
    order = rng.permutation(len(rows))
    src = np.array([[ev_t0[rows[i][0]], *ev_src[rows[i][0]]] for i in order])
    rcv = np.array([ev_rcv[rows[i][0]][rows[i][1]] for i in order])
    axes = [np.arange(n) * dx for n in nn]
    # (a device list: one replica of the grid per entry -- the events of the call are shared out among the replicas, their
    # matrices, rays and traveltimes put together in call order: MultiGrid::sharded_matrix_call)
    g = ttcr_amd.Grid3d(*axes, n_threads=n_threads, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=dt, device=device)
    g.set_slowness(s.reshape(nn, order="F"))
    out = g.raytrace(src, rcv, compute_M=True, return_rays=rays)
    tt, M = out[0], out[-1]
    assert len(M) == n_ev
    # the events in the order the Python layer finds them (first appearance), each against the restatement
    seen = []
    for i in order:
        if rows[i][0] not in seen:
            seen.append(rows[i][0])
    for m_idx, e in enumerate(seen):
        sel = [q for q, i in enumerate(order) if rows[i][0] == e]
        o = oracle.solve3d(dt, nc, dx, (0.0, 0.0, 0.0), s, ev_src[e:e + 1], t0=ev_t0[e:e + 1], rcv=rcv[sel], compute_m=True, return_rays=rays)
        np.testing.assert_array_equal(tt[sel], o["tt_rcv"])
        assert M[m_idx].shape == (len(sel), nn[0] * nn[1] * nn[2])
        for r, (j, v) in enumerate(o["m"]):
            row = M[m_idx].getrow(r)
            keep = j < nn[0] * nn[1] * nn[2]
            oo = np.argsort(j[keep], kind="stable")
            np.testing.assert_array_equal(row.indices, j[keep][oo])
            np.testing.assert_array_equal(row.data, v[keep][oo].astype(np.float64))
        if rays:
            for r, q in enumerate(sel):
                np.testing.assert_array_equal(out[1][q], o["rays"][r].astype(np.float64))


@pytest.mark.gpu
def test_hip_m_single_slot_call_after_a_paired_batch(oracle, monkeypatch):
    """Round-4 advice: on a grid that keeps its fields in pairs, a batched call pairs its sources by distance and permutes the
    logical -> physical slot map; a later single-slot m_data call (the adapters' per-thread overload, ttcr_fsm_raytrace_m) solves
    its source in the PHYSICAL slot and has to walk THAT field.  4 spread sources in one call, then compute_M slot by slot: every
    slot's traveltimes and matrix are the restatement's."""
    import ttcr_amd
    from ttcr_amd import _lib

    monkeypatch.setenv("TTCR_FSM_PAIR", "1")
    rng = np.random.default_rng(41)
    dt = np.float32
    nn = (37, 35, 33)
    nc = tuple(v - 1 for v in nn)
    dx = 0.5
    s = rng.uniform(0.5, 1.0, nn[0] * nn[1] * nn[2])
    hi = np.array(nc) * dx
    axes = [np.arange(n) * dx for n in nn]
    g = ttcr_amd.Grid3d(*axes, n_threads=4, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=dt)
    g.set_slowness(s.reshape(nn, order="F"))
    # sources 0 and 2 close to each other, 1 and 3 likewise: the distance pairing swaps slots
    srcs = np.array([[2.1, 2.3, 2.2], [14.9, 15.2, 13.8], [2.9, 3.1, 2.6], [15.6, 14.4, 14.7]])
    rcv1 = rng.uniform(0.7 * dx, hi - 0.7 * dx, (5, 3))
    g.raytrace(np.repeat(srcs, rcv1.shape[0], axis=0), np.tile(rcv1, (4, 1)))
    L = _lib.load()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for slot in range(4):
        q = (slot + 1) % 4                                    # another source than the batch left in this slot
        tx = np.ascontiguousarray(srcs[q:q + 1], dtype=dt); tt0 = np.zeros(1, dtype=dt); rx = np.ascontiguousarray(rcv1, dtype=dt)
        out = np.empty(rx.shape[0], dtype=dt)
        _lib.check(L.ttcr_fsm_raytrace_m(g._h, slot, 1, p(tx), p(tt0), rx.shape[0], p(rx), p(out)))
        o = oracle.solve3d(dt, nc, dx, (0.0, 0.0, 0.0), s, srcs[q:q + 1], rcv=rcv1, compute_m=True)
        np.testing.assert_array_equal(out, o["tt_rcv"])
        nrow, nnz = C.c_size_t(0), C.c_size_t(0)
        _lib.check(L.ttcr_fsm_slot_m_size(g._h, slot, C.byref(nrow), C.byref(nnz)))
        off = np.zeros(nrow.value + 1, dtype=np.int64); jj = np.empty(max(nnz.value, 1), dtype=np.int64); vv = np.empty(max(nnz.value, 1), dtype=dt)
        _lib.check(L.ttcr_fsm_get_slot_m(g._h, slot, p(off), p(jj), p(vv)))
        for n, (j, v) in enumerate(o["m"]):
            _same_entries(jj[off[n]:off[n + 1]], vv[off[n]:off[n + 1]], j, v)
        np.testing.assert_array_equal(g.get_grid_traveltimes(slot).flatten("F"), o["tt"])
