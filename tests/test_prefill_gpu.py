"""Re-initialisation off the path of a call (option "prefill", ttcr_amd/csrc/fsm_capi.hip: GridT::solve_batch): a second set of
traveltime fields is filled with max() on a side stream while a solve runs, and the next call that restarts every slot swaps it in.
Replaces the same reference code as the fill it stands in for -- `reinit` of every node, ttcr/Grid3Drnfs.h:92-94, Node3Dn.h:103-105.
Bar: every field, receiver value and iteration count equal to the grid that fills in place, and to the oracle."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt,weno,pair", [(np.float32, 0, 0), (np.float32, 0, 1), (np.float32, 1, 0), (np.float64, 0, 0)])
def test_prefill_swaps_fields_and_keeps_results(oracle, dt, weno, pair, monkeypatch):
    import ttcr_amd

    if pair: monkeypatch.setenv("TTCR_FSM_PAIR", "1")   # (fields of two slots interleaved: the layout of the big batches)

    n, dx, S = 33, 0.4, 4
    x = np.arange(n) * dx
    s = np.random.default_rng(21).uniform(0.3, 1.0, (n, n, n)).astype(dt)
    rcv1 = np.random.default_rng(4).uniform(0.5, x[-1] - 0.5, (5, 3))
    grids = []
    for prefill in (0, 1):
        g = ttcr_amd.Grid3d(x, x, x, n_threads=S, cell_slowness=0, method="FSM", tt_from_rp=0, weno=weno, dtype=dt)
        g.set_option("prefill", prefill)
        g.set_slowness(s)
        grids.append(g)
    rng = np.random.default_rng(8)
    for call in range(6):
        # calls 0, 1, 3, 5: every slot restarted (the swap); 2: more sources than slots (a full batch, then a partial one);
        # 4: a single source in a named slot (in-place fill of that slot, the others keep their fields)
        nsrc = {2: S + 2, 4: 1}.get(call, S)
        srcs = rng.uniform(0.6, x[-1] - 0.6, (nsrc, 3))
        if call == 3: srcs[0] = np.round(srcs[0] / dx) * dx   # a source on a node
        source = np.repeat(srcs, rcv1.shape[0], axis=0)
        rcv = np.tile(rcv1, (nsrc, 1))
        out = []
        for g in grids:
            t = g.raytrace(source, rcv, thread_no=2) if call == 4 else g.raytrace(source, rcv)
            out.append((t, [g.get_grid_traveltimes(q).copy() for q in range(S)], [g.get_niter(q) for q in range(S)]))
        np.testing.assert_array_equal(out[0][0], out[1][0])
        for q in range(S):
            np.testing.assert_array_equal(out[0][1][q], out[1][1][q])
        assert out[0][2] == out[1][2]
        if call in (0, 5):
            want = np.concatenate([oracle.solve3d(dt, (n - 1,) * 3, dx, (0, 0, 0), s.flatten("F"), [p], rcv=rcv1, weno=bool(weno))["tt_rcv"] for p in srcs])
            np.testing.assert_array_equal(out[1][0], want)
    assert grids[1].prefill_swaps() >= 4 and grids[0].prefill_swaps() == 0


def test_prefill_2d_and_device_view_contract():
    import ttcr_amd

    nx, nz, dx = 70, 45, 0.5
    x, z = np.arange(nx) * dx, np.arange(nz) * dx
    s = np.random.default_rng(2).uniform(0.3, 1.0, (nx, nz)).astype(np.float32)
    rcv = np.array([[3.0, 2.0], [20.0, 11.0]])
    g0 = ttcr_amd.Grid2d(x, z, n_threads=2, cell_slowness=0, method="FSM", dtype=np.float32)
    g1 = ttcr_amd.Grid2d(x, z, n_threads=2, cell_slowness=0, method="FSM", dtype=np.float32)
    g1.set_option("prefill", 1)
    for g in (g0, g1): g.set_slowness(s)
    rng = np.random.default_rng(5)
    for call in range(4):
        srcs = rng.uniform(1.0, [x[-1] - 1, z[-1] - 1], (2, 2))
        source, rc = np.repeat(srcs, 2, axis=0), np.tile(rcv, (2, 1))
        t0, t1 = g0.raytrace(source, rc), g1.raytrace(source, rc)
        np.testing.assert_array_equal(t0, t1)
        for q in range(2):
            np.testing.assert_array_equal(g0.get_grid_traveltimes(q), g1.get_grid_traveltimes(q))
    assert g1.prefill_swaps() == 3
