"""Host-side logic of the ttcrpy-style wrapper that runs before any device work: argument
checks (rgrid.pyx:194-196, :901-949), source de-duplication / receiver grouping
(rgrid.pyx:926-1028), slowness reshaping (rgrid.pyx:532-569)."""
import numpy as np
import pytest

import ttcr_amd
from ttcr_amd import rgrid


def bare3d(n=5, dtype=np.float64, cell=False):
    """a wrapper object without a device handle (only host-side helpers are exercised)"""
    cls = rgrid.Grid3d_d if dtype == np.float64 else rgrid.Grid3d_f
    g = cls.__new__(cls)
    rgrid._GridBase.__init__(g)
    g._x = g._y = g._z = np.arange(n, dtype=dtype)
    g.cell_slowness = cell
    g._n_threads = 1
    return g


def bare2d(nx=5, nz=7):
    g = rgrid.Grid2d_d.__new__(rgrid.Grid2d_d)
    rgrid._GridBase.__init__(g)
    g._x = np.arange(nx, dtype=np.float64)
    g._z = np.arange(nz, dtype=np.float64)
    g.cell_slowness = False
    g._n_threads = 1
    return g


def test_constructor_checks():
    x = np.arange(5.0)
    with pytest.raises(ValueError, match="cubic"):
        ttcr_amd.Grid3d(x, x * 2, x, method="FSM")
    with pytest.raises(ValueError, match="undefined"):
        ttcr_amd.Grid3d(x, x, x, method="XYZ")
    with pytest.raises(NotImplementedError):
        ttcr_amd.Grid3d(x, x, x, method="SPM")
    with pytest.raises(NotImplementedError, match="SPM"):
        ttcr_amd.Grid2d(x, x, method="FSM", aniso="elliptical")  # anisotropy only exists for the SPM grids
    with pytest.raises(ValueError, match="dtype"):
        ttcr_amd.Grid3d(x, x, x, method="FSM", dtype=np.int32)
    with pytest.raises(NotImplementedError):
        ttcr_amd.Grid2d(x, x)  # reference default method='SPM'
    with pytest.raises(ValueError, match="undefined"):
        ttcr_amd.Grid2d(x, x, method="nope")


def test_shapes_and_indexing():
    g = bare3d(5)
    assert g.shape == (5, 5, 5) and g.nparams == 125
    assert g.get_number_of_nodes() == 125 and g.get_number_of_cells() == 64
    assert g.ind(1, 2, 3) == (1 * 5 + 2) * 5 + 3 and g.indc(1, 2, 3) == (1 * 4 + 2) * 4 + 3
    gc = bare3d(5, cell=True)
    assert gc.shape == (4, 4, 4)
    assert g.is_outside(np.array([[0, 0, 4.5]])) and not g.is_outside(np.array([[0, 0, 4.0]]))


def test_slowness_flattening_is_x_fastest():
    g = bare3d(3)
    a = np.arange(27.0).reshape(3, 3, 3)  # a[i,j,k]
    flat = g._to_flat_F(a, "Slowness")
    # solver order: n = (k*ny + j)*nx + i
    for i in range(3):
        for j in range(3):
            for k in range(3):
                assert flat[(k * 3 + j) * 3 + i] == a[i, j, k]
    np.testing.assert_array_equal(g._to_flat_F(a.ravel(), "Slowness"), flat)  # C-order 1-D input
    with pytest.raises(ValueError, match="wrong size"):
        g._to_flat_F(np.zeros(26), "Slowness")
    with pytest.raises(ValueError, match="wrong shape"):
        g._to_flat_F(np.zeros((9, 3, 1)), "Slowness")
    g2 = bare2d()
    b = np.arange(35.0).reshape(5, 7)
    np.testing.assert_array_equal(g2._to_flat(b, "Slowness"), b.ravel())  # 2-D: C order == z-fastest


def test_source_dedup_keeps_first_occurrence_order():
    g = bare3d(11)
    src = np.array([[5., 5, 5], [1, 1, 1], [5, 5, 5], [1, 1, 1], [3, 3, 3]])
    rcv = np.array([[0., 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [4, 0, 0]])
    vTx, vt0, vRx, iRx = g._split_sources(src, rcv, False)
    assert [tuple(t[0]) for t in vTx] == [(5, 5, 5), (1, 1, 1), (3, 3, 3)]
    assert [list(i) for i in iRx] == [[0, 2], [1, 3], [4]]
    np.testing.assert_array_equal(vRx[1], rcv[[1, 3]])
    assert all(t.shape == (1,) and t[0] == 0 for t in vt0)


def test_source_grouping_matches_the_reference_procedure():
    """_split_sources against the literal steps of rgrid.pyx:926-938 / :1000-1007 (np.unique over the rows, then one
    row mask per unique source) on random row sets: repeated blocks, shuffled rows, -0.0, equal points with
    different origin times"""
    rng = np.random.default_rng(8)
    g = bare3d(11)
    for trial in range(60):
        ns, per = int(rng.integers(1, 7)), int(rng.integers(1, 6))
        pts = rng.integers(0, 4, (ns, 3)).astype(float)
        t0 = rng.integers(0, 2, ns).astype(float)
        src = np.repeat(np.column_stack([t0, pts]), per, axis=0)
        if trial % 3 == 0:
            src = src[rng.permutation(src.shape[0])]
        src[rng.random(src.shape) < 0.05] *= -1.0          # -0.0 entries (coordinates stay inside: only zeros flip
        src = np.where(src < 0, 0.0 * src, src)            # sign; the rest is restored)
        rcv = rng.uniform(0, 10, (src.shape[0], 3))
        _, ind = np.unique(src, axis=0, return_index=True)
        tmp = src[np.sort(ind)]
        if tmp.shape[0] == 1:
            continue                                        # (single-source branch: tested below)
        vTx, vt0, vRx, iRx = g._split_sources(src, rcv, False)
        assert len(vTx) == tmp.shape[0]
        for n in range(tmp.shape[0]):
            rows = np.nonzero(np.sum(tmp[n, 1:] == src[:, 1:], axis=1) == 3)[0]
            np.testing.assert_array_equal(iRx[n], rows)
            np.testing.assert_array_equal(vRx[n], rcv[rows])
            np.testing.assert_array_equal(vTx[n], tmp[n:n + 1, 1:])
            assert vt0[n][0] == tmp[n, 0]


def test_single_source_gets_all_receivers_and_aggregate():
    g = bare3d(11)
    rcv = np.array([[0., 0, 0], [1, 0, 0], [2, 0, 0]])
    vTx, vt0, vRx, iRx = g._split_sources(np.array([[2., 2, 2]]), rcv, False)
    assert len(vTx) == 1 and vRx[0].shape == (3, 3) and list(iRx[0]) == [0, 1, 2]
    # 4 columns: t0, x, y, z; aggregate_src -> one multi-point source
    src4 = np.array([[0.1, 2., 2, 2], [0.2, 3, 3, 3]])
    vTx, vt0, vRx, iRx = g._split_sources(src4, rcv, True)
    assert len(vTx) == 1 and vTx[0].shape == (2, 3)
    np.testing.assert_array_equal(vt0[0], [0.1, 0.2])
    with pytest.raises(ValueError, match="equal size"):
        g._split_sources(src4, rcv, False)


def test_event_id_sources():
    g = bare3d(11)
    src5 = np.array([[7, 0.5, 1., 1, 1], [3, 0.0, 2, 2, 2], [7, 0.5, 1, 1, 1]])
    rcv = np.array([[0., 0, 0], [1, 0, 0], [2, 0, 0]])
    vTx, vt0, vRx, iRx = g._split_sources(src5, rcv, False)
    # events sorted by id: 3 then 7
    assert tuple(vTx[0][0]) == (2, 2, 2) and tuple(vTx[1][0]) == (1, 1, 1)
    assert vt0[1][0] == 0.5 and [list(i) for i in iRx] == [[1], [0, 2]]


def test_outside_points_and_bad_shapes():
    g = bare3d(5)
    rcv = np.array([[0., 0, 0]])
    with pytest.raises(ValueError, match="Source point outside grid"):
        g._split_sources(np.array([[9., 0, 0]]), rcv, False)
    with pytest.raises(ValueError, match="Receiver outside grid"):
        g._split_sources(np.array([[1., 0, 0]]), np.array([[0., -1, 0]]), False)
    with pytest.raises(ValueError, match="nsrc x 3, 4 or 5"):
        g._split_sources(np.zeros((1, 6)), rcv, False)
    g2 = bare2d()
    with pytest.raises(ValueError, match="nsrc x 2 or 3"):
        g2._split_sources(np.zeros((1, 4)), np.zeros((1, 2)), False)
