"""Host-side helpers of the ttcrpy grid classes that need no device (ttcr_amd/inversion.py): compute_D, compute_K, data_kernel_straight_rays
(src/ttcrpy/rgrid.pyx:610-756, :1381-1816, :3565-3733, :4259-4470).  Checked against what the operators MEAN -- interpolation of linear
fields is exact, derivative stencils of polynomials, ray lengths that add up to the source-receiver distance -- and against a scalar
restatement of the reference's index conventions (parameters in C order of the grid shape)."""
import numpy as np
import pytest

from ttcr_amd import inversion as inv


def test_interp_matrix_nodes_reproduces_linear_fields_and_reference_ordering():
    rng = np.random.default_rng(3)
    x, y, z = np.arange(7) * 0.5 + 1.0, np.arange(5) * 0.5 - 2.0, np.arange(9) * 0.5
    X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
    f = 2.0 + 0.3 * X - 1.1 * Y + 0.7 * Z
    pts = np.column_stack([rng.uniform(x[0], x[-1], 40), rng.uniform(y[0], y[-1], 40), rng.uniform(z[0], z[-1], 40)])
    pts[0] = [x[-1], y[-1], z[-1]]          # on the upper corner: the last cell
    pts[1] = [x[2], y[1], z[3]]             # on a node
    D = inv.interp_matrix((x, y, z), pts, cell_slowness=False)
    assert D.shape == (40, x.size * y.size * z.size)
    np.testing.assert_allclose(D @ f.ravel(), 2.0 + 0.3 * pts[:, 0] - 1.1 * pts[:, 1] + 0.7 * pts[:, 2], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.asarray(D.sum(axis=1)).ravel(), 1.0, atol=1e-12)
    # the reference's scalar form for one interior point (rgrid.pyx:655-674): ind(i,j,k) = (i*ny + j)*nz + k
    p = pts[5]
    i1, j1, k1 = (int(1e-6 + (p[d] - a[0]) / 0.5) for d, a in enumerate((x, y, z)))
    row = np.zeros(D.shape[1])
    for i in (i1, i1 + 1):
        for j in (j1, j1 + 1):
            for k in (k1, k1 + 1):
                row[(i * y.size + j) * z.size + k] = (1 - abs(p[0] - x[i]) / 0.5) * (1 - abs(p[1] - y[j]) / 0.5) * (1 - abs(p[2] - z[k]) / 0.5)
    np.testing.assert_allclose(D[5].toarray().ravel(), row, atol=1e-14)
    # 2-D
    D2 = inv.interp_matrix((x, z), pts[:, [0, 2]], cell_slowness=False)
    np.testing.assert_allclose(D2 @ (2.0 + 0.3 * X[:, 0, :] + 0.7 * Z[:, 0, :]).ravel(), 2.0 + 0.3 * pts[:, 0] + 0.7 * pts[:, 2], atol=1e-12)


def test_interp_matrix_cells_picks_the_cell():
    x, y, z = np.arange(5) * 1.0, np.arange(4) * 1.0, np.arange(6) * 1.0
    pts = np.array([[0.2, 0.3, 0.4], [3.9, 2.9, 4.9], [2.5, 1.5, 3.5], [4.0, 3.0, 5.0]])
    D = inv.interp_matrix((x, y, z), pts, cell_slowness=True)
    assert D.shape == (4, 4 * 3 * 5) and D.nnz == 4
    want = [(0 * 3 + 0) * 5 + 0, (3 * 3 + 2) * 5 + 4, (2 * 3 + 1) * 5 + 3, (3 * 3 + 2) * 5 + 4]
    assert D.indices.tolist() == want and np.all(D.data == 1.0)


@pytest.mark.parametrize("order", [1, 2])
def test_smoothing_matrices_differentiate_polynomials(order):
    nx, ny, nz = 6, 5, 7
    hx, hy, hz = 0.5, 0.25, 2.0
    X, Y, Z = np.meshgrid(np.arange(nx) * hx, np.arange(ny) * hy, np.arange(nz) * hz, indexing="ij")
    if order == 2:
        f = 1.0 + X ** 2 + 3.0 * Y ** 2 - 2.0 * Z ** 2 + X * Y
        want = (2.0, 6.0, -4.0)          # exact for quadratics, one-sided rows included
    else:
        f = 1.0 + 2.0 * X - 3.0 * Y + 0.5 * Z
        want = (2.0, -3.0, 0.5)
    K = inv.smoothing_matrices((nx, ny, nz), (hx, hy, hz), order=order)
    assert len(K) == 3 and all(k.shape == (nx * ny * nz,) * 2 for k in K)
    for k, w in zip(K, want):
        np.testing.assert_allclose(k @ f.ravel(), w, atol=1e-10)
        assert k.nnz == (3 if order == 2 else 2) * nx * ny * nz
    # rows of the x operator on the first face use the stencil shifted inwards (rgrid.pyx:700-703): columns 0, ny*nz, 2*ny*nz of node (0,0,0)
    if order == 2:
        assert sorted(K[0][0].indices.tolist()) == [0, ny * nz, 2 * ny * nz]
    K2 = inv.smoothing_matrices((nx, nz), (hx, hz), order=order)
    assert len(K2) == 2 and K2[0].shape == (nx * nz, nx * nz)
    with pytest.raises(ValueError):
        inv.smoothing_matrices((nx, nz), (hx, hz), order=3)


def test_straight_ray_kernel_lengths_and_cells():
    rng = np.random.default_rng(8)
    gx, gy, gz = np.arange(6) * 1.0, np.array([0.0, 0.5, 1.5, 3.0, 4.0]), np.arange(8) * 0.5
    n = 30
    Tx = np.column_stack([rng.uniform(gx[0], gx[-1], n), rng.uniform(gy[0], gy[-1], n), rng.uniform(gz[0], gz[-1], n)])
    Rx = np.column_stack([rng.uniform(gx[0], gx[-1], n), rng.uniform(gy[0], gy[-1], n), rng.uniform(gz[0], gz[-1], n)])
    Tx[0], Rx[0] = [0.5, 0.25, 0.25], [0.5, 0.25, 3.25]      # vertical, inside one column of cells
    Tx[1], Rx[1] = [4.5, 3.5, 0.1], [0.5, 0.7, 0.1]          # x decreasing
    L = inv.straight_ray_kernel(Tx, Rx, (gx, gy, gz))
    assert L.shape == (n, 5 * 4 * 7)
    np.testing.assert_allclose(np.asarray(L.sum(axis=1)).ravel(), np.linalg.norm(Rx - Tx, axis=1), rtol=1e-12)
    assert np.all(L.data > 0)
    # the vertical ray crosses the cells (0, 0, 0..6): lengths 0.25, 0.5 x 5, 0.25
    r0 = L[0]
    assert r0.indices.tolist() == list(range(7)) and np.allclose(r0.data, [0.25, 0.5, 0.5, 0.5, 0.5, 0.5, 0.25])
    # a constant slowness gives distance x slowness; a ray's cells are face neighbours in sequence
    np.testing.assert_allclose(L @ np.full(L.shape[1], 0.4), 0.4 * np.linalg.norm(Rx - Tx, axis=1), rtol=1e-12)
    for q in range(n):
        c = L[q].indices
        ijk = np.column_stack(np.unravel_index(c, (5, 4, 7)))
        assert np.all(np.abs(np.diff(ijk, axis=0)).sum(axis=1) >= 1) and np.all(np.abs(np.diff(ijk, axis=0)).max(axis=1) <= 1)
    # 2-D, isotropic and the anisotropic form (x and z components in two blocks)
    L2 = inv.straight_ray_kernel(Tx[:, [0, 2]], Rx[:, [0, 2]], (gx, gz))
    La = inv.straight_ray_kernel(Tx[:, [0, 2]], Rx[:, [0, 2]], (gx, gz), aniso=True)
    assert L2.shape == (n, 35) and La.shape == (n, 70)
    d2 = Rx[:, [0, 2]] - Tx[:, [0, 2]]
    np.testing.assert_allclose(np.asarray(L2.sum(axis=1)).ravel(), np.linalg.norm(d2, axis=1), rtol=1e-12)
    np.testing.assert_allclose(np.asarray(La[:, :35].sum(axis=1)).ravel(), np.abs(d2[:, 0]), rtol=1e-12, atol=1e-14)
    sgn = np.where(d2[:, 0] != 0, np.sign(d2[:, 0]), 1.0)
    np.testing.assert_allclose(np.asarray(La[:, 35:].sum(axis=1)).ravel()[2:], (sgn * d2[:, 1])[2:], rtol=1e-12, atol=1e-14)
    # per cell: the isotropic length is the norm of the two components
    A = La.toarray()
    np.testing.assert_allclose(np.sqrt(A[:, :35] ** 2 + A[:, 35:] ** 2), L2.toarray(), atol=1e-12)


@pytest.mark.gpu
def test_grid_methods_forward_to_the_helpers(tmp_path):
    """Grid3d / Grid2d.compute_D, compute_K, data_kernel_straight_rays, _save_raypaths as a ttcrpy script calls them"""
    import ttcr_amd
    from ttcr_amd import io as tio

    x, y, z = np.arange(9) * 0.5, np.arange(7) * 0.5, np.arange(11) * 0.5
    for cell in (0, 1):
        g = ttcr_amd.Grid3d(x, y, z, cell_slowness=cell, method="FSM")
        pts = np.array([[1.2, 0.7, 3.3], [4.0, 3.0, 5.0]])
        D = g.compute_D(pts)
        assert D.shape == (2, g.nparams)
        Kx, Ky, Kz = g.compute_K()
        assert Kx.shape == (g.nparams, g.nparams) and Kz.nnz == 3 * g.nparams
        with pytest.raises(ValueError):
            g.compute_D(np.array([[99.0, 0.0, 0.0]]))
    s = np.full((8, 6, 10), 0.25)
    g.set_slowness(s)
    src, rcv = np.array([[0.3, 0.3, 0.3]]), np.array([[3.6, 2.7, 4.8]])
    L = ttcr_amd.Grid3d.data_kernel_straight_rays(src, rcv, x, y, z)
    tt_straight = float((L @ s.ravel())[0])
    tt = g.raytrace(src, rcv)
    assert abs(tt_straight - 0.25 * np.linalg.norm(rcv - src)) < 1e-12 and abs(tt[0] - tt_straight) / tt_straight < 0.05   # homogeneous: rays ARE straight
    L2, (xc, yc, zc) = g.data_kernel_straight_rays(src, rcv, x, y, z, centers=True)
    assert (L2 != L).nnz == 0 and xc.size == 8 and zc.size == 10
    tt, rays = g.raytrace(src, rcv, return_rays=True)
    g._save_raypaths(rays, str(tmp_path / "rays.vtp"))
    back = tio.read_vtp_lines(str(tmp_path / "rays.vtp"))
    assert len(back) == 1 and np.allclose(back[0], rays[0], atol=1e-6)
    g2 = ttcr_amd.Grid2d(x, z, cell_slowness=0, method="FSM")
    D2 = g2.compute_D(np.array([[1.1, 2.2]]))
    Kx2, Kz2 = g2.compute_K(order=2)
    assert D2.shape == (1, g2.nparams) and abs(D2.sum() - 1.0) < 1e-12 and Kx2.shape == (g2.nparams,) * 2
    La = ttcr_amd.Grid2d.data_kernel_straight_rays(np.array([[0.1, 0.2]]), np.array([[3.3, 4.4]]), x, z, aniso=True)
    assert La.shape == (1, 2 * 8 * 10)
