"""Host-side helpers of the ttcrpy grid classes that inversion codes call next to `raytrace`: interpolation weights of velocity data points
(`compute_D`), smoothing operators (`compute_K`), straight-ray data kernels (`data_kernel_straight_rays`).  They build scipy sparse
matrices from the grid geometry alone -- no solver, no device -- and follow the parameter order of the reference's wrapper
(src/ttcrpy/rgrid.pyx: parameters in C order of `shape`, i.e. z fastest).  Restated with array arithmetic, not ported loop by loop:

  interp_matrix        rgrid.pyx:610-677 (3-D), :3565-3627 (2-D)     one weight 1 per point (cell slowness) or the 2^d multilinear weights
  smoothing_matrices   rgrid.pyx:679-756 (3-D, second differences), :3630-3733 (2-D, order 1 or 2)   one-sided stencils on the faces
  straight_ray_kernel  rgrid.pyx:1381-1720 (3-D), :4259-4470 (2-D, optional elliptical-anisotropy form)   segment lengths per cell
"""
import numpy as np


def _sp():
    import scipy.sparse as sp
    return sp


def interp_matrix(axes, coord, cell_slowness):
    """D (npts x nparams) with D @ params = the parameter field at `coord`.  axes: node coordinates per dimension (uniform spacing).
    Cell slowness: the cell that holds the point.  Node slowness: multilinear weights of the 2^d nodes around it; the cell index is
    int(1e-6 + (c - x0) / dx) as in the reference, clamped so that points on the upper faces take the last cell (the reference indexes
    one node past the end there)."""
    sp = _sp()
    coord = np.asarray(coord, dtype=np.float64)
    nd = len(axes)
    if coord.ndim != 2 or coord.shape[1] != nd:
        raise ValueError('coord should be npts by {0:d}'.format(nd))
    npts = coord.shape[0]
    nn = [a.size for a in axes]
    h = [float(a[1] - a[0]) for a in axes]
    if cell_slowness:
        idx = [np.minimum(((coord[:, d] - axes[d][0]) / h[d]).astype(np.int64), nn[d] - 2) for d in range(nd)]
        col = idx[0]
        for d in range(1, nd):
            col = col * (nn[d] - 1) + idx[d]
        ncell = int(np.prod([n - 1 for n in nn]))
        return sp.csr_matrix((np.ones(npts), (np.arange(npts), col)), shape=(npts, ncell))
    lo = [np.clip((1.e-6 + (coord[:, d] - axes[d][0]) / h[d]).astype(np.int64), 0, nn[d] - 2) for d in range(nd)]
    rows, cols, vals = [], [], []
    for corner in range(1 << nd):
        col = np.zeros(npts, dtype=np.int64)
        w = np.ones(npts)
        for d in range(nd):
            i = lo[d] + ((corner >> (nd - 1 - d)) & 1)          # (i1, i2) outermost along x, like the reference's loops
            col = col * nn[d] + i
            w = w * (1. - np.abs(coord[:, d] - axes[d][i]) / h[d])
        rows.append(np.arange(npts)); cols.append(col); vals.append(w)
    rows = np.stack(rows, axis=1).ravel(); cols = np.stack(cols, axis=1).ravel(); vals = np.stack(vals, axis=1).ravel()
    return sp.csr_matrix((vals, (rows, cols)), shape=(npts, int(np.prod(nn))))


def _stencil_1d(n, h, order):
    """n x n operator along one axis: order 2 -- (1, -2, 1) / h^2 centred, the same stencil shifted inwards on the two end rows;
    order 1 -- (-1/2, 1/2) / h centred, (-1, 1) / h one-sided on the end rows."""
    sp = _sp()
    r = np.arange(n)
    if order == 2:
        if n < 3:
            raise ValueError('second-order smoothing needs at least 3 parameters along every axis')
        c = np.clip(r, 1, n - 2)
        rows = np.repeat(r, 3)
        cols = np.stack((c - 1, c, c + 1), axis=1).ravel()
        vals = np.tile(np.array([1., -2., 1.]) / (h * h), n)
    elif order == 1:
        if n < 2:
            raise ValueError('first-order smoothing needs at least 2 parameters along every axis')
        a = np.maximum(r - 1, 0); b = np.minimum(r + 1, n - 1)
        w = np.where((r == 0) | (r == n - 1), 1.0, 0.5) / h
        rows = np.repeat(r, 2)
        cols = np.stack((a, b), axis=1).ravel()
        vals = np.stack((-w, w), axis=1).ravel()
    else:
        raise ValueError('order value not valid (1 or 2 accepted)')
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, n))


def smoothing_matrices(shape, spacings, order=2):
    """One derivative operator per axis acting on the parameters in C order of `shape`: K_d = I x ... x S_d x ... x I."""
    sp = _sp()
    out = []
    for d in range(len(shape)):
        m = None
        for e in range(len(shape)):
            f = _stencil_1d(shape[e], spacings[e], order) if e == d else sp.identity(shape[e], format='csr')
            m = f if m is None else sp.kron(m, f, format='csr')
        out.append(m.tocsr())
    return tuple(out)


def straight_ray_kernel(Tx, Rx, axes, aniso=False):
    """L (nrays x ncells) with L @ slowness = traveltimes along the straight segments Tx[n] -> Rx[n] through the cells of the grid `axes`
    (node coordinates per dimension, any spacing).  Cells in C order ((ix * ncy + iy) * ncz + iz).  aniso (2-D only): L is nrays x 2 ncells,
    the x components of the segments in the first block and the z components in the second (signed, with the ray oriented towards
    increasing x -- increasing z for a vertical ray -- as the reference walks it)."""
    sp = _sp()
    Tx = np.asarray(Tx, dtype=np.float64); Rx = np.asarray(Rx, dtype=np.float64)
    nd = len(axes)
    if Tx.shape != Rx.shape or Tx.ndim != 2 or Tx.shape[1] != nd:
        raise ValueError('Tx and Rx should be nrays by {0:d}, one row per source-receiver pair'.format(nd))
    if aniso and nd != 2:
        raise ValueError('aniso: 2-D grids only')
    axes = [np.asarray(a, dtype=np.float64) for a in axes]
    nc = [a.size - 1 for a in axes]
    ncell = int(np.prod(nc))
    data, indices, indptr = [], [], [0]
    for n in range(Tx.shape[0]):
        p1, p2 = Tx[n], Rx[n]
        if p1[0] > p2[0] or (p1[0] == p2[0] and p1[-1] > p2[-1]):
            p1, p2 = p2, p1
        v = p2 - p1
        d = float(np.sqrt(np.sum(v * v)))
        if d == 0.0:
            indptr.append(len(data))
            continue
        ts = [np.array([0.0, 1.0])]
        for a in range(nd):
            if v[a] != 0.0:
                t = (axes[a] - p1[a]) / v[a]
                ts.append(t[(t > 0.0) & (t < 1.0)])
        t = np.unique(np.concatenate(ts))
        dt = np.diff(t)
        keep = dt > 1.e-14
        tm = 0.5 * (t[:-1] + t[1:])[keep]
        dt = dt[keep]
        cell = np.zeros(tm.size, dtype=np.int64)
        for a in range(nd):
            ia = np.clip(np.searchsorted(axes[a], p1[a] + tm * v[a], side='right') - 1, 0, nc[a] - 1)
            cell = cell * nc[a] + ia
        if not aniso:
            data.extend((dt * d).tolist()); indices.extend(cell.tolist())
        else:
            for c, s in zip(cell.tolist(), dt.tolist()):
                indices.append(c); data.append(s * v[0])
                indices.append(c + ncell); data.append(s * v[1])
        indptr.append(len(data))
    return sp.csr_matrix((np.array(data), np.array(indices, dtype=np.int64), np.array(indptr, dtype=np.int64)),
                         shape=(Tx.shape[0], 2 * ncell if aniso else ncell))
