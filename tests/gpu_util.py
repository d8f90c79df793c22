"""Helpers for the -m gpu parity tests: drive the HIP path through ttcr_amd (C ABI)."""
import numpy as np

import ttcr_amd


def grid_from_case(c, dt, n_threads=1, weno=0, rotated=0):
    nc = c["ncells"]
    o = c["origin"]
    if c["dim"] == 3:
        x = o[0] + np.arange(nc[0] + 1) * c["dx"]
        y = o[1] + np.arange(nc[1] + 1) * c["dx"]
        z = o[2] + np.arange(nc[2] + 1) * c["dx"]
        g = ttcr_amd.Grid3d(x, y, z, n_threads=n_threads, cell_slowness=c["cell_slowness"], method="FSM",
                            tt_from_rp=0, weno=weno, translate_grid=c["translate"], dtype=dt)
        shape = g.shape
        # fixtures hold the solver's flat order (x-fastest); the wrapper takes (nx,ny,nz) arrays
        s = np.asarray(c["slowness"], dtype=np.float64).reshape(shape, order="F")
    else:
        x = o[0] + np.arange(nc[0] + 1) * c["dx"]
        z = o[1] + np.arange(nc[1] + 1) * c["dz"]
        g = ttcr_amd.Grid2d(x, z, n_threads=n_threads, cell_slowness=c["cell_slowness"], method="FSM",
                            tt_from_rp=0, weno=weno, rotated_template=rotated, dtype=dt)
        s = np.asarray(c["slowness"], dtype=np.float64).reshape(g.shape)
    return g, s


def source_array(c):
    """ttcrpy-style source array: (t0, x, y, z) rows, aggregated into ONE event."""
    return np.hstack([np.asarray(c["t0"], dtype=np.float64)[:, None], c["src"]])


def run_case(c, dt, weno=0, rotated=0):
    g, s = grid_from_case(c, dt, weno=weno, rotated=rotated)
    tt_rcv = g.raytrace(source_array(c), c["rcv"], slowness=s, aggregate_src=True)
    if c["dim"] == 3:
        field = g.get_grid_traveltimes().flatten("F")
    else:
        field = g.get_grid_traveltimes().ravel()
    return dict(tt=field, tt_rcv=tt_rcv, niter=g.get_niter(0), niterw=g.get_niterw(0), grid=g)
