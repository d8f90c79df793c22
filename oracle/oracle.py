"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of the CPU checker(s):

* ``liboracle.so``          this repo's plain-C restatement of the reference FSM
                            solver (oracle/fsm_oracle.c); travels to the GPU box prebuilt.
* ``_ref/libttcr_ref.so``   the unmodified reference headers compiled where they lie under
                            /root/reference (build container only, see oracle/Makefile).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / the timed CPU baseline.  The product
(``ttcr_amd``) never imports it.

All flat 3-D arrays are x-fastest (n = (k*nny + j)*nnx + i, ttcr/Grid3Drn.h:2823); flat 2-D
arrays are z-fastest (n = i*nnz + j, ttcr/Grid2Drn.h:720).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(with_ref=True):
    """Compile liboracle.so (and _ref/libttcr_ref.so when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if with_ref and os.path.isdir("/root/reference/ttcr"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libttcr_ref.so"))


class _G3(C.Structure):
    pass


def _g3(ct):
    class G(C.Structure):
        _fields_ = [("nnx", C.c_size_t), ("nny", C.c_size_t), ("nnz", C.c_size_t)] + [
            (n, ct) for n in ("dx", "xmin", "ymin", "zmin", "xmax", "ymax", "zmax", "ox", "oy", "oz")
        ]

    return G


def _g2(ct):
    class G(C.Structure):
        _fields_ = [("nnx", C.c_size_t), ("nnz", C.c_size_t)] + [
            (n, ct) for n in ("dx", "dz", "xmin", "zmin", "xmax", "zmax")
        ]

    return G


_TYPES = {
    np.dtype(np.float32): ("f32", C.c_float, _g3(C.c_float), _g2(C.c_float)),
    np.dtype(np.float64): ("f64", C.c_double, _g3(C.c_double), _g2(C.c_double)),
}


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("TTCR_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # (override: the ASan build)
        if not os.environ.get("TTCR_ORACLE_LIB"):
            try:   # (make is a no-op when the library is newer than its sources: never check against a stale restatement)
                build(with_ref=False)
            except Exception:
                if not os.path.exists(path):
                    raise
        _LIB = C.CDLL(path)
        for sfx, ct, _, _ in _TYPES.values():
            getattr(_LIB, "fsm_interp3d_" + sfx).restype = ct
            getattr(_LIB, "fsm_interp2d_" + sfx).restype = ct
            getattr(_LIB, "fsm_compute_slowness3d_" + sfx).restype = ct
            getattr(_LIB, "fsm_compute_slowness2d_" + sfx).restype = ct
    return _LIB


def ref():
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libttcr_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref/libttcr_ref.so not built (needs /root/reference)")
        _REF = C.CDLL(path)
        _REF.ref_last_error.restype = C.c_char_p
    return _REF


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _prep_pts(dt, pts, ncol):
    pts = np.ascontiguousarray(np.asarray(pts, dtype=dt).reshape(-1, ncol))
    return pts


# --------------------------------------------------------------------------- 3D


def solve3d(dtype, ncells, dx, origin, slowness, src, t0=None, eps=1e-5, maxit=50,
            cell_slowness=False, translate=False, rcv=None, weno=False, tt_from_rp=False, interp_vel=False,
            return_rays=False, compute_m=False):
    """Restatement of Grid3Drnfs / Grid3Drcfs ::raytrace (tt_from_rp=False; weno selects the
    two-stage first-order + WENO3 driver).

    ncells = (ncx, ncy, ncz) CELL counts, as in the reference constructors.
    Returns dict(tt=flat node field, niter, change=per-iteration L1 change, tt_rcv).
    """
    dt = np.dtype(dtype)
    sfx, ct, G3, _ = _TYPES[dt]
    L = lib()
    ncx, ncy, ncz = (int(v) for v in ncells)
    g = G3()
    getattr(L, "fsm_grid3d_init_" + sfx)(C.byref(g), C.c_uint32(ncx), C.c_uint32(ncy), C.c_uint32(ncz),
                                         ct(dx), ct(origin[0]), ct(origin[1]), ct(origin[2]),
                                         C.c_int(int(translate)))
    nn = (ncx + 1) * (ncy + 1) * (ncz + 1)
    s = np.ascontiguousarray(np.asarray(slowness, dtype=dt).ravel())
    if cell_slowness:
        assert s.size == ncx * ncy * ncz
        sn = np.empty(nn, dtype=dt)
        getattr(L, "fsm_cells_to_nodes3d_" + sfx)(C.c_size_t(ncx), C.c_size_t(ncy), C.c_size_t(ncz),
                                                  _p(s), _p(sn))
    else:
        assert s.size == nn
        sn = s
    src = _prep_pts(dt, src, 3).copy()
    nsrc = src.shape[0]
    t0 = np.zeros(nsrc, dtype=dt) if t0 is None else np.ascontiguousarray(np.asarray(t0, dtype=dt))
    if translate:  # Grid3D::raytrace subtracts the stored origin (ttcr/Grid3D.h:478-485)
        src -= np.array([g.ox, g.oy, g.oz], dtype=dt)
    if getattr(L, "fsm_outside3d_" + sfx)(C.byref(g), C.c_int(nsrc), _p(src)):
        raise RuntimeError("Error: Point outside grid.")
    T = np.empty(nn, dtype=dt)
    hist = np.zeros(2 * maxit, dtype=dt)
    nw = C.c_int(0)
    niter = getattr(L, "fsm_solve3d_" + sfx)(C.byref(g), _p(sn), C.c_int(nsrc), _p(src), _p(t0), ct(eps),
                                             C.c_int(maxit), C.c_int(int(weno)), _p(T), _p(hist), C.byref(nw))
    out = dict(tt=T, niter=int(niter), niterw=int(nw.value), change=hist[:niter].copy(),
               changew=hist[maxit:maxit + nw.value].copy(), node_slowness=sn)
    if rcv is not None:
        r = _prep_pts(dt, rcv, 3).copy()
        if translate:
            r -= np.array([g.ox, g.oy, g.oz], dtype=dt)
        if getattr(L, "fsm_outside3d_" + sfx)(C.byref(g), C.c_int(r.shape[0]), _p(r)):   # Grid3Drnfs::raytrace: checkPts(Rx)
            raise RuntimeError("Error: Point outside grid.")
        if return_rays and compute_m:
            # Grid3D::raytrace(Tx,t0,Rx,tt,r_data,m_data,threadNo) (ttcr/Grid3D.h:646-680): the rays and, per receiver, the
            # entries of M of the overload that keeps both (its segments carry their lengths)
            frm = getattr(L, "fsm_raypath3d_rm_" + sfx)
            vals = np.empty(r.shape[0], dtype=dt)
            rays, ms = [], []
            cap = 4 * (ncx + ncy + ncz) + 64
            for n, pnt in enumerate(r):
                pp = np.ascontiguousarray(pnt, dtype=dt)
                v = ct(0)
                while True:
                    buf = np.empty((cap, 3), dtype=dt)
                    mj = np.empty(16 * cap, dtype=np.int64)
                    mv = np.empty(16 * cap, dtype=dt)
                    npts, nm = C.c_long(0), C.c_long(0)
                    rc = frm(C.byref(g), _p(sn), _p(T), C.c_int(nsrc), _p(src), _p(t0), _p(pp), C.c_int(int(interp_vel)),
                             C.c_long(1000000), C.byref(v), _p(buf), C.c_long(cap), C.byref(npts), _p(mj), _p(mv), C.c_long(16 * cap),
                             C.byref(nm))
                    if rc != 3:
                        break
                    cap *= 4
                if rc == 1:
                    raise RuntimeError("Error while computing raypaths: going outside grid")
                if rc == 2:
                    raise RuntimeError("raypath did not reach the source")
                vals[n] = v.value
                ray = buf[:npts.value].copy()
                if translate:
                    ray += np.array([g.ox, g.oy, g.oz], dtype=dt)
                rays.append(ray)
                ms.append((mj[:nm.value].copy(), mv[:nm.value].copy()))
            out["tt_rcv"] = vals
            out["rays"] = rays
            out["m"] = ms
        elif return_rays:
            # Grid3D::raytrace(Tx,t0,Rx,tt,r_data,threadNo) (ttcr/Grid3D.h:546-586): getRaypath with tt for
            # every receiver; rays are shifted back by the origin of a translated grid (:579-584)
            frp = getattr(L, "fsm_raypath3d_" + sfx)
            vals = np.empty(r.shape[0], dtype=dt)
            rays = []
            cap = 4 * (ncx + ncy + ncz) + 64
            for n, pnt in enumerate(r):
                pp = np.ascontiguousarray(pnt, dtype=dt)
                v = ct(0)
                while True:
                    buf = np.empty((cap, 3), dtype=dt)
                    npts = C.c_long(0)
                    rc = frp(C.byref(g), _p(sn), _p(T), C.c_int(nsrc), _p(src), _p(t0), _p(pp), C.c_int(int(interp_vel)),
                             C.c_long(1000000), C.byref(v), _p(buf), C.c_long(cap), C.byref(npts))
                    if rc != 3:
                        break
                    cap *= 4
                if rc == 1:
                    raise RuntimeError("Error while computing raypaths: going outside grid")
                if rc == 2:
                    raise RuntimeError("raypath did not reach the source")
                vals[n] = v.value
                ray = buf[:npts.value].copy()
                if translate:
                    ray += np.array([g.ox, g.oy, g.oz], dtype=dt)
                rays.append(ray)
            out["tt_rcv"] = vals
            out["rays"] = rays
        elif compute_m:
            # Grid3D::raytrace(Tx,t0,Rx,tt,m_data,threadNo) (ttcr/Grid3D.h:743-772): per receiver the (node, value) entries of
            # the matrix M in the order the reference pushes them, and the traveltime of that overload
            fm = getattr(L, "fsm_raypath3d_m_" + sfx)
            vals = np.empty(r.shape[0], dtype=dt)
            ms = []
            cap = 64 * (ncx + ncy + ncz) + 256
            for n, pnt in enumerate(r):
                pp = np.ascontiguousarray(pnt, dtype=dt)
                v = ct(0)
                while True:
                    mj = np.empty(cap, dtype=np.int64)
                    mv = np.empty(cap, dtype=dt)
                    nm = C.c_long(0)
                    rc = fm(C.byref(g), _p(sn), _p(T), C.c_int(nsrc), _p(src), _p(t0), _p(pp), C.c_int(int(interp_vel)),
                            C.c_long(1000000), C.byref(v), _p(mj), _p(mv), C.c_long(cap), C.byref(nm))
                    if rc != 3:
                        break
                    cap *= 4
                if rc == 1:
                    raise RuntimeError("Error while computing raypaths: going outside grid")
                if rc == 2:
                    raise RuntimeError("raypath did not reach the source")
                vals[n] = v.value
                ms.append((mj[:nm.value].copy(), mv[:nm.value].copy()))
            out["tt_rcv"] = vals
            out["m"] = ms
        elif tt_from_rp:
            # Grid3D::raytrace with tt_from_rp (ttcr/Grid3D.h:493-496): traveltime integrated along the ray
            frp = getattr(L, "fsm_tt_from_raypath3d_" + sfx)
            vals = np.empty(r.shape[0], dtype=dt)
            for n, pnt in enumerate(r):
                pp = np.ascontiguousarray(pnt, dtype=dt)
                v = ct(0)
                rc = frp(C.byref(g), _p(sn), _p(T), C.c_int(nsrc), _p(src), _p(t0), _p(pp), C.c_int(int(interp_vel)),
                         C.c_long(1000000), C.byref(v))
                if rc == 1:
                    raise RuntimeError("Error while computing raypaths: going outside grid")
                if rc == 2:
                    raise RuntimeError("raypath did not reach the source")
                vals[n] = v.value
            out["tt_rcv"] = vals
        else:
            f = getattr(L, "fsm_interp3d_" + sfx)
            out["tt_rcv"] = np.array([f(C.byref(g), _p(T), ct(p[0]), ct(p[1]), ct(p[2])) for p in r], dtype=dt)
    return out


def cells_to_nodes3d(dtype, ncells, sc):
    dt = np.dtype(dtype)
    sfx = _TYPES[dt][0]
    ncx, ncy, ncz = (int(v) for v in ncells)
    sc = np.ascontiguousarray(np.asarray(sc, dtype=dt).ravel())
    sn = np.empty((ncx + 1) * (ncy + 1) * (ncz + 1), dtype=dt)
    getattr(lib(), "fsm_cells_to_nodes3d_" + sfx)(C.c_size_t(ncx), C.c_size_t(ncy), C.c_size_t(ncz), _p(sc), _p(sn))
    return sn


def ref_solve3d(dtype, ncells, dx, origin, slowness, src, t0=None, eps=1e-5, maxit=50,
                cell_slowness=False, translate=False, rcv=None, weno=False, tt_from_rp=False, interp_vel=False,
                return_rays=False, compute_m=False):
    """The compiled, unmodified reference (build container only)."""
    dt = np.dtype(dtype)
    sfx, ct = _TYPES[dt][:2]
    R = ref()
    ncx, ncy, ncz = (int(v) for v in ncells)
    nn = (ncx + 1) * (ncy + 1) * (ncz + 1)
    s = np.ascontiguousarray(np.asarray(slowness, dtype=dt).ravel())
    src = _prep_pts(dt, src, 3)
    nsrc = src.shape[0]
    t0 = np.zeros(nsrc, dtype=dt) if t0 is None else np.ascontiguousarray(np.asarray(t0, dtype=dt))
    r = _prep_pts(dt, rcv if rcv is not None else np.zeros((0, 3)), 3)
    tt_rcv = np.empty(r.shape[0], dtype=dt)
    T = np.empty(nn, dtype=dt)
    niter = (C.c_int * 2)()
    cap = (4 * (ncx + ncy + ncz) + 64) * max(r.shape[0], 1)
    while True:
        if return_rays:
            rbuf = np.empty((cap, 3), dtype=np.float64)
            roff = np.zeros(r.shape[0] + 1, dtype=np.int64)
            R.ref_set_rays(_p(rbuf), C.c_long(cap), _p(roff))
        if compute_m:
            mcap = 64 * cap
            mjb = np.empty(mcap, dtype=np.int64)
            mvb = np.empty(mcap, dtype=np.float64)
            moff = np.zeros(r.shape[0] + 1, dtype=np.int64)
            R.ref_set_m(_p(mjb), _p(mvb), C.c_long(mcap), _p(moff))
        try:
            rc = getattr(R, "ref_fsm3d_" + sfx)(C.c_int(int(cell_slowness)), C.c_uint32(ncx), C.c_uint32(ncy),
                                                C.c_uint32(ncz), ct(dx), ct(origin[0]), ct(origin[1]), ct(origin[2]),
                                                ct(eps), C.c_int(maxit), C.c_int(int(weno)), C.c_int(int(translate)),
                                                C.c_int(int(tt_from_rp)), C.c_int(int(interp_vel)),
                                                _p(s), C.c_int(nsrc), _p(src), _p(t0), C.c_int(r.shape[0]), _p(r),
                                                _p(tt_rcv), _p(T), niter)
        finally:
            if return_rays:
                R.ref_set_rays(None, C.c_long(0), None)
            if compute_m:
                R.ref_set_m(None, None, C.c_long(0), None)
        if rc != 0:
            raise RuntimeError(R.ref_last_error().decode())
        if compute_m and moff[-1] > mcap:
            cap = int(moff[-1]) // 64 + 1
            continue
        if not return_rays or roff[-1] <= cap:
            break
        cap = int(roff[-1])
    out = dict(tt=T, niter=int(niter[0]), niterw=int(niter[1]), tt_rcv=tt_rcv)
    if return_rays:
        out["rays"] = [rbuf[roff[n]:roff[n + 1]].astype(dt) for n in range(r.shape[0])]
    if compute_m:
        out["m"] = [(mjb[moff[n]:moff[n + 1]].copy(), mvb[moff[n]:moff[n + 1]].astype(dt)) for n in range(r.shape[0])]
    return out


def compute_slowness3d(dtype, ncells, dx, origin, slowness, pts, cell_slowness=False, translate=False, interp_vel=False,
                       use_ref=False):
    """Grid3D::computeSlowness(pt) at every row of pts (original coordinates), as get_s0 calls it
    (src/ttcrpy/rgrid.pyx:824); use_ref: the compiled reference instead of the restatement."""
    dt = np.dtype(dtype)
    sfx, ct, G3, _ = _TYPES[dt]
    ncx, ncy, ncz = (int(v) for v in ncells)
    s = np.ascontiguousarray(np.asarray(slowness, dtype=dt).ravel())
    p = _prep_pts(dt, pts, 3).copy()
    out = np.empty(p.shape[0], dtype=dt)
    if use_ref:
        rc = getattr(ref(), "ref_compute_slowness3d_" + sfx)(C.c_int(int(cell_slowness)), C.c_uint32(ncx), C.c_uint32(ncy),
                                                             C.c_uint32(ncz), ct(dx), ct(origin[0]), ct(origin[1]), ct(origin[2]),
                                                             C.c_int(int(translate)), C.c_int(int(interp_vel)), _p(s),
                                                             C.c_int(p.shape[0]), _p(p), _p(out))
        if rc != 0:
            raise RuntimeError(ref().ref_last_error().decode())
        return out
    L = lib()
    g = G3()
    getattr(L, "fsm_grid3d_init_" + sfx)(C.byref(g), C.c_uint32(ncx), C.c_uint32(ncy), C.c_uint32(ncz), ct(dx), ct(origin[0]),
                                         ct(origin[1]), ct(origin[2]), C.c_int(int(translate)))
    sn = cells_to_nodes3d(dt, ncells, s) if cell_slowness else s
    if translate:
        p -= np.array([g.ox, g.oy, g.oz], dtype=dt)
    f = getattr(L, "fsm_compute_slowness3d_" + sfx)
    for n, q in enumerate(p):
        out[n] = f(C.byref(g), _p(sn), ct(q[0]), ct(q[1]), ct(q[2]), C.c_int(int(interp_vel)))
    return out


def compute_slowness2d(dtype, ncells, dx, dz, origin, slowness, pts, cell_slowness=False, use_ref=False):
    """Grid2Drn::computeSlowness(pt), ttcr/Grid2Drn.h:262-330 (node and cell FSM grids alike)."""
    dt = np.dtype(dtype)
    sfx, ct, _, G2 = _TYPES[dt]
    ncx, ncz = (int(v) for v in ncells)
    s = np.ascontiguousarray(np.asarray(slowness, dtype=dt).ravel())
    p = _prep_pts(dt, pts, 2)
    out = np.empty(p.shape[0], dtype=dt)
    if use_ref:
        rc = getattr(ref(), "ref_compute_slowness2d_" + sfx)(C.c_int(int(cell_slowness)), C.c_uint32(ncx), C.c_uint32(ncz), ct(dx),
                                                             ct(dz), ct(origin[0]), ct(origin[1]), _p(s), C.c_int(p.shape[0]),
                                                             _p(p), _p(out))
        if rc != 0:
            raise RuntimeError(ref().ref_last_error().decode())
        return out
    L = lib()
    g = G2()
    getattr(L, "fsm_grid2d_init_" + sfx)(C.byref(g), C.c_uint32(ncx), C.c_uint32(ncz), ct(dx), ct(dz), ct(origin[0]), ct(origin[1]))
    if cell_slowness:
        sn = np.empty((ncx + 1) * (ncz + 1), dtype=dt)
        getattr(L, "fsm_cells_to_nodes2d_" + sfx)(C.c_size_t(ncx), C.c_size_t(ncz), _p(s), _p(sn))
    else:
        sn = s
    f = getattr(L, "fsm_compute_slowness2d_" + sfx)
    for n, q in enumerate(p):
        out[n] = f(C.byref(g), _p(sn), ct(q[0]), ct(q[1]))
    return out


# --------------------------------------------------------------------------- 2D


def solve2d(dtype, ncells, dx, dz, origin, slowness, src, t0=None, eps=1e-5, maxit=50,
            cell_slowness=False, rcv=None, weno=False, rotated=False, tt_from_rp=False, return_rays=False, compute_L=False):
    """compute_L: the overloads with l_data (ttcr/Grid2D.h:583-640): out["l"] = per receiver (cells, lengths) sorted by cell
    (stable -- the reference's std::sort may order the entries of ONE cell differently), traveltimes of that overload.
    Restatement of Grid2Drnfs / Grid2Drcfs ::raytrace (rotated: sweep45 after every sweep, ttcr/Grid2Drnfs.h:277-286;
    tt_from_rp / return_rays: Grid2Drn::getTraveltimeFromRaypath / getRaypath, ttcr/Grid2Drn.h:1478-1850)."""
    dt = np.dtype(dtype)
    sfx, ct, _, G2 = _TYPES[dt]
    L = lib()
    ncx, ncz = (int(v) for v in ncells)
    g = G2()
    getattr(L, "fsm_grid2d_init_" + sfx)(C.byref(g), C.c_uint32(ncx), C.c_uint32(ncz), ct(dx), ct(dz),
                                         ct(origin[0]), ct(origin[1]))
    nn = (ncx + 1) * (ncz + 1)
    s = np.ascontiguousarray(np.asarray(slowness, dtype=dt).ravel())
    if cell_slowness:
        assert s.size == ncx * ncz
        sn = np.empty(nn, dtype=dt)
        getattr(L, "fsm_cells_to_nodes2d_" + sfx)(C.c_size_t(ncx), C.c_size_t(ncz), _p(s), _p(sn))
    else:
        assert s.size == nn
        sn = s
    src = _prep_pts(dt, src, 2)
    nsrc = src.shape[0]
    t0 = np.zeros(nsrc, dtype=dt) if t0 is None else np.ascontiguousarray(np.asarray(t0, dtype=dt))
    if getattr(L, "fsm_outside2d_" + sfx)(C.byref(g), C.c_int(nsrc), _p(src)):
        raise RuntimeError("Error: Point outside grid.")
    T = np.empty(nn, dtype=dt)
    hist = np.zeros(2 * maxit, dtype=dt)
    nw = C.c_int(0)
    niter = getattr(L, "fsm_solve2d_" + sfx)(C.byref(g), _p(sn), C.c_int(nsrc), _p(src), _p(t0), ct(eps),
                                             C.c_int(maxit), C.c_int(int(bool(weno)) | (2 if rotated else 0)), _p(T), _p(hist),
                                             C.byref(nw))
    out = dict(tt=T, niter=int(niter), niterw=int(nw.value), change=hist[:niter].copy(),
               changew=hist[maxit:maxit + nw.value].copy(), node_slowness=sn)
    if rcv is not None:
        r_chk = _prep_pts(dt, rcv, 2)
        if getattr(L, "fsm_outside2d_" + sfx)(C.byref(g), C.c_int(r_chk.shape[0]), _p(r_chk)):   # Grid2Drnfs::raytrace: checkPts(Rx)
            raise RuntimeError("Error: Point outside grid.")
    if rcv is not None and compute_L:
        r = _prep_pts(dt, rcv, 2)
        fl = getattr(L, "fsm_raypath2d_l_" + sfx)
        vals = np.empty(r.shape[0], dtype=dt)
        rays, ls = [], []
        cap = 4 * (ncx + ncz) + 64
        sc_p = _p(s) if cell_slowness else None
        for n, pnt in enumerate(r):
            pp = np.ascontiguousarray(pnt, dtype=dt)
            v = ct(0)
            while True:
                buf = np.empty((cap, 2), dtype=dt)
                lc = np.empty(cap, dtype=np.int64)
                lv = np.empty(cap, dtype=dt)
                npts, nl = C.c_long(0), C.c_long(0)
                rc = fl(C.byref(g), _p(sn), sc_p, _p(T), C.c_int(nsrc), _p(src), _p(t0), _p(pp), C.c_int(int(return_rays)),
                        C.c_long(1000000), C.byref(v), _p(buf), C.c_long(cap), C.byref(npts), _p(lc), _p(lv), C.c_long(cap), C.byref(nl))
                if rc != 3:
                    break
                cap *= 4
            if rc == 1:
                raise RuntimeError("Error while computing raypaths: going outside grid")
            if rc == 2:
                raise RuntimeError("raypath did not reach the source")
            vals[n] = v.value
            rays.append(buf[:npts.value].copy())
            o = np.argsort(lc[:nl.value], kind="stable")
            ls.append((lc[:nl.value][o].copy(), lv[:nl.value][o].copy()))
        out["tt_rcv"] = vals
        out["l"] = ls
        if return_rays:
            out["rays"] = rays
    elif rcv is not None and (tt_from_rp or return_rays):
        r = _prep_pts(dt, rcv, 2)
        frp = getattr(L, "fsm_raypath2d_" + sfx)
        vals = np.empty(r.shape[0], dtype=dt)
        rays = []
        cap = 4 * (ncx + ncz) + 64
        sc_p = _p(s) if cell_slowness else None   # Grid2Drcfs keeps the cell slowness (hasCellSlowness)
        for n, pnt in enumerate(r):
            pp = np.ascontiguousarray(pnt, dtype=dt)
            v = ct(0)
            while True:
                buf = np.empty((cap, 2), dtype=dt)
                npts = C.c_long(0)
                rc = frp(C.byref(g), _p(sn), sc_p, _p(T), C.c_int(nsrc), _p(src), _p(t0), _p(pp), C.c_int(int(return_rays)),
                         C.c_long(1000000), C.byref(v), _p(buf), C.c_long(cap), C.byref(npts))
                if rc != 3:
                    break
                cap *= 4
            if rc == 1:
                raise RuntimeError("Error while computing raypaths: going outside grid")
            if rc == 2:
                raise RuntimeError("raypath did not reach the source")
            vals[n] = v.value
            rays.append(buf[:npts.value].copy())
        out["tt_rcv"] = vals
        if return_rays:
            out["rays"] = rays
    elif rcv is not None:
        r = _prep_pts(dt, rcv, 2)
        f = getattr(L, "fsm_interp2d_" + sfx)
        out["tt_rcv"] = np.array([f(C.byref(g), _p(T), ct(p[0]), ct(p[1])) for p in r], dtype=dt)
    return out


def ref_solve2d(dtype, ncells, dx, dz, origin, slowness, src, t0=None, eps=1e-5, maxit=50,
                cell_slowness=False, rcv=None, weno=False, rotated=False, tt_from_rp=False, return_rays=False, compute_L=False):
    dt = np.dtype(dtype)
    sfx, ct = _TYPES[dt][:2]
    R = ref()
    ncx, ncz = (int(v) for v in ncells)
    nn = (ncx + 1) * (ncz + 1)
    s = np.ascontiguousarray(np.asarray(slowness, dtype=dt).ravel())
    src = _prep_pts(dt, src, 2)
    nsrc = src.shape[0]
    t0 = np.zeros(nsrc, dtype=dt) if t0 is None else np.ascontiguousarray(np.asarray(t0, dtype=dt))
    r = _prep_pts(dt, rcv if rcv is not None else np.zeros((0, 2)), 2)
    tt_rcv = np.empty(r.shape[0], dtype=dt)
    T = np.empty(nn, dtype=dt)
    niter = (C.c_int * 2)()
    cap = (4 * (ncx + ncz) + 64) * max(r.shape[0], 1)
    while True:
        if return_rays:
            rbuf = np.empty((cap, 2), dtype=np.float64)
            roff = np.zeros(r.shape[0] + 1, dtype=np.int64)
            R.ref_set_rays(_p(rbuf), C.c_long(cap), _p(roff))
        if compute_L:
            lcb = np.empty(cap, dtype=np.int64)
            lvb = np.empty(cap, dtype=np.float64)
            loff = np.zeros(r.shape[0] + 1, dtype=np.int64)
            R.ref_set_l(_p(lcb), _p(lvb), C.c_long(cap), _p(loff))
        try:
            rc = getattr(R, "ref_fsm2d_" + sfx)(C.c_int(int(cell_slowness)), C.c_uint32(ncx), C.c_uint32(ncz), ct(dx),
                                                ct(dz), ct(origin[0]), ct(origin[1]), ct(eps), C.c_int(maxit),
                                                C.c_int(int(weno)), C.c_int(int(rotated)), _p(s), C.c_int(nsrc), _p(src),
                                                _p(t0), C.c_int(r.shape[0]), _p(r), _p(tt_rcv), _p(T), niter,
                                                C.c_int(int(tt_from_rp)))
        finally:
            if return_rays:
                R.ref_set_rays(None, C.c_long(0), None)
            if compute_L:
                R.ref_set_l(None, None, C.c_long(0), None)
        if rc != 0:
            raise RuntimeError(R.ref_last_error().decode())
        need = max(int(roff[-1]) if return_rays else 0, int(loff[-1]) if compute_L else 0)
        if need <= cap:
            break
        cap = need
    out = dict(tt=T, niter=int(niter[0]), niterw=int(niter[1]), tt_rcv=tt_rcv)
    if return_rays:
        out["rays"] = [rbuf[roff[n]:roff[n + 1]].astype(dt) for n in range(r.shape[0])]
    if compute_L:
        out["l"] = [(lcb[loff[n]:loff[n + 1]].copy(), lvb[loff[n]:loff[n + 1]].astype(dt)) for n in range(r.shape[0])]
    return out


# --------------------------------------------------------------------------- file formats (reference side)


def ref_set_save(base, all=0, format=1):
    """make the next ref_solve* also call saveTT(base, all, 0, format) on the reference grid ('' = off)"""
    ref().ref_set_save(C.c_char_p((base or "").encode()), C.c_int(int(all)), C.c_int(int(format)))


def ref_read_src(fname, max_n=100000):
    xyz = np.zeros((max_n, 3))
    t0 = np.zeros(max_n)
    R = ref()
    R.ref_read_src.restype = C.c_int
    n = R.ref_read_src(C.c_char_p(fname.encode()), _p(xyz), _p(t0), C.c_int(max_n))
    return xyz[:n].copy(), t0[:n].copy()


def ref_read_rcv(fname, max_n=100000):
    xyz = np.zeros((max_n, 3))
    R = ref()
    R.ref_read_rcv.restype = C.c_int
    n = R.ref_read_rcv(C.c_char_p(fname.encode()), _p(xyz), C.c_int(max_n))
    return xyz[:n].copy()


def ref_write_rcv(rcvfile, ttfile, xyz, tt):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    tt = np.ascontiguousarray(tt, dtype=np.float64)
    ref().ref_write_rcv(C.c_char_p(rcvfile.encode()), C.c_char_p(ttfile.encode()), C.c_int(xyz.shape[0]), _p(xyz), _p(tt))
