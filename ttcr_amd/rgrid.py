"""ttcr_amd.rgrid -- the `Grid3d` / `Grid2d` subset of ttcrpy.rgrid for method='FSM',
running on MI355X through the C ABI of include/ttcr_amd.h.

Mirrors src/ttcrpy/rgrid.pyx of the reference (same names, argument meaning, return shapes
and error behaviour) for the fast-sweeping path:

  Grid3d factory          rgrid.pyx:5580-5620     Grid2d factory          rgrid.pyx:5646-5687
  Grid3d_d.__cinit__      rgrid.pyx:155-282       Grid2d_d.__cinit__      rgrid.pyx:2859-2975
  set_slowness            rgrid.pyx:532-569       (2-D) rgrid.pyx:3171-3205
  raytrace                rgrid.pyx:828-1199      (2-D) rgrid.pyx:3804-4143
  get_grid_traveltimes    rgrid.pyx:410-435       (2-D) rgrid.pyx:3102-3127

weno=True (the reference's default: first-order sweeps, then third-order WENO sweeps) is
supported, and so is tt_from_rp=True in 3-D (the 3-D default: traveltimes integrated along the
ray traced back through the field).  What is not on the FSM hot path raises NotImplementedError
(SPM/DSPM, compute_L).  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib

_verbose = 0


def set_verbose(v):
    """set_verbose(v): mirror of rgrid.pyx:38-47 (messages are only printed by this wrapper)."""
    global _verbose
    _verbose = int(v)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _row_groups(a):
    """Group equal rows of `a`: (order, bounds, keys) with order = the stable row-lexicographic argsort of the
    rows, bounds[g]:bounds[g+1] the slice of `order` holding group g (row numbers ascending), keys[g] its row
    (-0.0 folded into 0.0, like an elementwise ==).  Runs of consecutive equal rows -- the usual layout, every
    source repeated once per receiver -- are collapsed first, so the sort only sees one row per run."""
    key = np.ascontiguousarray(a, dtype=np.float64) + 0.0
    n = key.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.intp), np.zeros(1, dtype=np.intp), key
    ne = key[1:] != key[:-1]
    diff = ne[:, 0].copy()
    for c in range(1, key.shape[1]):
        diff |= ne[:, c]
    run0 = np.r_[0, np.nonzero(diff)[0] + 1]             # first row of every run
    lens = np.diff(np.r_[run0, n])
    reps = key[run0]
    ro = np.lexsort(reps.T[::-1])                        # stable: runs of a group stay in row order
    reps, run0, lens = reps[ro], run0[ro], lens[ro]
    first = np.cumsum(lens) - lens                       # position of every (sorted) run in `order`
    order = np.arange(n) - np.repeat(first, lens) + np.repeat(run0, lens)
    g0 = np.r_[True, np.any(reps[1:] != reps[:-1], axis=1)]
    return order, np.r_[first[g0], n], reps[g0]


def _first_rows(a):
    """np.sort(np.unique(a, axis=0, return_index=True)[1]) -- the index of the first occurrence of every distinct
    row, ascending (rgrid.pyx:926-938) -- without the structured-dtype sort behind np.unique (14 ms for 64
    sources x 441 receivers)"""
    order, bounds, _ = _row_groups(a)
    return np.sort(order[bounds[:-1]])


def _device_arg(device):
    """`device` keyword of the grid classes: a HIP device ordinal (-1: the current device; with TTCR_AMD_DEVICES set,
    the devices it lists), or a sequence of ordinals -- one replica of the grid per entry, the traveltime slots
    (n_threads) divided over them and the sources of a call block-distributed over all slots (ttcr_fsm3d_create_multi)."""
    if isinstance(device, (list, tuple, np.ndarray)):
        devs = tuple(int(d) for d in device)
        if not devs:
            raise ValueError("device list is empty")
        return devs if len(devs) > 1 else devs[0]
    return int(device)


class _GridBase:
    """State and helpers shared by the 3-D and 2-D wrappers."""

    _dtype = np.float64
    _ndim = 3

    def __init__(self):
        self._h = C.c_void_p()
        self._lib = None

    # -- life cycle (rgrid.pyx:284-285)
    def __del__(self):
        try:
            if self._lib is not None and self._h:
                self._lib.ttcr_fsm_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @property
    def n_threads(self):
        """int: number of threads (here: device-resident traveltime slots) for raytracing"""
        return self._n_threads

    @property
    def n_devices(self):
        """int: replicas of the grid (one per device of the `device` list) behind this object"""
        return int(self._lib.ttcr_fsm_n_devices(self._h))

    def set_use_thread_pool(self, use_thread_pool):
        """No-op: sources of one call are always swept concurrently on the device
        (the reference's pool/blocks choice, ttcr/Grid3D.h:821-851, has no equivalent)."""
        self._use_pool = bool(use_thread_pool)

    def set_traveltime_from_raypath(self, traveltime_from_raypath):
        """Set option to compute traveltime using raypath (rgrid.pyx:373-384)"""
        self.set_option("tt_from_rp", 1 if traveltime_from_raypath else 0)
        self.tt_from_rp = bool(traveltime_from_raypath)

    def set_option(self, key, value):
        """Backend knob (fixed_iters, max_batch, use_graph); no reference equivalent."""
        _lib.check(self._lib.ttcr_fsm_set_option(self._h, key.encode(), float(value)))

    def set_slowness_device(self, device_ptr, n):
        """Assign slowness that is already resident in HBM on this grid's device: `device_ptr`
        is a raw device address (e.g. torch_tensor.data_ptr()) of `n` values of the grid dtype in
        the solver's flat order (3-D: x-fastest, 2-D: z-fastest).  No PCIe copy."""
        _lib.check(self._lib.ttcr_fsm_set_slowness_device(self._h, C.c_void_p(int(device_ptr)), int(n)))

    def get_niter(self, thread_no=0):
        """Sweep-iterations of the last solve in a slot (Grid3Drnfs::get_niter, ttcr/Grid3Drnfs.h:56)."""
        a, b = C.c_int(), C.c_int()
        _lib.check(self._lib.ttcr_fsm_get_niter(self._h, int(thread_no), C.byref(a), C.byref(b)))
        return a.value

    def get_niterw(self, thread_no=0):
        """WENO sweep-iterations of the last solve in a slot (get_niterw, ttcr/Grid3Drnfs.h:57)."""
        a, b = C.c_int(), C.c_int()
        _lib.check(self._lib.ttcr_fsm_get_niter(self._h, int(thread_no), C.byref(a), C.byref(b)))
        return b.value

    def get_changes(self, thread_no=0):
        """(first-order, WENO) arrays with the L1 change of every sweep-iteration of the last solve in a slot -- the
        quantity the stopping rule compares with eps * N (`change`, ttcr/Grid3Drnfs.h:141-152); fp64 sums of decreases here."""
        n1, n2 = self.get_niter(thread_no), self.get_niterw(thread_no)
        a, b = (C.c_double * max(n1, 1))(), (C.c_double * max(n2, 1))()
        _lib.check(self._lib.ttcr_fsm_get_changes(self._h, int(thread_no), a, n1, b, n2))
        return np.array(a[:n1]), np.array(b[:n2])

    def get_reference_changes(self, thread_no=0):
        """(first-order, WENO) arrays like get_changes(), holding the reference's own sum -- sequential, in node order, in the grid's
        precision (ttcr/Grid3Drnfs.h:141-152) -- for the iterations that were decided with it (option stopping_rule), NaN elsewhere."""
        n1, n2 = self.get_niter(thread_no), self.get_niterw(thread_no)
        a, b = (C.c_double * max(n1, 1))(), (C.c_double * max(n2, 1))()
        _lib.check(self._lib.ttcr_fsm_get_reference_changes(self._h, int(thread_no), a, n1, b, n2))
        return np.array(a[:n1]), np.array(b[:n2])

    def timing(self):
        """HIP-event timing of the last raytrace call (dict)."""
        t = _lib.Timing()
        _lib.check(self._lib.ttcr_fsm_last_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def stopping_stats(self):
        """Decisions of the stopping rule since the grid was created (dict): iterations decided by the reference's sequential
        sum, iterations that landed in its window without a snapshot, rounds of the parallel sum (ttcr_fsm_stopping_stats)."""
        a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        _lib.check(self._lib.ttcr_fsm_stopping_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"reference_sums": a.value, "reference_sums_missed": b.value, "rounds": c.value}

    def prefill_swaps(self):
        """Calls that took traveltime fields initialised on the side stream (option "prefill", ttcr_fsm_prefill_swaps)."""
        a = C.c_longlong(0)
        _lib.check(self._lib.ttcr_fsm_prefill_swaps(self._h, C.byref(a)))
        return a.value

    def reference_change(self, times, field, parallel=True):
        """The reference's `change` of two fields (ttcr/Grid3Drnfs.h:141-152): the sequential sum, in node order and in the grid's
        precision, of abs(times[n] - field[n]).  Flat arrays of get_number_of_nodes() values in node order."""
        dt = self._dtype
        a = np.ascontiguousarray(times, dtype=dt).ravel()
        b = np.ascontiguousarray(field, dtype=dt).ravel()
        n = self.get_number_of_nodes()
        if a.size != n or b.size != n:
            raise ValueError('reference_change: fields of get_number_of_nodes() values expected')
        out = np.zeros(1, dtype=dt)
        _lib.check(self._lib.ttcr_fsm_reference_change(self._h, a.ctypes.data, b.ctypes.data, 1 if parallel else 0, out.ctypes.data))
        return out[0]

    def last_kernel(self):
        """Instantiation of the sweep kernel the last solve launched (string)."""
        buf = C.create_string_buffer(256)
        _lib.check(self._lib.ttcr_fsm_last_kernel(self._h, buf, 256))
        return buf.value.decode()

    def _flat_tt(self, thread_no):
        if thread_no >= self._n_threads:
            raise ValueError('Thread number is larger than number of threads')
        n = self.get_number_of_nodes()
        out = np.empty(n, dtype=self._dtype)
        _lib.check(self._lib.ttcr_fsm_get_tt(self._h, int(thread_no), _ptr(out), n))
        return out

    def get_s0(self, hypo, slowness=None):
        """get_s0(hypo, slowness=None): slowness at the source points (rgrid.pyx:758-826, 2-D :3735-3802).
        hypo: event ID, origin time, source coordinates (5 columns in 3-D, 4 in 2-D); every row of an event gets the
        value at the event's first row (Grid3D::computeSlowness there)."""
        nd = self._ndim
        hypo = np.asarray(hypo)
        if hypo.ndim != 2 or hypo.shape[1] != nd + 2:
            raise ValueError('hypo should be npts x %d' % (nd + 2))
        src = hypo[:, 2:2 + nd]
        evID = hypo[:, 0]
        eid = np.sort(np.unique(evID))
        if slowness is not None:
            self.set_slowness(slowness)
        first = np.array([int(np.nonzero(evID == e)[0][0]) for e in eid], dtype=np.int64)
        pts = np.ascontiguousarray(src[first, :], dtype=self._dtype)
        out = np.empty(max(len(eid), 1), dtype=self._dtype)
        _lib.check(self._lib.ttcr_fsm_compute_slowness(self._h, len(eid), _ptr(pts), 0, _ptr(out)))
        s0 = np.zeros((src.shape[0],))
        for n, e in enumerate(eid):
            s0[evID == e] = out[n]   # s / vTx[n].size() with one point per event
        return s0

    def tt_device_ptr(self, thread_no=0):
        """Raw device address of n_nodes CONTIGUOUS traveltimes of a slot, in the solver's flat order.  On a
        first-order 3-D grid with n_threads >= 2 the fields of two slots are interleaved in HBM and this is a
        de-interleaved copy in a scratch buffer of the grid (valid until the next tt_device_ptr /
        get_grid_traveltimes call); see tt_device_view for the zero-copy form."""
        p = C.c_void_p()
        _lib.check(self._lib.ttcr_fsm_get_tt_device(self._h, int(thread_no), C.byref(p)))
        return p.value

    def tt_device_view(self, thread_no=0):
        """(device address, element stride) of a slot's traveltime field where it lies: node n of the flat order is
        at address + n * stride * itemsize (stride 1, or 2 where two slots share an interleaved field: first-order 3-D
        grids with n_threads >= 2)."""
        p, st = C.c_void_p(), C.c_size_t()
        _lib.check(self._lib.ttcr_fsm_get_tt_device_view(self._h, int(thread_no), C.byref(p), C.byref(st)))
        return p.value, st.value

    # -- source / receiver bookkeeping shared by 3-D and 2-D raytrace()
    def to_vtk(self, fields, filename):
        """to_vtk(fields, filename): save node/cell variables to filename + '.vtr' and lists of raypaths
        to filename + '_' + key + '.vtp' (rgrid.pyx:1201-1312, 2-D :4145-4260).  Arrays may be shaped like
        the grid or flat in C order; the file is written in VTK order (x fastest), Float64, like the reference."""
        from . import io as _io

        if self._ndim == 3:
            nodes, cells = (self._x.size, self._y.size, self._z.size), (self._x.size - 1, self._y.size - 1, self._z.size - 1)
            xyz = (self._x, self._y, self._z)
        else:
            nodes, cells = (self._x.size, self._z.size), (self._x.size - 1, self._z.size - 1)
            xyz = (self._x, np.array([0.0]), self._z)
        pd, cd = {}, {}
        for fn in fields:
            data = fields[fn]
            if isinstance(data, list):
                _io.write_vtp_lines(filename + '_' + fn + '.vtp', data)
                continue
            data = np.asarray(data)
            for shape, dest in ((nodes, pd), (cells, cd)):
                if data.size == int(np.prod(shape)):
                    if data.ndim == len(shape):
                        if data.shape != shape:
                            raise ValueError('Field {0:s} has incorrect shape'.format(fn))
                        tmp = data.flatten(order='F')
                    elif data.ndim == 1 or (data.ndim == 2 and data.shape[0] == data.size):
                        tmp = data.reshape(shape).flatten(order='F')
                    else:
                        raise ValueError('Field {0:s} has incorrect ndim ({1:d})'.format(fn, data.ndim))
                    dest[fn] = tmp.astype(np.float64)
                    break
            else:
                raise ValueError('Field {0:s} has incorrect size'.format(fn))
        if pd or cd:
            _io.write_vtr(filename + '.vtr', xyz[0], xyz[1], xyz[2], point_data=pd, cell_data=cd)

    def _split_sources(self, source, rcv, aggregate_src):
        """The data rows of a raytrace call as a list of events -> (points, origin times, receivers, receiver rows), one
        entry per event.  What ttcrpy's raytrace does before it reaches C++ (rgrid.pyx:917-1030): an event is a distinct
        source row (3 / 4 columns: x y z / t0 x y z; 2-D one column less), taken in order of first appearance, or an
        event number (5 columns: evID t0 x y z, 3-D only), taken in ascending order; its receivers are the rows that carry
        it.  One event in all: every row is a receiver of it.  aggregate_src: the distinct rows are the points of ONE
        source.  Same checks, same messages."""
        nd = self._ndim
        if source.ndim != 2 or rcv.ndim != 2:
            raise ValueError('source and rcv should be 2D arrays')
        width = source.shape[1]
        by_number = nd == 3 and width == 5
        if not by_number and width not in (nd, nd + 1):
            raise ValueError('source should be either nsrc x 3, 4 or 5' if nd == 3 else 'source should be either nsrc x 2 or 3')
        xyz = source[:, width - nd:]                       # coordinates are always the last nd columns
        if xyz.shape[1] != nd or rcv.shape[1] != nd:
            raise ValueError('src and rcv should be ndata x %d' % nd)
        if self.is_outside(xyz):
            raise ValueError('Source point outside grid')
        if self.is_outside(rcv):
            raise ValueError('Receiver outside grid')
        times = source[:, width - nd - 1] if width > nd else None   # the column in front of the coordinates, when there is one
        everything = np.arange(rcv.shape[0])

        if by_number:
            if xyz.shape != rcv.shape:
                raise ValueError('src and rcv should be of equal size')
            numbers = source[:, 0]
            events = []
            for ev in np.unique(numbers):                  # ascending
                rows = np.nonzero(numbers == ev)[0]
                events.append((xyz[rows[:1]], times[rows[:1]].copy(), rows))
        else:
            # distinct rows of the whole source array (t0 included), first occurrences in row order
            order, bounds, _ = _row_groups(source)
            heads = order[bounds[:-1]]                     # first row of every group (the grouping sort is stable)
            rank = np.argsort(heads, kind='stable')        # groups in order of first appearance
            heads = heads[rank]
            t_of = (lambda r: times[r].copy()) if times is not None else (lambda r: np.zeros(len(r)))
            if len(heads) == 1:
                events = [(xyz[:1], t_of(heads), everything)]
            elif aggregate_src:
                events = [(xyz[heads], t_of(heads), everything)]
            else:
                if xyz.shape != rcv.shape:
                    raise ValueError('src and rcv should be of equal size')
                events = []
                for h, gidx in zip(heads, rank):
                    if times is None:
                        rows = order[bounds[gidx]:bounds[gidx + 1]]
                    else:
                        # the reference matches receivers on the coordinates alone (rgrid.pyx:1000-1007): rows with the
                        # same point and another origin time belong to every event at that point
                        rows = np.nonzero(np.all(xyz == xyz[h], axis=1))[0]
                    events.append((xyz[h:h + 1], t_of(np.array([h])), rows))
        pts = [e[0] for e in events]
        tms = [np.asarray(e[1], dtype=np.float64) for e in events]
        return pts, tms, [rcv[e[2], :] for e in events], [e[2] for e in events]

    def _run(self, vTx, vt0, vRx, iRx, n_rcv, thread_no, return_rays=False):
        dt = self._dtype
        nd = self._ndim
        nTx = len(vTx)
        tx = np.ascontiguousarray(np.vstack(vTx), dtype=dt)
        t0 = np.ascontiguousarray(np.concatenate(vt0), dtype=dt)
        rx = np.ascontiguousarray(np.vstack(vRx), dtype=dt).reshape(-1, nd)
        tx_off = np.zeros(nTx + 1, dtype=np.int32)
        rx_off = np.zeros(nTx + 1, dtype=np.int32)
        tx_off[1:] = np.cumsum([len(t) for t in vTx])
        rx_off[1:] = np.cumsum([len(r) for r in vRx])
        out = np.empty(max(rx.shape[0], 1), dtype=dt)
        if thread_no is not None and self._n_threads > 1:
            assert nTx == 1  # "we should be here for just one event" (rgrid.pyx:1064-1066)
            st = self._lib.ttcr_fsm_raytrace(self._h, int(thread_no), int(tx.shape[0]), _ptr(tx), _ptr(t0),
                                             int(rx.shape[0]), _ptr(rx), _ptr(out))
        else:
            st = self._lib.ttcr_fsm_raytrace_multi(self._h, nTx, _ptr(tx_off), _ptr(tx), _ptr(t0), _ptr(rx_off),
                                                   _ptr(rx), _ptr(out))
        _lib.check(st)
        tt = np.zeros((n_rcv,), dtype=dt)
        for n in range(nTx):
            tt[iRx[n]] = out[rx_off[n]:rx_off[n + 1]]
        if not return_rays:
            return tt
        # r_data of the raytrace overloads (rgrid.pyx:1096-1107): one (npts, 3) array per receiver row
        nr, npnt = C.c_size_t(0), C.c_size_t(0)
        _lib.check(self._lib.ttcr_fsm_rays_size(self._h, C.byref(nr), C.byref(npnt)))
        off = np.zeros(nr.value + 1, dtype=np.int64)
        pts = np.empty((max(npnt.value, 1), nd), dtype=dt)
        _lib.check(self._lib.ttcr_fsm_get_rays(self._h, _ptr(off), _ptr(pts)))
        rays = [[0.0] for _ in range(n_rcv)]
        for n in range(nTx):
            for k, row in enumerate(iRx[n]):
                a, b = off[rx_off[n] + k], off[rx_off[n] + k + 1]
                rays[row] = np.array(pts[a:b], dtype=np.float64)
        return tt, rays


# ======================================================================================= 3-D
class _Grid3d(_GridBase):
    """3-D rectilinear grid, fast-sweeping method on MI355X.

    Constructor (same signature as ttcrpy's Grid3d_d / Grid3d_f):

    Grid3d_x(x, y, z, n_threads=1, cell_slowness=1, method='FSM', tt_from_rp=1, interp_vel=0,
             eps=1.e-5, maxit=50, weno=1, nsnx=5, nsny=5, nsnz=5, n_secondary=2, n_tertiary=2,
             radius_factor_tertiary=3.0, translate_grid=False, fsm_gpu=False)

    x, y, z are NODE coordinates (cells = size-1).  weno=1 (default) runs the two-stage solver
    (first-order sweeps, then WENO3 sweeps).  fsm_gpu is accepted and ignored: this
    backend always runs on the GPU.  `device` (extra keyword) selects the HIP device.
    """
    _ndim = 3

    def __init__(self, x, y, z, n_threads=1, cell_slowness=1, method='FSM', tt_from_rp=1, interp_vel=0,
                 eps=1.e-5, maxit=50, weno=1, nsnx=5, nsny=5, nsnz=5, n_secondary=2, n_tertiary=2,
                 radius_factor_tertiary=3.0, translate_grid=False, fsm_gpu=False, device=-1):
        super().__init__()
        dt = self._dtype
        x = np.ascontiguousarray(x, dtype=dt).ravel()
        y = np.ascontiguousarray(y, dtype=dt).ravel()
        z = np.ascontiguousarray(z, dtype=dt).ravel()
        if x.size < 2 or y.size < 2 or z.size < 2:
            raise ValueError('x, y and z need at least two nodes')
        self._x, self._y, self._z = x, y, z
        # dx = x[1]-x[0] in the array's dtype, then widened (rgrid.pyx:170-172, :1878-1880)
        self._dx = float(x[1] - x[0])
        self._dy = float(y[1] - y[0])
        self._dz = float(z[1] - z[0])
        self.cell_slowness = bool(cell_slowness)
        self._n_threads = int(n_threads)
        self.tt_from_rp = bool(tt_from_rp)
        self.interp_vel = bool(interp_vel)
        self.eps = float(eps)
        self.maxit = int(maxit)
        self.weno = bool(weno)
        self.nsnx, self.nsny, self.nsnz = int(nsnx), int(nsny), int(nsnz)
        self.n_secondary, self.n_tertiary = int(n_secondary), int(n_tertiary)
        self.radius_factor_tertiary = float(radius_factor_tertiary)
        self.translate_grid = bool(translate_grid)
        self.fsm_gpu = bool(fsm_gpu)
        self.method = method
        self._device = _device_arg(device)

        if method == 'FSM':
            if np.abs(self._dx - self._dy) > 0.000001 or np.abs(self._dx - self._dz) > 0.000001:
                raise ValueError('FSM: Grid cells must be cubic')
        elif method in ('SPM', 'DSPM'):
            raise NotImplementedError("method '%s' is outside the MI355X FSM path (SURVEY.md section 8)" % method)
        else:
            raise ValueError('Method {0:s} undefined'.format(method))
        self._lib = _lib.load()
        args = (C.byref(self._h), _lib.TTCR_F32 if dt == np.float32 else _lib.TTCR_F64,
                int(self.cell_slowness), x.size - 1, y.size - 1, z.size - 1, self._dx,
                float(x[0]), float(y[0]), float(z[0]), self.eps, self.maxit, int(self.weno),
                self._n_threads, int(self.translate_grid))
        if isinstance(self._device, tuple):   # one replica of the grid per listed device, slots divided (ttcr_amd.h)
            devs = (C.c_int * len(self._device))(*self._device)
            st = self._lib.ttcr_fsm3d_create_multi(*args, devs, len(self._device))
        else:
            st = self._lib.ttcr_fsm3d_create(*args, self._device)
        _lib.check(st)
        if self.tt_from_rp:
            self.set_option("tt_from_rp", 1)
        if self.interp_vel:
            self.set_option("interp_vel", 1)

    def __reduce__(self):
        params = (self.n_threads, self.cell_slowness, self.method, self.tt_from_rp, self.interp_vel, self.eps,
                  self.maxit, self.weno, self.nsnx, self.nsny, self.nsnz, self.n_secondary, self.n_tertiary,
                  self.radius_factor_tertiary, self.translate_grid, self.fsm_gpu)
        return (_rebuild3d, (type(self), self.x, self.y, self.z, params))

    # -- properties (rgrid.pyx:303-364)
    @property
    def x(self):
        """np.ndarray: node coordinates along x"""
        return np.array(self._x)

    @property
    def y(self):
        """np.ndarray: node coordinates along y"""
        return np.array(self._y)

    @property
    def z(self):
        """np.ndarray: node coordinates along z"""
        return np.array(self._z)

    @property
    def dx(self):
        return self._dx

    @property
    def dy(self):
        return self._dy

    @property
    def dz(self):
        return self._dz

    @property
    def shape(self):
        """number of parameters along each dimension"""
        if self.cell_slowness:
            return (self._x.size - 1, self._y.size - 1, self._z.size - 1)
        return (self._x.size, self._y.size, self._z.size)

    @property
    def nparams(self):
        return int(np.prod(self.shape))

    def get_number_of_nodes(self):
        return self._x.size * self._y.size * self._z.size

    def get_number_of_cells(self):
        return (self._x.size - 1) * (self._y.size - 1) * (self._z.size - 1)

    def ind(self, i, j, k):
        return (i * self._y.size + j) * self._z.size + k

    def indc(self, i, j, k):
        return (i * (self._y.size - 1) + j) * (self._z.size - 1) + k

    def is_outside(self, pts):
        """True if at least one point is outside the grid (rgrid.pyx:487-505)"""
        pts = np.asarray(pts)
        return bool(np.min(pts[:, 0]) < self._x[0] or np.max(pts[:, 0]) > self._x[-1] or
                    np.min(pts[:, 1]) < self._y[0] or np.max(pts[:, 1]) > self._y[-1] or
                    np.min(pts[:, 2]) < self._z[0] or np.max(pts[:, 2]) > self._z[-1])

    def compute_D(self, coord):
        """compute_D(coord) -> csr_matrix (npts x nparams) of interpolation weights for velocity data points (rgrid.pyx:610-677)"""
        from . import inversion as _inv
        coord = np.asarray(coord, dtype=np.float64)
        if self.is_outside(coord):
            raise ValueError('Velocity data point outside grid')
        return _inv.interp_matrix((self._x, self._y, self._z), coord, self.cell_slowness)

    def compute_K(self):
        """compute_K() -> (Kx, Ky, Kz): second-derivative smoothing matrices on the parameters (rgrid.pyx:679-756)"""
        from . import inversion as _inv
        return _inv.smoothing_matrices(self.shape, (self.dx, self.dy, self.dz), order=2)

    def _save_raypaths(self, rays, filename):
        """rays (list of npts x 3 arrays) as a vtkPolyData of poly-lines (rgrid.pyx:1284-1312)"""
        from . import io as _io
        _io.write_vtp_lines(filename, rays)

    @staticmethod
    def data_kernel_straight_rays(Tx, Rx, grx, gry, grz, centers=False):
        """data_kernel_straight_rays(Tx, Rx, grx, gry, grz, centers=False) -> L[, (xc, yc, zc)]: straight rays through the cells of
        the grid with node coordinates grx, gry, grz; tt = L @ slowness (rgrid.pyx:1381-1816)"""
        from . import inversion as _inv
        grx, gry, grz = (np.asarray(a, dtype=np.float64) for a in (grx, gry, grz))
        L = _inv.straight_ray_kernel(Tx, Rx, (grx, gry, grz))
        if centers:
            return L, ((grx[1:] + grx[:-1]) / 2, (gry[1:] + gry[:-1]) / 2, (grz[1:] + grz[:-1]) / 2)
        return L

    def get_grid_traveltimes(self, thread_no=0):
        """traveltimes at the grid nodes, shape (nx, ny, nz)  (rgrid.pyx:410-435)"""
        tt = self._flat_tt(thread_no)
        return tt.reshape((self._x.size, self._y.size, self._z.size), order='F')

    def get_slowness(self):
        """NODE slowness held by the solver, shape (nx, ny, nz) of nodes."""
        n = self.get_number_of_nodes()
        out = np.empty(n, dtype=self._dtype)
        _lib.check(self._lib.ttcr_fsm_get_slowness(self._h, _ptr(out), n))
        return out.reshape((self._x.size, self._y.size, self._z.size), order='F')

    def _to_flat_F(self, a, what):
        nx, ny, nz = self.shape
        a = np.asarray(a)
        if a.size != nx * ny * nz:
            raise ValueError('%s vector has wrong size' % what)
        if a.ndim == 3:
            if a.shape != (nx, ny, nz):
                raise ValueError('%s has wrong shape' % what)
            return a.flatten('F')
        elif a.ndim == 1:
            # C order in, x-fastest ('F') order to the solver (rgrid.pyx:562-565)
            return a.reshape((nx, ny, nz)).flatten('F')
        raise ValueError('%s must be 1D or 3D ndarray' % what)

    def _c_order(self, a, what):
        """the checks of _to_flat_F without its strided copy: the array in C order, flat"""
        nx, ny, nz = self.shape
        a = np.asarray(a)
        if a.size != nx * ny * nz:
            raise ValueError('%s vector has wrong size' % what)
        if a.ndim == 3:
            if a.shape != (nx, ny, nz):
                raise ValueError('%s has wrong shape' % what)
        elif a.ndim != 1:
            raise ValueError('%s must be 1D or 3D ndarray' % what)
        return np.ascontiguousarray(a).reshape(-1)

    def set_slowness(self, slowness):
        """Assign slowness, shape (nx, ny, nz) or flattened in C order (rgrid.pyx:532-569); the permutation to
        the solver's x-fastest order is done on the device (ttcr_fsm_set_slowness_c_order)"""
        s = np.ascontiguousarray(self._c_order(slowness, 'Slowness'), dtype=self._dtype)
        _lib.check(self._lib.ttcr_fsm_set_slowness_c_order(self._h, _ptr(s), s.size))

    def set_velocity(self, velocity):
        """Assign velocity (rgrid.pyx:571-608): slowness = 1/velocity"""
        v = self._c_order(velocity, 'velocity')
        s = np.ascontiguousarray(1. / v, dtype=self._dtype)
        _lib.check(self._lib.ttcr_fsm_set_slowness_c_order(self._h, _ptr(s), s.size))

    def raytrace(self, source, rcv, slowness=None, thread_no=None, aggregate_src=False, compute_L=False,
                 compute_M=False, return_rays=False):
        """raytrace(source, rcv, slowness=None, thread_no=None, aggregate_src=False, ...) -> tt

        Same contract as ttcrpy (rgrid.pyx:828-1199): source has 3 (xyz), 4 (t0,xyz) or 5
        (evID,t0,xyz) columns; duplicate source rows are collapsed keeping first-occurrence
        order; receivers are grouped per unique source; tt comes back in rcv order.
        """
        source = np.asarray(source)
        rcv = np.asarray(rcv)
        if source.ndim != 2 or rcv.ndim != 2:
            raise ValueError('source and rcv should be 2D arrays')
        if compute_L and compute_M:
            raise ValueError('compute_L and compute_M are mutually exclusive')
        if self.cell_slowness and compute_M:
            raise NotImplementedError('compute_M not defined for grids with slowness defined for cells')
        if compute_L and not self.cell_slowness:
            raise NotImplementedError('compute_L defined only for grids with slowness defined for cells')
        if compute_L:
            raise NotImplementedError('compute_L defined for the FSM')
        vTx, vt0, vRx, iRx = self._split_sources(source, rcv, aggregate_src)
        if slowness is not None:
            self.set_slowness(slowness)
        if compute_M:
            return self._run_m(vTx, vt0, vRx, iRx, rcv.shape[0], return_rays)
        if not return_rays:
            return self._run(vTx, vt0, vRx, iRx, rcv.shape[0], thread_no)
        # -> (tt, rays): the overloads with r_data; traveltimes are then integrated along the rays whatever
        # tt_from_rp says (Grid3D::raytrace, ttcr/Grid3D.h:546-586 -> getRaypath, ttcr/Grid3Drn.h:1339-1500)
        self.set_option("return_rays", 1)
        try:
            return self._run(vTx, vt0, vRx, iRx, rcv.shape[0], thread_no, return_rays=True)
        finally:
            self.set_option("return_rays", 0)

    def _run_m(self, vTx, vt0, vRx, iRx, n_rcv, return_rays):
        """compute_M=True (rgrid.pyx:1040-1060, :1171-1199): per event the overload with m_data (with r_data and m_data when the
        rays are asked for as well), then one scipy CSR matrix
        (receivers of the event x nodes) per event, columns ascending, entries whose node index is not below the node count
        dropped -- what the reference's Python layer builds.  Traveltimes are those of that overload."""
        import scipy.sparse as sp

        dt = self._dtype
        tt = np.zeros((n_rcv,), dtype=dt)
        rays = [[0.0] for _ in range(n_rcv)]
        M = []
        NN = self.get_number_of_nodes()

        def csr(off, jj, vv, r0, nrows):
            indptr = np.zeros(nrows + 1, dtype=np.int64)
            ind, val = [], []
            for i in range(nrows):
                j, v = jj[off[r0 + i]:off[r0 + i + 1]], vv[off[r0 + i]:off[r0 + i + 1]]
                keep = j < NN
                j, v = j[keep], v[keep]
                o = np.argsort(j, kind='stable')
                ind.append(j[o]); val.append(v[o].astype(np.float64))
                indptr[i + 1] = indptr[i] + j.size
            return sp.csr_matrix((np.concatenate(val) if val else np.zeros(0), np.concatenate(ind) if ind else np.zeros(0, dtype=np.int64),
                                  indptr), shape=(nrows, NN))

        if len(vTx) > 1:
            # several events: ONE call -- the fields are solved in batches of n_threads sources, the walks follow each batch
            # (what Grid3D's multi-source overloads with m_data do with host threads, rgrid.pyx:1096-1102)
            tx = np.ascontiguousarray(np.vstack(vTx), dtype=dt).reshape(-1, 3)
            t0 = np.ascontiguousarray(np.concatenate(vt0), dtype=dt)
            rx = np.ascontiguousarray(np.vstack(vRx), dtype=dt).reshape(-1, 3)
            tx_off = np.zeros(len(vTx) + 1, dtype=np.int32); rx_off = np.zeros(len(vTx) + 1, dtype=np.int32)
            tx_off[1:] = np.cumsum([len(t) for t in vTx]); rx_off[1:] = np.cumsum([len(r) for r in vRx])
            out = np.empty(max(rx.shape[0], 1), dtype=dt)
            _lib.check(self._lib.ttcr_fsm_raytrace_multi_m(self._h, len(vTx), _ptr(tx_off), _ptr(tx), _ptr(t0), _ptr(rx_off), _ptr(rx),
                                                           _ptr(out), int(bool(return_rays))))
            nrow, nnz = C.c_size_t(0), C.c_size_t(0)
            _lib.check(self._lib.ttcr_fsm_multi_m_size(self._h, C.byref(nrow), C.byref(nnz)))
            off = np.zeros(nrow.value + 1, dtype=np.int64)
            jj = np.empty(max(nnz.value, 1), dtype=np.int64)
            vv = np.empty(max(nnz.value, 1), dtype=dt)
            _lib.check(self._lib.ttcr_fsm_get_multi_m(self._h, _ptr(off), _ptr(jj), _ptr(vv)))
            if return_rays:
                nr, npnt = C.c_size_t(0), C.c_size_t(0)
                _lib.check(self._lib.ttcr_fsm_rays_size(self._h, C.byref(nr), C.byref(npnt)))
                roff = np.zeros(nr.value + 1, dtype=np.int64)
                pts = np.empty((max(npnt.value, 1), 3), dtype=dt)
                _lib.check(self._lib.ttcr_fsm_get_rays(self._h, _ptr(roff), _ptr(pts)))
            for n in range(len(vTx)):
                tt[iRx[n]] = out[rx_off[n]:rx_off[n + 1]]
                M.append(csr(off, jj, vv, int(rx_off[n]), int(rx_off[n + 1] - rx_off[n])))
                if return_rays:
                    for k, row in enumerate(iRx[n]):
                        rays[row] = np.array(pts[roff[rx_off[n] + k]:roff[rx_off[n] + k + 1]], dtype=np.float64)
            return (tt, rays, M) if return_rays else (tt, M)
        for n in range(len(vTx)):
            slot = n % self._n_threads
            tx = np.ascontiguousarray(vTx[n], dtype=dt)
            t0 = np.ascontiguousarray(vt0[n], dtype=dt)
            rx = np.ascontiguousarray(vRx[n], dtype=dt).reshape(-1, 3)
            out = np.empty(rx.shape[0], dtype=dt)
            # (with return_rays ttcrpy calls the overload that keeps r_data AND m_data, rgrid.pyx:1050 -- its matrix is another one)
            call = self._lib.ttcr_fsm_raytrace_rm if return_rays else self._lib.ttcr_fsm_raytrace_m
            _lib.check(call(self._h, slot, tx.shape[0], _ptr(tx), _ptr(t0), rx.shape[0], _ptr(rx), _ptr(out)))
            tt[iRx[n]] = out
            nrow, nnz = C.c_size_t(0), C.c_size_t(0)
            _lib.check(self._lib.ttcr_fsm_slot_m_size(self._h, slot, C.byref(nrow), C.byref(nnz)))
            off = np.zeros(nrow.value + 1, dtype=np.int64)
            jj = np.empty(max(nnz.value, 1), dtype=np.int64)
            vv = np.empty(max(nnz.value, 1), dtype=dt)
            _lib.check(self._lib.ttcr_fsm_get_slot_m(self._h, slot, _ptr(off), _ptr(jj), _ptr(vv)))
            M.append(csr(off, jj, vv, 0, rx.shape[0]))
            if return_rays:
                nr, npnt = C.c_size_t(0), C.c_size_t(0)
                _lib.check(self._lib.ttcr_fsm_slot_rays_size(self._h, slot, C.byref(nr), C.byref(npnt)))
                roff = np.zeros(nr.value + 1, dtype=np.int64)
                pts = np.empty((max(npnt.value, 1), 3), dtype=dt)
                _lib.check(self._lib.ttcr_fsm_get_slot_rays(self._h, slot, _ptr(roff), _ptr(pts)))
                for k, row in enumerate(iRx[n]):
                    rays[row] = np.array(pts[roff[k]:roff[k + 1]], dtype=np.float64)
        if return_rays:
            return tt, rays, M
        return tt, M


def _builder3d(cls, filename, n_threads=1, method='FSM', tt_from_rp=1, interp_vel=0, eps=1.e-5, maxit=50, weno=1,
               nsnx=5, nsny=5, nsnz=5, n_secondary=2, n_tertiary=2, radius_factor_tertiary=3.0, translate_grid=0,
               device=-1):
    """builder(filename, n_threads=1, method='FSM', ...): grid from a vtkRectilinearGrid file holding a point
    or cell array named Slowness, slowness, Velocity, velocity or P-wave velocity (rgrid.pyx:1315-1379)"""
    from . import io as _io

    m = _io.model_from_vtr(filename)
    x, y, z = m['x'], m['y'], m['z']
    dim = (x.size - 1, y.size - 1, z.size - 1) if m['cell_slowness'] else (x.size, y.size, z.size)
    g = cls(x, y, z, n_threads, m['cell_slowness'], method, tt_from_rp, interp_vel, eps, maxit, weno, nsnx, nsny,
            nsnz, n_secondary, n_tertiary, radius_factor_tertiary, translate_grid, device=device)
    g.set_slowness(m['slowness'].reshape(dim, order='F').flatten())
    return g


class Grid3d_d(_Grid3d):
    """double-precision 3-D grid (ttcrpy Grid3d_d)"""
    _dtype = np.float64
    builder = classmethod(_builder3d)


class Grid3d_f(_Grid3d):
    """single-precision 3-D grid (ttcrpy Grid3d_f)"""
    _dtype = np.float32
    builder = classmethod(_builder3d)


def _rebuild3d(cls, x, y, z, p):
    return cls(x, y, z, *p)


def Grid3d(x, y, z, n_threads=1, cell_slowness=1, method='FSM', tt_from_rp=1, interp_vel=0, eps=1.e-5, maxit=50,
           weno=1, nsnx=5, nsny=5, nsnz=5, n_secondary=2, n_tertiary=2, radius_factor_tertiary=3.0,
           translate_grid=False, fsm_gpu=False, dtype=np.float64, device=-1):
    """Grid3d(x, y, z, ..., dtype=np.float64) -> Grid3d_d or Grid3d_f  (rgrid.pyx:5580-5620)"""
    if np.dtype(dtype) == np.dtype(np.float64):
        cls = Grid3d_d
    elif np.dtype(dtype) == np.dtype(np.float32):
        cls = Grid3d_f
    else:
        raise ValueError('dtype must be np.float32 or np.float64, got {}'.format(dtype))
    return cls(x, y, z, n_threads, cell_slowness, method, tt_from_rp, interp_vel, eps, maxit, weno, nsnx, nsny, nsnz,
               n_secondary, n_tertiary, radius_factor_tertiary, translate_grid, fsm_gpu, device=device)


# ======================================================================================= 2-D
class _Grid2d(_GridBase):
    """2-D rectilinear grid, fast-sweeping method on MI355X.

    Grid2d_x(x, z, n_threads=1, cell_slowness=1, method='SPM', aniso='iso', eps=1.e-5, maxit=50, weno=1,
             rotated_template=0, nsnx=10, nsnz=10, n_secondary=3, n_tertiary=3,
             radius_factor_tertiary=3.0, tt_from_rp=0, fsm_gpu=False)
    (same signature as ttcrpy; only method='FSM', aniso='iso' is on this backend's path).
    """
    _ndim = 2

    def __init__(self, x, z, n_threads=1, cell_slowness=1, method='SPM', aniso='iso', eps=1.e-5, maxit=50, weno=1,
                 rotated_template=0, nsnx=10, nsnz=10, n_secondary=3, n_tertiary=3, radius_factor_tertiary=3.0,
                 tt_from_rp=0, fsm_gpu=False, device=-1):
        super().__init__()
        dt = self._dtype
        x = np.ascontiguousarray(x, dtype=dt).ravel()
        z = np.ascontiguousarray(z, dtype=dt).ravel()
        if x.size < 2 or z.size < 2:
            raise ValueError('x and z need at least two nodes')
        self._x, self._z = x, z
        self._dx = float(x[1] - x[0])
        self._dz = float(z[1] - z[0])
        self.cell_slowness = bool(cell_slowness)
        self._n_threads = int(n_threads)
        self.eps = float(eps)
        self.maxit = int(maxit)
        self.weno = bool(weno)
        self.rotated_template = bool(rotated_template)
        self.nsnx, self.nsnz = int(nsnx), int(nsnz)
        self.n_secondary, self.n_tertiary = int(n_secondary), int(n_tertiary)
        self.radius_factor_tertiary = float(radius_factor_tertiary)
        self.tt_from_rp = bool(tt_from_rp)
        self.fsm_gpu = bool(fsm_gpu)
        self.method = method
        self.aniso = aniso
        self._device = _device_arg(device)
        if method in ('SPM', 'DSPM'):
            raise NotImplementedError("method '%s' is outside the MI355X FSM path (SURVEY.md section 8)" % method)
        if method != 'FSM':
            raise ValueError('Method {0:s} undefined'.format(method))
        if aniso != 'iso':
            raise NotImplementedError('Anisotropic raytracing implemented only for SPM')
        self._lib = _lib.load()
        args = (C.byref(self._h), _lib.TTCR_F32 if dt == np.float32 else _lib.TTCR_F64,
                int(self.cell_slowness), x.size - 1, z.size - 1, self._dx, self._dz,
                float(x[0]), float(z[0]), self.eps, self.maxit, int(self.weno),
                int(self.rotated_template), self._n_threads)
        if isinstance(self._device, tuple):
            devs = (C.c_int * len(self._device))(*self._device)
            st = self._lib.ttcr_fsm2d_create_multi(*args, devs, len(self._device))
        else:
            st = self._lib.ttcr_fsm2d_create(*args, self._device)
        _lib.check(st)
        if self.tt_from_rp:   # Grid2Drn::getTraveltimeFromRaypath (ttcr/Grid2Drn.h:1478-1661)
            self.set_option("tt_from_rp", 1)

    def __reduce__(self):
        params = (self.n_threads, self.cell_slowness, self.method, self.aniso, self.eps, self.maxit, self.weno,
                  self.rotated_template, self.nsnx, self.nsnz, self.n_secondary, self.n_tertiary,
                  self.radius_factor_tertiary, self.tt_from_rp, self.fsm_gpu)
        return (_rebuild2d, (type(self), self.x, self.z, params))

    @property
    def x(self):
        return np.array(self._x)

    @property
    def z(self):
        return np.array(self._z)

    @property
    def dx(self):
        return self._dx

    @property
    def dz(self):
        return self._dz

    @property
    def shape(self):
        if self.cell_slowness:
            return (self._x.size - 1, self._z.size - 1)
        return (self._x.size, self._z.size)

    @property
    def nparams(self):
        return int(np.prod(self.shape))

    def get_number_of_nodes(self):
        return self._x.size * self._z.size

    def get_number_of_cells(self):
        return (self._x.size - 1) * (self._z.size - 1)

    def ind(self, i, j):
        return i * self._z.size + j

    def indc(self, i, j):
        return i * (self._z.size - 1) + j

    def is_outside(self, pts):
        pts = np.asarray(pts)
        return bool(np.min(pts[:, 0]) < self._x[0] or np.max(pts[:, 0]) > self._x[-1] or
                    np.min(pts[:, 1]) < self._z[0] or np.max(pts[:, 1]) > self._z[-1])

    def compute_D(self, coord):
        """compute_D(coord) -> csr_matrix (npts x nparams) of interpolation weights for velocity data points (rgrid.pyx:3565-3627)"""
        from . import inversion as _inv
        coord = np.asarray(coord, dtype=np.float64)
        if self.is_outside(coord):
            raise ValueError('Velocity data point outside grid')
        return _inv.interp_matrix((self._x, self._z), coord, self.cell_slowness)

    def compute_K(self, order=1):
        """compute_K(order=1) -> (Kx, Kz): first- or second-derivative smoothing matrices on the parameters (rgrid.pyx:3630-3733)"""
        from . import inversion as _inv
        return _inv.smoothing_matrices(self.shape, (self.dx, self.dz), order=order)

    def _save_raypaths(self, rays, filename):
        """rays (list of npts x 2 arrays, x and z) as a vtkPolyData of poly-lines in the plane y = 0 (rgrid.pyx:4228-4256)"""
        from . import io as _io
        _io.write_vtp_lines(filename, rays)

    @staticmethod
    def data_kernel_straight_rays(Tx, Rx, grx, grz, aniso=False):
        """data_kernel_straight_rays(Tx, Rx, grx, grz, aniso=False) -> L: straight rays through the cells of the grid with node
        coordinates grx, grz; tt = L @ slowness, or the x / z components of the segments in two blocks of columns for an elliptically
        anisotropic medium (rgrid.pyx:4259-4470)"""
        from . import inversion as _inv
        return _inv.straight_ray_kernel(Tx, Rx, (np.asarray(grx, dtype=np.float64), np.asarray(grz, dtype=np.float64)), aniso=aniso)

    def get_grid_traveltimes(self, thread_no=0):
        """traveltimes at the grid nodes, shape (nx, nz)  (rgrid.pyx:3102-3127)"""
        return self._flat_tt(thread_no).reshape((self._x.size, self._z.size))

    def get_slowness(self):
        n = self.get_number_of_nodes()
        out = np.empty(n, dtype=self._dtype)
        _lib.check(self._lib.ttcr_fsm_get_slowness(self._h, _ptr(out), n))
        return out.reshape((self._x.size, self._z.size))

    def _to_flat(self, a, what):
        nx, nz = self.shape
        a = np.asarray(a)
        if a.size != nx * nz:
            raise ValueError('%s vector has wrong size' % what)
        if a.ndim == 2:
            if a.shape != (nx, nz):
                raise ValueError('%s has wrong shape' % what)
            return a.flatten()
        elif a.ndim == 1:
            return a
        raise ValueError('%s must be 1D or 2D ndarray' % what)

    def set_slowness(self, slowness):
        """Assign slowness, shape (nx, nz) or flattened in C order (rgrid.pyx:3171-3205)"""
        s = np.ascontiguousarray(self._to_flat(slowness, 'Slowness'), dtype=self._dtype)
        _lib.check(self._lib.ttcr_fsm_set_slowness(self._h, _ptr(s), s.size))

    def set_velocity(self, velocity):
        s = np.ascontiguousarray(1. / self._to_flat(velocity, 'velocity'), dtype=self._dtype)
        _lib.check(self._lib.ttcr_fsm_set_slowness(self._h, _ptr(s), s.size))

    def raytrace(self, source, rcv, slowness=None, xi=None, theta=None, Vp0=None, Vs0=None, delta=None,
                 epsilon=None, gamma=None, thread_no=None, aggregate_src=False, compute_L=False, return_rays=False):
        """raytrace(source, rcv, slowness=None, ..., thread_no=None, aggregate_src=False) -> tt
        (rgrid.pyx:3804-4143): source has 2 (x,z) or 3 (t0,x,z) columns."""
        source = np.asarray(source)
        rcv = np.asarray(rcv)
        if source.ndim != 2 or rcv.ndim != 2:
            raise ValueError('source and rcv should be 2D arrays')
        if any(v is not None for v in (xi, theta, Vp0, Vs0, delta, epsilon, gamma)):
            raise NotImplementedError('Anisotropic raytracing implemented only for SPM')
        if compute_L and not self.cell_slowness:
            raise NotImplementedError('compute_L defined only for grids with slowness defined for cells')
        vTx, vt0, vRx, iRx = self._split_sources(source, rcv, aggregate_src)
        if slowness is not None:
            self.set_slowness(slowness)
        if compute_L:
            return self._run_l(vTx, vt0, vRx, iRx, rcv.shape[0], return_rays)
        if not return_rays:
            return self._run(vTx, vt0, vRx, iRx, rcv.shape[0], thread_no)
        # -> (tt, rays): Grid2D::raytrace(Tx,t0,Rx,tt,r_data,threadNo) -> Grid2Drn::getRaypath (ttcr/Grid2Drn.h:1663-1850)
        self.set_option("return_rays", 1)
        try:
            return self._run(vTx, vt0, vRx, iRx, rcv.shape[0], thread_no, return_rays=True)
        finally:
            self.set_option("return_rays", 0)


    def _run_l(self, vTx, vt0, vRx, iRx, n_rcv, return_rays):
        """compute_L=True (rgrid.pyx:3996-4010, :4043-4143): per event the overload with l_data (and r_data), one scipy CSR matrix
        (receivers of the event x cells) per event with the entries in the order the reference's loops emit them (ascending cell,
        entries of one cell in the order of the sorted l_data), stacked, then rows taken as `tmp[itmp, :]` like the reference does."""
        import scipy.sparse as sp

        dt = self._dtype
        tt = np.zeros((n_rcv,), dtype=dt)
        rays = [[0.0] for _ in range(n_rcv)]
        L = []
        NN = self.get_number_of_cells()
        if len(vTx) > 1:
            # several events: ONE call -- batched solves, the walks follow each batch (ttcr_fsm_raytrace_multi_l)
            tx = np.ascontiguousarray(np.vstack(vTx), dtype=dt).reshape(-1, 2)
            t0 = np.ascontiguousarray(np.concatenate(vt0), dtype=dt)
            rx = np.ascontiguousarray(np.vstack(vRx), dtype=dt).reshape(-1, 2)
            tx_off = np.zeros(len(vTx) + 1, dtype=np.int32); rx_off = np.zeros(len(vTx) + 1, dtype=np.int32)
            tx_off[1:] = np.cumsum([len(t) for t in vTx]); rx_off[1:] = np.cumsum([len(r) for r in vRx])
            out = np.empty(max(rx.shape[0], 1), dtype=dt)
            _lib.check(self._lib.ttcr_fsm_raytrace_multi_l(self._h, len(vTx), _ptr(tx_off), _ptr(tx), _ptr(t0), _ptr(rx_off), _ptr(rx),
                                                           _ptr(out), int(bool(return_rays))))
            nrow, nnz = C.c_size_t(0), C.c_size_t(0)
            _lib.check(self._lib.ttcr_fsm_multi_l_size(self._h, C.byref(nrow), C.byref(nnz)))
            off = np.zeros(nrow.value + 1, dtype=np.int64)
            jj = np.empty(max(nnz.value, 1), dtype=np.int64)
            vv = np.empty(max(nnz.value, 1), dtype=dt)
            _lib.check(self._lib.ttcr_fsm_get_multi_l(self._h, _ptr(off), _ptr(jj), _ptr(vv)))
            tmp = sp.csr_matrix((vv[:nnz.value].astype(np.float64), jj[:nnz.value], off), shape=(rx.shape[0], NN))
            if return_rays:
                nr, npnt = C.c_size_t(0), C.c_size_t(0)
                _lib.check(self._lib.ttcr_fsm_rays_size(self._h, C.byref(nr), C.byref(npnt)))
                roff = np.zeros(nr.value + 1, dtype=np.int64)
                pts = np.empty((max(npnt.value, 1), 2), dtype=dt)
                _lib.check(self._lib.ttcr_fsm_get_rays(self._h, _ptr(roff), _ptr(pts)))
            for n in range(len(vTx)):
                tt[iRx[n]] = out[rx_off[n]:rx_off[n + 1]]
                if return_rays:
                    for k, row in enumerate(iRx[n]):
                        rays[row] = np.array(pts[roff[rx_off[n] + k]:roff[rx_off[n] + k + 1]], dtype=np.float64)
            itmp = [row for n in range(len(vTx)) for row in iRx[n]]
            Lm = tmp[itmp, :]
            return (tt, rays, Lm) if return_rays else (tt, Lm)
        for n in range(len(vTx)):
            slot = n % self._n_threads
            tx = np.ascontiguousarray(vTx[n], dtype=dt)
            t0 = np.ascontiguousarray(vt0[n], dtype=dt)
            rx = np.ascontiguousarray(vRx[n], dtype=dt).reshape(-1, 2)
            out = np.empty(max(rx.shape[0], 1), dtype=dt)
            _lib.check(self._lib.ttcr_fsm_raytrace_l(self._h, slot, tx.shape[0], _ptr(tx), _ptr(t0), rx.shape[0], _ptr(rx), _ptr(out),
                                                     1 if return_rays else 0))
            tt[iRx[n]] = out[:rx.shape[0]]
            nrow, nnz = C.c_size_t(0), C.c_size_t(0)
            _lib.check(self._lib.ttcr_fsm_slot_l_size(self._h, slot, C.byref(nrow), C.byref(nnz)))
            off = np.zeros(nrow.value + 1, dtype=np.int64)
            jj = np.empty(max(nnz.value, 1), dtype=np.int64)
            vv = np.empty(max(nnz.value, 1), dtype=dt)
            _lib.check(self._lib.ttcr_fsm_get_slot_l(self._h, slot, _ptr(off), _ptr(jj), _ptr(vv)))
            L.append(sp.csr_matrix((vv[:nnz.value].astype(np.float64), jj[:nnz.value], off), shape=(rx.shape[0], NN)))
            if return_rays:
                nr, npnt = C.c_size_t(0), C.c_size_t(0)
                _lib.check(self._lib.ttcr_fsm_slot_rays_size(self._h, slot, C.byref(nr), C.byref(npnt)))
                roff = np.zeros(nr.value + 1, dtype=np.int64)
                pts = np.empty((max(npnt.value, 1), 2), dtype=dt)
                _lib.check(self._lib.ttcr_fsm_get_slot_rays(self._h, slot, _ptr(roff), _ptr(pts)))
                for k, row in enumerate(iRx[n]):
                    rays[row] = np.array(pts[roff[k]:roff[k + 1]], dtype=np.float64)
        tmp = sp.vstack(L).tocsr()
        itmp = [row for n in range(len(vTx)) for row in iRx[n]]
        Lm = tmp[itmp, :]
        if return_rays:
            return tt, rays, Lm
        return tt, Lm


class Grid2d_d(_Grid2d):
    _dtype = np.float64


class Grid2d_f(_Grid2d):
    _dtype = np.float32


def _rebuild2d(cls, x, z, p):
    return cls(x, z, *p)


def Grid2d(x, z, n_threads=1, cell_slowness=1, method='SPM', aniso='iso', eps=1.e-5, maxit=50, weno=1,
           rotated_template=0, nsnx=10, nsnz=10, n_secondary=3, n_tertiary=3, radius_factor_tertiary=3.0,
           tt_from_rp=0, fsm_gpu=False, dtype=np.float64, device=-1):
    """Grid2d(x, z, ..., dtype=np.float64) -> Grid2d_d or Grid2d_f  (rgrid.pyx:5646-5687)"""
    if np.dtype(dtype) == np.dtype(np.float64):
        cls = Grid2d_d
    elif np.dtype(dtype) == np.dtype(np.float32):
        cls = Grid2d_f
    else:
        raise ValueError('dtype must be np.float32 or np.float64, got {}'.format(dtype))
    return cls(x, z, n_threads, cell_slowness, method, aniso, eps, maxit, weno, rotated_template, nsnx, nsnz,
               n_secondary, n_tertiary, radius_factor_tertiary, tt_from_rp, fsm_gpu, device=device)


Grid3d.builder = Grid3d_d.builder
Grid3d.data_kernel_straight_rays = Grid3d_d.data_kernel_straight_rays   # (rgrid.pyx:5624)


Grid2d.data_kernel_straight_rays = Grid2d_d.data_kernel_straight_rays   # (rgrid.pyx:5690)
