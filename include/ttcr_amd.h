/* include/ttcr_amd.h -- C ABI of the MI355X-native fast-sweeping (FSM) eikonal solver.
 *
 * This is the drop-in boundary for the rectilinear FSM path of groupeLIAMG/ttcr:
 * every entry point replaces one member of the C++ interface that ttcrpy's Cython
 * layer programs against (src/ttcrpy/rgrid.pxd:30-145 -> ttcr/Grid3D.h, ttcr/Grid2D.h
 * and the Grid{2,3}Dr{n,c}fs leaves).  Paths below are relative to the reference root.
 * Plain pointers and sizes only; no C++/torch types; no exceptions cross the boundary
 * (integer status + ttcr_fsm_last_error()).
 *
 * Conventions (identical to the reference):
 *   - 3-D flat arrays are x-fastest:  n = (k*(ncy+1)+j)*(ncx+1)+i   (ttcr/Grid3Drn.h:2823)
 *     3-D cell arrays:                c = (k*ncy+j)*ncx+i            (ttcr/Grid3Drcfs.h:96-171)
 *   - 2-D flat arrays are z-fastest:  n = i*(ncz+1)+j                (ttcr/Grid2Drn.h:720)
 *     2-D cell arrays:                c = i*ncz+j                    (ttcr/Grid2Drcfs.h:113-137)
 *   - nc* are CELL counts (nodes = cells+1), as in the reference constructors.
 *   - `dtype` selects the reference instantiation: TTCR_F32 <-> <float,uint32_t>,
 *     TTCR_F64 <-> <double,uint32_t>; all `const void*` arrays hold that type.
 *   - a "slot" is the reference's threadNo: a private traveltime field kept on the
 *     device until the next solve in the same slot (ttcr/Node3Dn.h:109-110).
 *   - threads: every entry point that takes a grid locks that grid (one stream, one captured launch
 *     sequence and the pinned / scratch buffers are shared by its slots), so calls on ONE handle from several
 *     host threads are safe.  Single-source ttcr_fsm_raytrace calls that arrive together on a grid with several
 *     slots -- the way Grid3D's multi-source overload reaches a backend: nt host threads, each with its own
 *     threadNo (ttcr/Grid3D.h:810-853) -- are gathered for a short window (option "combine_window_us", default 200)
 *     and solved as ONE device batch, side by side like a ttcr_fsm_raytrace_multi call; every caller gets its own
 *     status and message.  Other calls on one handle run one after the other.  Different handles are independent.
 *     ttcr_fsm_last_error() is per host thread.
 */
#ifndef TTCR_AMD_H
#define TTCR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ttcr_fsm_grid ttcr_fsm_grid; /* opaque */

enum { TTCR_F32 = 0, TTCR_F64 = 1 };

/* status codes; the Python layer maps them to the exceptions the reference raises */
enum {
    TTCR_OK = 0,
    TTCR_ERR_VALUE = 1,       /* bad argument (Python ValueError in rgrid.pyx)            */
    TTCR_ERR_RUNTIME = 2,     /* std::runtime_error / length_error / logic_error -> RuntimeError */
    TTCR_ERR_DEVICE = 3,      /* HIP failure (no silent CPU fallback: the call fails)     */
    TTCR_ERR_UNSUPPORTED = 4  /* feature of the reference interface not built yet         */
};

/* Number of visible HIP devices (0 when none / runtime missing). */
int ttcr_fsm_device_count(void);

/* Message of the last failing call on this thread (also valid when create fails). */
const char* ttcr_fsm_last_error(void);

/* Replaces: Grid3Drnfs<T,uint32_t>::Grid3Drnfs  (ttcr/Grid3Drnfs.h:39-50)  [cell_slowness = 0]
 *           Grid3Drcfs<T,uint32_t>::Grid3Drcfs  (ttcr/Grid3Drcfs.h:41-52)  [cell_slowness = 1]
 * as constructed in src/ttcrpy/rgrid.pyx:217-224, 254-261 (and the _f twins :1917-1954).
 * eps is the per-node tolerance (scaled by the node count like the ctor does, :49),
 * n_slots is the reference's `nt`, device is the HIP device ordinal (-1: current). */
int ttcr_fsm3d_create(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncy,
                      uint32_t ncz, double dx, double xmin, double ymin, double zmin, double eps,
                      int maxit, int weno, int n_slots, int translate_origin, int device);

/* Replaces: Grid2Drnfs<T,uint32_t,sxz<T>>::Grid2Drnfs (ttcr/Grid2Drnfs.h:84-95)
 *           Grid2Drcfs<...>::Grid2Drcfs               (ttcr/Grid2Drcfs.h:45-58)
 * as constructed in src/ttcrpy/rgrid.pyx:2962-2966.  rotated_template: Grid2Drn::sweep45
 * (ttcr/Grid2Drn.h:756-794) after every sweep of the first-order solver when dx == dz and weno is
 * off (ttcr/Grid2Drnfs.h:277-286); ignored otherwise, like the reference does. */
int ttcr_fsm2d_create(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncz,
                      double dx, double dz, double xmin, double zmin, double eps, int maxit, int weno,
                      int rotated_template, int n_slots, int device);

/* The same grids on SEVERAL devices of one node -- what Grid3D's multi-source overload does with host threads
 * (ttcr/Grid3D.h:810-853; the reference's OpenCL backend keeps one solver per thread slot in one process,
 * ttcr/Grid3Drnfs_OpenCL.h:172-193): one replica of the grid per entry of `devices` (the same ordinal may appear more
 * than once), the slowness replicated, the n_slots traveltime slots divided over the replicas (slot s lives on replica
 * s / ceil(n_slots / n_devices)), the sources of ttcr_fsm_raytrace_multi block-distributed over ALL slots like
 * get_blk_size (ttcr/Grid3D.h:451-465) and every replica driven by its own host thread: no exchange between devices
 * during a solve, results identical to the one-device grid.  Every other entry point takes such a handle unchanged
 * (a slot argument is routed to the replica that owns the slot).  ttcr_fsm3d_create / ttcr_fsm2d_create with
 * device = -1 do the same when the environment variable TTCR_AMD_DEVICES holds a comma-separated device list, so an
 * unmodified caller can be spread over the GPUs of a node.  ttcr_fsm_n_devices: replicas behind a handle (1: plain). */
int ttcr_fsm3d_create_multi(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncy,
                            uint32_t ncz, double dx, double xmin, double ymin, double zmin, double eps,
                            int maxit, int weno, int n_slots, int translate_origin, const int* devices, int n_devices);
int ttcr_fsm2d_create_multi(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncz,
                            double dx, double dz, double xmin, double zmin, double eps, int maxit, int weno,
                            int rotated_template, int n_slots, const int* devices, int n_devices);
int ttcr_fsm_n_devices(const ttcr_fsm_grid* g);

/* Replaces: `del self.grid` (src/ttcrpy/rgrid.pyx:284-285). */
void ttcr_fsm_destroy(ttcr_fsm_grid* g);

/* Replaces: Grid3Drn::setSlowness (ttcr/Grid3Drn.h:82-89), Grid3Drcfs::setSlowness
 * (ttcr/Grid3Drcfs.h:88-171), Grid2Drn::setSlowness (ttcr/Grid2Drn.h:71-78),
 * Grid2Drcfs::setSlowness (ttcr/Grid2Drcfs.h:98-138).  n must equal the node count
 * (node grids) or the cell count (cell grids), else TTCR_ERR_RUNTIME with the
 * reference's message.  `s` is a host pointer; the _device variant takes a pointer
 * that is already resident in HBM on the grid's device (no PCIe copy). */
int ttcr_fsm_set_slowness(ttcr_fsm_grid* g, const void* s, size_t n);
int ttcr_fsm_set_slowness_device(ttcr_fsm_grid* g, const void* d_s, size_t n);
/* The same for a 3-D model that lies in host memory as an (nx, ny, nz) array in C order (z fastest) -- what
 * the Python classes receive (src/ttcrpy/rgrid.pyx:532-569 flattens it to x-fastest on the host before
 * Grid3D::setSlowness): uploaded as it lies, permuted on the device.  2-D grids: identical to
 * ttcr_fsm_set_slowness (their flat order is C order already). */
int ttcr_fsm_set_slowness_c_order(ttcr_fsm_grid* g, const void* s, size_t n);

/* Replaces: Grid3Drn::getSlowness (ttcr/Grid3Drn.h:90-97): NODE slowness, n = node count. */
int ttcr_fsm_get_slowness(ttcr_fsm_grid* g, void* out, size_t n);

/* Replaces: Grid3D::raytrace(Tx,t0,Rx,traveltimes,threadNo) (ttcr/Grid3D.h:470-502) ->
 * Grid3Drnfs::raytrace (ttcr/Grid3Drnfs.h:84-155) with tt_from_rp = false, and the 2-D
 * twin Grid2D::raytrace -> Grid2Drnfs::raytrace (ttcr/Grid2Drnfs.h:198-299).
 * tx: n_tx points (3 or 2 coordinates each) forming ONE source, t0: n_tx origin times,
 * rx: n_rx receivers, tt_out: n_rx traveltimes (Grid3Drn::getTraveltime, :794-930).
 * Points outside the grid -> TTCR_ERR_RUNTIME "Error: Point (...) outside grid." */
int ttcr_fsm_raytrace(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx,
                      const void* rx, void* tt_out);

/* Replaces: the multi-source overload Grid3D::raytrace(vector<vector<sxyz>>&...)
 * (ttcr/Grid3D.h:810-853).  Source n owns tx/t0 rows [tx_off[n], tx_off[n+1]) and rx /
 * tt_out rows [rx_off[n], rx_off[n+1]).  Sources are block-distributed over the slots
 * like get_blk_size (ttcr/Grid3D.h:451-465) and solved concurrently on the device. */
int ttcr_fsm_raytrace_multi(ttcr_fsm_grid* g, int n_src, const int* tx_off, const void* tx,
                            const void* t0, const int* rx_off, const void* rx, void* tt_out);

/* Replaces: Grid3Drn::getTT(tt, threadNo) (ttcr/Grid3Drn.h:102-108). n = node count. */
int ttcr_fsm_get_tt(ttcr_fsm_grid* g, int slot, void* out, size_t n);
/* The same field for a consumer on the device (no reference equivalent; the reference only has the host copy above).
 * ttcr_fsm_get_tt_device: pointer to n_nodes CONTIGUOUS values of slot `slot` in the flat order above.  Where every
 *   slot has its own field (one slot; 2-D grids; weno = 1) this is the field itself; on a first-order 3-D grid whose
 *   batches are big enough for source pairs to pay (n_slots x 16x16 column patches of a sweep > 6 144: e.g. 8 slots
 *   at 512^3, 32 at 256^3) the fields of two slots are interleaved in HBM, so the call de-interleaves into a scratch buffer of
 *   the grid: the pointer stays valid until the next ttcr_fsm_get_tt / ttcr_fsm_get_tt_device call or the
 *   destruction of the grid.
 * ttcr_fsm_get_tt_device_view: zero copy.  *d_ptr addresses node 0 of the slot's field where it lies and *stride is
 *   the distance between consecutive nodes in ELEMENTS of the grid dtype -- 1, or 2 for the interleaved layout
 *   T[p/2][node][2], p = the storage the slot's field occupies (p = slot with option "pair_sources" = 0; otherwise a
 *   multi-source call may have given the sources of its batch to each other's storage): always take pointer and stride
 *   from the call; valid until the next raytrace call that solves a source for that slot.  Both stay owned by the grid. */
int ttcr_fsm_get_tt_device(ttcr_fsm_grid* g, int slot, void** d_ptr);
int ttcr_fsm_get_tt_device_view(ttcr_fsm_grid* g, int slot, void** d_ptr, size_t* stride);

/* Replaces: Grid3Drn::getTraveltime(pt, nt) (ttcr/Grid3Drn.h:794-930) / 2-D (:359-414). */
int ttcr_fsm_interp(ttcr_fsm_grid* g, int slot, int n_pts, const void* pts, void* tt_out);

/* Replaces: Grid3Drn::computeSlowness(pt, isTranslated) (ttcr/Grid3Drn.h:2451-2676) and Grid2Drn::computeSlowness(pt)
 * (ttcr/Grid2Drn.h:262-330), what get_s0 of the Python classes calls per source point (src/ttcrpy/rgrid.pyx:824,
 * :3799): the node slowness interpolated at n_pts points (velocity is interpolated instead when the grid was
 * made with interp_vel, 3-D).  translated != 0: the points are already relative to the origin of a grid built with
 * translate_origin.  A point outside the grid -> TTCR_ERR_RUNTIME (the reference reads its arrays unchecked). */
int ttcr_fsm_compute_slowness(ttcr_fsm_grid* g, int n_pts, const void* pts, int translated, void* out);

/* Replaces: get_niter()/get_niterw() (ttcr/Grid3Drnfs.h:56-57); per slot here
 * (the reference keeps one racy value per grid). */
int ttcr_fsm_get_niter(ttcr_fsm_grid* g, int slot, int* niter, int* niterw);
/* The quantity the stopping rule of the driver loop compared with eps * N (ttcr/Grid3Drnfs.h:141-152: `change`), for
 * every sweep-iteration of the last solve of `slot`: first_order[0..n_first) and weno[0..n_weno) (entries beyond the
 * iterations that ran are 0).  Here it is the fp64 sum of the decreases of all nodes over the iteration's sweeps (equal to
 * the reference's sum of |T_old - T_new| in exact arithmetic; the reference adds it up sequentially in T1). */
int ttcr_fsm_get_changes(ttcr_fsm_grid* g, int slot, double* first_order, int n_first, double* weno, int n_weno);
/* The same iterations as the REFERENCE summed them (ttcr/Grid3Drnfs.h:141-152: sequentially, in node order, in T1), for the iterations
 * that were decided with that sum itself (options "stopping_rule", "stopping_shortcuts"); NaN for the others.  A value below eps * N is what `change >= epsilon`
 * saw (:153), bit for bit; one at or above it may have been cut short there (the sum only grows: the decision is taken). */
int ttcr_fsm_get_reference_changes(ttcr_fsm_grid* g, int slot, double* first_order, int n_first, double* weno, int n_weno);

/* Replaces: getNthreads() (ttcr/Grid3D.h) and the node/cell counts of rgrid.pyx:386-404. */
int ttcr_fsm_n_slots(const ttcr_fsm_grid* g);
size_t ttcr_fsm_n_nodes(const ttcr_fsm_grid* g);
size_t ttcr_fsm_n_cells(const ttcr_fsm_grid* g);

/* Tuning / measurement knobs (no reference equivalent):
 *   "fixed_iters"  > 0: run exactly that many sweep-iterations, ignore eps
 *   "max_batch"    sources swept concurrently by one launch sequence (default: n_slots)
 *   "use_graph"    1: the tile-per-launch driver (mode 0) replays its launch sequence from a hipGraph, the persistent drivers
 *                  launch their kernels directly (default); 0: no graphs; 2: graphs for every driver
 *   "stopping_rule" 1 (default): the reference ends a solve when `change`, the SEQUENTIAL sum in T1, in node order, of abs(times[n] -
 *                  T[n]) falls below eps * N (ttcr/Grid3Drnfs.h:141-152; 2-D: ttcr/Grid2Drnfs.h:265-290).  The sweep kernels
 *                  accumulate the same quantity as an fp64 sum of decreases; at 1.3e8 fp32 nodes the reference's sum reads 1-3 %
 *                  low near the threshold, so an iteration whose fp64 change lies within [1/2, 16] x eps * N (fp64 grids: 1e-6
 *                  either side) is decided by the reference's own sum, computed on the device from a snapshot of the field taken
 *                  before the iteration -- exactly, and in parallel (fsm_refsum_* in fsm_kernels.h: while the running sum stays in
 *                  one binade an addition is an integer increment that depends on the sum only through its parity; blocks of
 *                  elements are summarised for either parity and composed in order).  The snapshot is taken whenever the iteration
 *                  before came within 1e3 windows of the threshold, before the first WENO iteration, and always on grids of up to
 *                  2^24 nodes; an iteration that lands in the window without one is decided by the fp64 sum and counted
 *                  (ttcr_fsm_stopping_stats).  0: the fp64 sum alone.  2: as 1 with the sum as ONE chain of additions (the
 *                  round-4 kernel: 0.5 s per 512^3 field; kept as the checker of the parallel form).  tests/test_stopping_rule_gpu.py
 *                  The sum runs over the non-zero terms alone, in node order (zeros add nothing; in the iterations the rule decides
 *                  all but 1e-4 ... 5e-2 of the nodes did not change): one pass writes the terms of the fields that were asked for
 *                  and brings the snapshot up to the current field, a scan and a second pass compact them.
 *   "stopping_shortcuts"  bits (default 3).  1: on 3-D grids whose sweep kernels keep dirty-brick stamps (exact skipping on), those passes
 *                  and the snapshots read only the 16^3 bricks that changed since the snapshot was last right.  2: the ordered pass is left
 *                  out where it cannot change the decision: with M non-zero terms the sequential T1 sum lies within M u / (1 - M u) of the
 *                  exact sum (u the unit roundoff of T1), which the fp64 sum of decreases gives to 2 (nx + ny + nz) u; `change >= epsilon`
 *                  is then decided as the reference decides it, and ttcr_fsm_get_reference_changes has no value (NaN) for that
 *                  iteration.  0: whole fields, every sum (tests, bisecting).  tests/test_stopping_rule_gpu.py
 *   "lone_chunk"   levels per chunk of the 3-D sweep kernels that keep ONE field per workgroup, where the longer chunk is faster (16, default):
 *                  fp32 first-order sweeps of lone sources and of batches below the pairing threshold (512^3: lone source 6.7 instead of
 *                  7.2 ms per sweep-iteration, 4 sources 9.6 instead of 10.6; 256^3 x 4 sources 3.5 instead of 4.3), fp64 first-order
 *                  sweeps of up to four sources (256^3: 7.1 instead of 8.1 ms for one, 9.1 instead of 10.4 for four), the WENO stage of
 *                  one or two sources (256^3: 319 instead of 381 ms per solve, fp64 537 instead of 664).  8: the chunk length of every
 *                  other kernel, everywhere.  The partial order of the node updates and therefore every result is the same; exact
 *                  skipping steps over chunks of that length.  env TTCR_FSM_LONE_CHUNK.  tests/test_lone_chunk_gpu.py,
 *                  profiles/r05/experiment_chunk_length.txt
 *   "arith"        0 (default): the reference's arithmetic -- every result bit-identical to the reference.  1: tolerance-grade local solvers
 *                  in the first-order sweeps of fp32 grids (Grid3Drn::update_node / Grid2Drn::update_node with dx == dz, ttcr/Grid3Drn.h:
 *                  2936-2956, ttcr/Grid2Drn.h:945-950): the same quadratics evaluated in fp32 on differences from the smallest neighbour
 *                  scaled by 1/(s dx), one v_sqrt_f32 per update instead of two correctly rounded fp64 roots (update3_fast / update2_fast,
 *                  fsm_kernels.h).  NOT bit-identical: a result differs from the reference's by an ulp of the traveltime here and there
 *                  (512^3 gradient model: RMS 8e-7 s on traveltimes up to 13 s, north_star's bound is 1e-5 s; iteration counts the same on
 *                  every model tested).  What it buys: a lone 512^3 source 6.6 -> 5.4 ms per sweep-iteration, 64 sources 87 -> 73 ms,
 *                  8 sources 17.2 -> 14.2 ms (profiles/r06).  Whole-iteration launches only ("mode" 2); fp64 grids, the rotated template
 *                  and 2-D grids with dx != dz keep the reference's arithmetic.  Grids WITH the WENO stage (weno = 1, ttcrpy's default)
 *                  keep it in both stages under arith = 1: the WENO iteration amplifies an ulp of difference in the first-order field to
 *                  1e-3 s (the exact WENO stage behind a tolerance-grade first-order stage ends 4e-5 s RMS, 4e-3 s max from the reference:
 *                  profiles/r06/weno_sensitivity.txt) -- only bit-identical input reproduces the reference there.  2: as 1, and both
 *                  stages of grids with the WENO stage as well (weno_axis_fast, solve3_literal_fast): OUTSIDE the 1e-5 s bound (RMS up to
 *                  5e-5 s, max 4e-3 s at 128^3 - 256^3 -- the size of the WENO stage's own response to rounding), for 256^3 solves of
 *                  240 instead of 319 ms (one source) and 423 instead of 762 ms (eight).  env TTCR_FSM_ARITH.  tests/test_arith_mode_gpu.py
 *   "prefill"      a second set of traveltime fields: while a call that restarted every slot runs, a low-priority side stream fills the
 *                  other set with max() (the reference's reinit, ttcr/Grid3Drnfs.h:92-94), and the next call that restarts every
 *                  slot swaps the sets instead of writing n_slots x n_nodes values in front of its first sweep (512^3 x 64: 32 GB,
 *                  6.8 ms).  Calls that restart some slots fill those in place.  1 on, 0 off, -1 (default): on when the fields take
 *                  >= 64 MiB and twice that is at most half the device memory.  env TTCR_FSM_PREFILL.  tests/test_prefill_gpu.py
 *   "pair_sources" 1 (default): where the grid keeps its fields in pairs (n_slots x patches of a sweep > 6 144, env
 *                  TTCR_FSM_PAIR_UNITS; TTCR_FSM_PAIR = 1 / 0 forces / forbids the pair layout at grid creation) the sources
 *                  of a batch are paired by distance before they share a workgroup two by two
 *   "pair_layout"  grids whose slots x patches lie between the pairing threshold and 2.5 x that (512^3 nodes: 8 ... 15 slots) change their
 *                  field layout between calls that restart every slot: pairs while the call before evaluated more than 0.55 of its node
 *                  updates (rough models: 311 against 360 ms per solve of 8 sources), one field per workgroup on 16-level chunks once it
 *                  evaluated less (smooth models: 31.5 against 35.5 ms per step), back to pairs above 0.70.  -1 (default): that rule;
 *                  0 / 1: one field per workgroup / pairs, always.  Results do not depend on the layout.  tests/test_pair_layout_gpu.py
 *   "combine_window_us"  single-source calls from several host threads wait this long for one another before they go
 *                  to the device as one batch (default 200; 0: every call on its own)
 *   "mode"         2: persistent sweep kernel, ONE launch per sweep-iteration: patches ordered by
 *                     progress counters in HBM, the next directional sweep starts on the patches the
 *                     previous one has finished (default); 1: same kernel, one launch per directional
 *                     sweep; 0: one launch per tile wavefront.  All three give bit-identical fields.
 *                  (env TTCR_FSM_MODE overrides the default at grid creation)
 *   "tt_from_rp"   1: receiver traveltimes are integrated along the ray traced back through the
 *                     traveltime field (replaces setTraveltimeFromRaypath(bool), ttcr/Grid3D.h, and the
 *                     `ttrp` constructor argument; Grid3Drn::getTraveltimeFromRaypath, ttcr/Grid3Drn.h:
 *                     1103-1243; 2-D: Grid2Drn::getTraveltimeFromRaypath, ttcr/Grid2Drn.h:1478-1661);
 *                     0: tri/bilinear interpolation (getTraveltime).  Default 0.
 *   "interp_vel"   1: the ray integration interpolates velocity instead of slowness (`intVel`
 *                     constructor argument / processVel, ttcr/Grid3Drn.h:2451-2676).  Default 0.
 *   "pair_sources" 1 (default): the sources of a ttcr_fsm_raytrace_multi batch are paired by distance before they share
 *                     the sweep kernel two by two (first-order 3-D grids): the field of the source the block distribution
 *                     gives to thread t may then LIVE in another slot's storage, every entry point that takes a slot
 *                     looks it up, and a caller sees what it would see without (same fields, iteration counts, receivers
 *                     under the same thread numbers).  0: every source in the storage of its own thread number.
 *   "return_rays"  1: the raytrace calls follow the overloads with r_data (Grid3D::raytrace(Tx,t0,Rx,tt,
 *                     r_data,threadNo), ttcr/Grid3D.h:546-586): receiver traveltimes AND raypaths come from
 *                     Grid3Drn::getRaypath(Tx,t0,Rx,r_data,tt,threadNo) (ttcr/Grid3Drn.h:1339-1500); the rays
 *                     stay in the grid until the next raytrace call, see ttcr_fsm_get_rays (2-D: Grid2Drn::
 *                     getRaypath, ttcr/Grid2Drn.h:1663-1850, points are (x, z) pairs).
 *   "skip"         1: the persistent kernels step over chunks, work units and whole sweeps that cannot change a node:
 *                     a chunk whose read set (bricks of 16^3 nodes stamped with the sweep of their last change; the
 *                     change flags its upwind patches publish with their progress) holds no change since its nodes
 *                     were last visited is a no-op, and so is every sweep that follows a sweep without a change --
 *                     exact, fields and iteration counts are unchanged; 0: evaluate every chunk; -1 (default): on
 *                     where it was measured to pay (2-D grids, the WENO stage, first-order 3-D batches with at least 2048 work
 *                     units per sweep); fp32 first-order 3-D launches of 1024 ... 2047 units (a lone 512^3 source) follow the model:
 *                     skipping while the grid's last skipping solve evaluated less than 0.6 of its node updates (smooth models: 13.3 ->
 *                     12.8 ms per solve), everything evaluated above that (rough models lose 4 - 9 % with it), tried again after eight
 *                     solves; smaller launches never skip.  DESIGN.md 4a, profiles/r06/lone_skip_models.txt */
int ttcr_fsm_set_option(ttcr_fsm_grid* g, const char* key, double value);

/* Replaces: the r_data output of the raytrace overloads above (std::vector<std::vector<sxyz<T1>>>&,
 * src/ttcrpy/rgrid.pxd:60-75).  After a raytrace call with option "return_rays" = 1: one ray per receiver
 * row of that call, in row order; ray n is points [offsets[n], offsets[n+1]) of pts (x,y,z triples -- x,z pairs in 2-D -- of the
 * grid's dtype), from the receiver to the source.  Two calls: sizes first, then the copy into caller
 * buffers of n_rays+1 offsets and 3*n_points (2-D: 2*n_points) coordinates. */
int ttcr_fsm_rays_size(const ttcr_fsm_grid* g, size_t* n_rays, size_t* n_points);
int ttcr_fsm_get_rays(const ttcr_fsm_grid* g, long long* offsets, void* pts);

/* Replaces: Grid3D::raytrace(Tx, t0, Rx, traveltimes, r_data, threadNo) (ttcr/Grid3D.h:546-586; 2-D: ttcr/Grid2D.h) as
 * Grid3D's multi-source r_data overload calls it -- from nt host threads at once, each with its own threadNo
 * (ttcr/Grid3D.h:855-905).  One call solves the source in `slot` and keeps traveltimes AND rays, whatever the
 * "return_rays" option says; the rays are stored PER SLOT, so a thread finds the rays of its own call with
 * ttcr_fsm_slot_rays_size / ttcr_fsm_get_slot_rays (same layout as ttcr_fsm_get_rays) whatever other threads do on other
 * slots meanwhile.  They stay until the next ttcr_fsm_raytrace_rays call on that slot. */
int ttcr_fsm_raytrace_rays(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx,
                           const void* rx, void* tt_out);
int ttcr_fsm_slot_rays_size(const ttcr_fsm_grid* g, int slot, size_t* n_rays, size_t* n_points);
int ttcr_fsm_get_slot_rays(const ttcr_fsm_grid* g, int slot, long long* offsets, void* pts);

/* Replaces: Grid3D::raytrace(Tx, t0, Rx, traveltimes, m_data, threadNo) (ttcr/Grid3D.h:743-772) -> Grid3Drn::getRaypath(Tx,
 * t0, Rx, m_data, RxNo, tt, threadNo) (ttcr/Grid3Drn.h:1503-1800): what `compute_M=True` of the Python layer reaches
 * (src/ttcrpy/rgrid.pyx:1059, :1171-1191).  One call solves the source in `slot`, walks every receiver's ray on the device
 * with the walk of THAT overload (a kernel of its own) and assembles, per receiver, the (node index, value) entries of the
 * matrix of traveltime derivatives in the order the reference pushes them; traveltimes are those of that overload
 * (integrated along the ray; 0 for a receiver on the source).  The reference's formula is restated AS IT STANDS: it
 * overwrites prev_pt with curr_pt before it forms a segment's mid-point and length (:1590-1597), so every step of the walk
 * contributes signed zeros at the eight nodes around the step's end point and only the last hop (or two) to each source point
 * within a cell diagonal carries weight; its weights drop xmin and its node indices may lie one node past the grid (the
 * Python layer drops those).  Bit-identical to the compiled reference (tests/golden/m_golden.npz, tests/test_m_matrix.py),
 * sources of several points -- closely spaced ones included -- as well.  3-D node grids (TTCR_ERR_UNSUPPORTED otherwise, like
 * the Python layer).
 * ttcr_fsm_raytrace_rm replaces the overload that keeps the rays too, Grid3D::raytrace(Tx, t0, Rx, traveltimes, r_data, m_data,
 * threadNo) (ttcr/Grid3D.h:646-680) -> Grid3Drn::getRaypath(Tx, t0, Rx, r_data, m_data, RxNo, tt, threadNo) (ttcr/Grid3Drn.h:
 * 2144-2470), what `compute_M=True, return_rays=True` reaches (rgrid.pyx:1050).  NOT the same matrix: this overload takes
 * prev_pt before it pushes the point, so its segments carry their lengths (one exception, restated as well: the plane point
 * between the walk and a source point, :2359-2366).  The rays of the call: ttcr_fsm_slot_rays_size / ttcr_fsm_get_slot_rays.
 * ttcr_fsm_slot_m_size: rows (= receivers) and entries of the last call on `slot`; ttcr_fsm_get_slot_m: row_off[n_rows+1],
 * j[nnz] node indices, v[nnz] values of the grid dtype. */
int ttcr_fsm_raytrace_m(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx,
                        const void* rx, void* tt_out);
int ttcr_fsm_raytrace_rm(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx,
                         const void* rx, void* tt_out);
int ttcr_fsm_slot_m_size(const ttcr_fsm_grid* g, int slot, size_t* n_rows, size_t* nnz);
int ttcr_fsm_get_slot_m(const ttcr_fsm_grid* g, int slot, long long* row_off, long long* j, void* v);
/* Replaces: the multi-source overloads with m_data, Grid3D::raytrace(Tx[], t0[], Rx[], traveltimes[], [r_data[],] m_data[])
 * (ttcr/Grid3D.h:896-1000), which run the single-source overload per source on host threads -- what ttcrpy reaches with
 * compute_M and several events (src/ttcrpy/rgrid.pyx:1096-1102).  Sources and receivers laid out like ttcr_fsm_raytrace_multi;
 * the fields are solved in batches of n_slots sources (one sweep launch per batch and sweep-iteration instead of one chain of
 * launches per source), the walks follow each batch.  with_rays != 0: the overload with r_data and m_data for every source
 * (rays: ttcr_fsm_rays_size / ttcr_fsm_get_rays).  Results identical to n_src calls of ttcr_fsm_raytrace_m / _rm.
 * ttcr_fsm_multi_m_size / ttcr_fsm_get_multi_m: ONE CSR over all receiver rows of the call (row r = receiver row r). */
int ttcr_fsm_raytrace_multi_m(ttcr_fsm_grid* g, int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off,
                              const void* rx, void* tt_out, int with_rays);
int ttcr_fsm_multi_m_size(const ttcr_fsm_grid* g, size_t* n_rows, size_t* nnz);
int ttcr_fsm_get_multi_m(const ttcr_fsm_grid* g, long long* row_off, long long* j, void* v);

/* Replaces: Grid2D::raytrace(Tx, t0, Rx, traveltimes, l_data, threadNo) (ttcr/Grid2D.h:616-640) and the overload with r_data
 * AND l_data (:583-614) -> Grid2Drn::getRaypath(Tx, t0, Rx, [r_data,] l_data, tt, threadNo) (ttcr/Grid2Drn.h:1852-2190): what
 * `compute_L=True` of the Python layer reaches for 2-D grids with cell slowness (src/ttcrpy/rgrid.pyx:3889-3893, :4060-4143).
 * One call solves the source in `slot`, walks every receiver's ray on the device and keeps, per receiver, the (cell index,
 * segment length) entries of the ray-projection matrix L, sorted by cell with the reference's comparator (CompareSiv_i,
 * ttcr/ttcr_t.h:417-422).  Restated AS IT STANDS: this walk gives up where the plain raypath walk retries along a face
 * ("going outside grid"); when the last two segments of a ray lie in one cell the reference pushes the entry of the first
 * AND an entry holding the sum, and the overload without r_data prices the last hop with that sum.  Traveltimes are those
 * of the overload (with_rays selects which).  Bit-identical to the compiled reference (tests/golden/l_golden.npz,
 * tests/test_l_matrix.py).  2-D grids with cell slowness only (TTCR_ERR_UNSUPPORTED otherwise: in 3-D ttcrpy itself raises
 * "compute_L defined for the FSM", rgrid.pyx:916-917; node grids: rgrid.pyx:3889-3890).
 * ttcr_fsm_slot_l_size: rows (= receivers) and entries of the last call on `slot`; ttcr_fsm_get_slot_l: row_off[n_rows+1],
 * cell[nnz] cell indices (x-major, z fastest, like Grid2Drn::getCellNo), v[nnz] lengths of the grid dtype.  With with_rays != 0
 * the rays of the call are available through ttcr_fsm_slot_rays_size / ttcr_fsm_get_slot_rays (two coordinates per point). */
int ttcr_fsm_raytrace_l(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx,
                        const void* rx, void* tt_out, int with_rays);
int ttcr_fsm_slot_l_size(const ttcr_fsm_grid* g, int slot, size_t* n_rows, size_t* nnz);
int ttcr_fsm_get_slot_l(const ttcr_fsm_grid* g, int slot, long long* row_off, long long* cell, void* v);
/* The same for every source of a call -- Grid2D's multi-source overloads with l_data (they run the single-source overload per source
 * on host threads), what ttcrpy reaches with compute_L and several events.  Sources and receivers laid out like
 * ttcr_fsm_raytrace_multi; batched solves, the walks follow each batch; results identical to n_src calls of ttcr_fsm_raytrace_l.
 * ONE CSR over all receiver rows of the call (ttcr_fsm_multi_l_size / ttcr_fsm_get_multi_l); with_rays != 0: the rays of the call
 * through ttcr_fsm_rays_size / ttcr_fsm_get_rays. */
int ttcr_fsm_raytrace_multi_l(ttcr_fsm_grid* g, int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off,
                              const void* rx, void* tt_out, int with_rays);
int ttcr_fsm_multi_l_size(const ttcr_fsm_grid* g, size_t* n_rows, size_t* nnz);
int ttcr_fsm_get_multi_l(const ttcr_fsm_grid* g, long long* row_off, long long* cell, void* v);

typedef struct {
    double sweep_ms;        /* HIP-event time of all sweep launches of the last raytrace call   */
    double total_ms;        /* wall time of the last raytrace call (host clock, incl. copies)  */
    long long kernel_launches; /* sweep-tile kernel launches in the last call                  */
    long long node_updates;    /* nodes * 8 (or 4) * iterations, summed over sources           */
    long long evaluated_updates; /* node updates actually evaluated (chunks whose inputs did not
                                    change since their last evaluation are skipped, exactly)   */
    int iterations;         /* max sweep-iterations over the sources of the last call          */
    int n_sources;
} ttcr_fsm_timing;
int ttcr_fsm_last_timing(const ttcr_fsm_grid* g, ttcr_fsm_timing* out);
/* Decisions of the stopping rule since the grid was created: iterations decided by the reference's sequential sum, iterations
 * that landed in the window without a snapshot (decided by the fp64 sum), rounds of the parallel sum.  Any pointer may be NULL. */
int ttcr_fsm_stopping_stats(const ttcr_fsm_grid* g, long long* reference_sums, long long* reference_sums_missed, long long* rounds);
/* Calls (solve batches) that found their traveltime fields initialised by the side stream instead of filling them (option "prefill";
 * the fill it replaces: `reinit` of every node, ttcr/Grid3Drnfs.h:92-94).  No reference equivalent: tests and bench.py read it. */
int ttcr_fsm_prefill_swaps(const ttcr_fsm_grid* g, long long* swaps);
/* The reference's `change` (ttcr/Grid3Drnfs.h:141-152) of two fields given on the host, n_nodes values of the grid's type each, node
 * order: the sum, in node order, in T1, of abs(times[n] - field[n]), into *out (a T1).  parallel != 0: the parallel form the solver
 * uses; 0: one chain of additions.  Both exact; tests compare them. */
int ttcr_fsm_reference_change(ttcr_fsm_grid* g, const void* times, const void* field, int parallel, void* out);
/* Name of the sweep-kernel instantiation the last solve launched (first-order stage of the last batch; no reference
 * equivalent: bench.py reports it beside the roofline figures).  Written into buf (n bytes, NUL terminated). */
int ttcr_fsm_last_kernel(const ttcr_fsm_grid* g, char* buf, size_t n);
/* Build provenance: the 16-hex-digit hash of the kernel sources and compiler flags this library was compiled from (ttcr_amd/build.py,
 * source_hash(), baked in at compile time).  The Python layer refuses a library whose id is not the hash of the sources beside it, and
 * bench.py / the tests print it: a measured binary is provably the committed code.  "unknown" for a build that did not pass the id. */
const char* ttcr_fsm_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* TTCR_AMD_H */
