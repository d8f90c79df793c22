// ttcr_amd/csrc/fsm_capi.hip -- host side of the MI355X FSM solver + the C ABI of
// include/ttcr_amd.h.  Mirrors the *interface* of the reference's Grid3D/Grid2D FSM leaves
// (ttcr/Grid3Drnfs.h, Grid3Drcfs.h, Grid2Drnfs.h, Grid2Drcfs.h) -- constructor arguments,
// setSlowness/getTT/raytrace semantics, error messages -- on top of the kernels in
// fsm_kernels.h.  There is no CPU fallback: without a HIP device every call fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ttcr_amd.h"
#include "fsm_kernels.h"
#include "fsm_fast_api.h"

#ifndef FSM_CHUNK3
#define FSM_CHUNK3 8
#endif

namespace ttcr_amd {

static thread_local std::string g_last_error;

struct ValueError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct DeviceError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct Unsupported : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            std::ostringstream _m;                                                               \
            _m << "HIP error " << hipGetErrorString(_e) << " at " << __FILE__ << ":" << __LINE__ \
               << " (" #expr ")";                                                                \
            throw DeviceError(_m.str());                                                         \
        }                                                                                        \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t guard = 0;   // elements allocated in front of p and behind p + n (readable, never meaningful)
    void reserve(size_t m, size_t guard_ = 0) {
        if (m <= n && guard_ <= guard) return;
        release();
        char* raw = nullptr;
        HIP_CHECK(hipMalloc((void**)&raw, (m + 2 * guard_) * sizeof(T)));
        guard = guard_;
        p = (T*)raw + guard;
        n = m;
    }
    void release() {
        if (p) (void)hipFree((void*)(p - guard));
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

struct Timing {
    double sweep_ms = 0, total_ms = 0;
    long long launches = 0, node_updates = 0, evaluated_updates = 0;
    int iterations = 0, n_sources = 0;
};

class GridBase {
   public:
    virtual ~GridBase() {}
    virtual void set_slowness(const void* s, size_t n, bool on_device, bool c_order = false) = 0;
    virtual void get_slowness(void* out, size_t n) = 0;
    // explicit_slots (optional): source n is solved in slot explicit_slots[n] (ascending, distinct) -- the combined
    // single-source calls of several host threads; otherwise the sources are block-distributed like get_blk_size
    // force_rays: keep the rays of this call whatever the "return_rays" option says
    virtual void raytrace_multi(int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off,
                                const void* rx, void* tt_out, int forced_slot, const int* explicit_slots = nullptr,
                                bool force_rays = false) = 0;
    // one source in `slot`, traveltimes AND rays, the rays kept per slot (raytrace overloads with r_data called by several
    // host threads at once, ttcr/Grid3D.h:855-905: every thread finds its own rays whatever the others do meanwhile)
    virtual void raytrace_rays(int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx, void* tt_out) = 0;
    virtual void slot_rays_size(int slot, size_t* n_rays, size_t* n_points) const = 0;
    virtual void get_slot_rays(int slot, long long* offsets, void* pts) const = 0;
    // the same with the entries of the matrix M per receiver (m_data overloads), kept per slot like the rays
    // both: the overload that keeps the rays as well (its terms differ, see fsm_raypath3d_m)
    virtual void raytrace_m(int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx, void* tt_out, bool both) = 0;
    virtual void slot_m_size(int slot, size_t* n_rows, size_t* nnz) const = 0;
    virtual void get_slot_m(int slot, long long* row_off, long long* j, void* v) const = 0;
    // every source of a call at once (batched solves, then the walks); one CSR over all receiver rows of the call
    virtual void raytrace_multi_m(int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off, const void* rx,
                                  void* tt_out, bool both) = 0;
    virtual void multi_m_size(size_t* n_rows, size_t* nnz) const = 0;
    virtual void get_multi_m(long long* row_off, long long* j, void* v) const = 0;
    // the raytrace overloads with l_data (2-D cell grids): ray-projection matrix L, one CSR row per receiver
    virtual void raytrace_l(int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx, void* tt_out, bool with_rays) = 0;
    virtual void slot_l_size(int slot, size_t* n_rows, size_t* nnz) const = 0;
    virtual void get_slot_l(int slot, long long* row_off, long long* cell, void* v) const = 0;
    virtual void raytrace_multi_l(int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off, const void* rx,
                                  void* tt_out, bool with_rays) = 0;
    virtual void multi_l_size(size_t* n_rows, size_t* nnz) const = 0;
    virtual void get_multi_l(long long* row_off, long long* cell, void* v) const = 0;
    virtual void validate_points(int n_tx, const void* tx, int n_rx, const void* rx) = 0;   // throws like raytrace would
    // single-source calls that arrive together (ttcrpy's thread pool: nt host threads, one slot each) are solved together
    struct Request {
        int slot, n_tx, n_rx;
        const void *tx, *t0, *rx;
        void* tt;
        int status = TTCR_OK;
        std::string err;
        bool done = false;
    };
    std::mutex q_mu;
    std::condition_variable q_cv;
    std::vector<Request*> queue;
    bool leader_active = false;
    bool had_company = false;      // two calls have been seen on this handle at the same time (guarded by q_mu)
    std::atomic<int> combine_window_us{200};   // how long a call waits for company (option "combine_window_us"; 0: never)
    size_t elem_size = 4;
    virtual void get_tt(int slot, void* out, size_t n) = 0;
    virtual void* tt_device(int slot) = 0;                      // contiguous copy of the field (see ttcr_amd.h)
    virtual void* tt_device_view(int slot, size_t* stride) = 0;  // the field where it lies + its element stride
    std::mutex mu;  // one call at a time per handle: stream, graph capture, pinned and scratch buffers are shared
    virtual void interp(int slot, int n, const void* pts, void* out) = 0;
    virtual void compute_slowness(int n, const void* pts, bool translated, void* out) = 0;
    virtual void rays_size(size_t* n_rays, size_t* n_points) const = 0;
    virtual void get_rays(long long* offsets, void* pts) const = 0;
    int dim = 3, dtype = 0, n_slots = 1, device = 0;
    size_t n_nodes = 0, n_cells = 0;
    std::vector<int> niter, niterw;
    // Where the field of (logical) slot s lives: phys[s].  A slot is what the caller's thread number names; the batch driver
    // may give the sources of a call to other physical slots than the block distribution names (it pairs sources that lie
    // close to each other, see raytrace_multi) and records here where each one went.  Always a permutation; everything
    // inside the grid (solve_batch, the per-slot device arrays, niter, change_hist) is indexed by PHYSICAL slot, every entry
    // point that takes a slot translates.
    std::vector<int> phys;
    int P(int slot) const { return phys.empty() ? slot : phys[slot]; }
    int stopping_shortcuts = 3; // option "stopping_shortcuts", bits: 1 the passes of the stopping rule over a field and its snapshot read only the bricks
                                // the sweep kernels stamped as changed, where such stamps are kept; 2 the ordered sum is left out where bounds on
                                // it already put it on one side of eps * N (GridT::decide_go_on).  0: whole fields, every sum (tests, bisecting)
    int stopping_rule = 1;      // option "stopping_rule": 1 (default) the reference's sequential T1 sum decides wherever it could differ from the
                                // fp64 sum of decreases, 0 the fp64 sum alone (default: the sequential sum is 1.3e8 dependent additions
                                // per 512^3 field and iteration it is asked for -- 11.7 s instead of 0.32 s for the heterogeneous bench leg)
    long long reference_sums = 0, reference_sums_missed = 0;   // decisions taken with the reference's sum / that would have needed a snapshot
    long long refsum_rounds = 0;                               // rounds of the parallel form of that sum (fsm_refsum_*)
    virtual void stopping_stats(long long* sums, long long* missed, long long* rounds) const {
        if (sums) *sums = reference_sums;
        if (missed) *missed = reference_sums_missed;
        if (rounds) *rounds = refsum_rounds;
    }
    // the reference's `change` of two fields given on the host (n_nodes values each, node order): sum of abs(times[n] - field[n]) in T1,
    // in node order; parallel: the exact parallel form, else the one-chain kernel (tests compare the two)
    virtual void reference_change_host(const void* times, const void* field, bool parallel, void* out) = 0;
    int pair_by_distance = 1;   // option "pair_sources" (0: every source in the slot the block distribution names)
    virtual void set_pair_layout(int) {}
    virtual long long prefill_swap_count() const { return 0; }   // calls that took fields initialised on the side stream (ttcr_fsm_prefill_swaps)
    int lone_chunk = 16;   // option "lone_chunk" / TTCR_FSM_LONE_CHUNK: levels per chunk of the fp32 first-order 3-D kernels with one field per workgroup (8 or 16)
    int arith = 0;      // option "arith" / TTCR_FSM_ARITH: 0 (default) the reference's arithmetic, results bit-identical to it; 1 tolerance-grade
                        // fp32 local solvers in the first-order sweeps of fp32 grids without the WENO stage (update3_fast / update2_fast,
                        // fsm_kernels.h): within north_star's 1e-5 s RMS of the reference, NOT bit-identical; 2: the WENO stage too
                        // (weno_axis_fast; outside that bound, see fast_now); whole-iteration launches only
    int prefill = -1;   // option "prefill" / TTCR_FSM_PREFILL: a second set of traveltime fields, re-initialised on a side stream while a
                        // solve runs, which the next call that restarts EVERY slot swaps in instead of filling (GridT::solve_batch);
                        // 1 on, 0 off, -1 (default): on when the fields take at least 64 MiB and twice that is at most half the device memory
    // L1 change of every sweep-iteration of the last solve of a slot (what the stopping rule compared with eps * N), first-
    // order stage then WENO stage
    std::vector<std::vector<double>> change_hist, change_histw;
    // ... and the reference's own sum (sequential, in T1) for the iterations that were decided with it (option stopping_rule; NaN: the others)
    std::vector<std::vector<double>> ref_hist, ref_histw;
    virtual void get_reference_changes(int slot, double* first, int n_first, double* wen, int n_weno) const {
        if (slot < 0 || slot >= n_slots) throw ValueError("Thread number is larger than number of threads");
        slot = P(slot);
        for (int q = 0; q < n_first; ++q) first[q] = (!ref_hist.empty() && q < (int)ref_hist[slot].size()) ? ref_hist[slot][q] : std::nan("");
        for (int q = 0; q < n_weno; ++q) wen[q] = (!ref_histw.empty() && q < (int)ref_histw[slot].size()) ? ref_histw[slot][q] : std::nan("");
    }
    virtual void get_changes(int slot, double* first, int n_first, double* wen, int n_weno) const {
        if (slot < 0 || slot >= n_slots) throw ValueError("Thread number is larger than number of threads");
        slot = P(slot);
        for (int q = 0; q < n_first; ++q) first[q] = (!change_hist.empty() && q < (int)change_hist[slot].size()) ? change_hist[slot][q] : 0.0;
        for (int q = 0; q < n_weno; ++q) wen[q] = (!change_histw.empty() && q < (int)change_histw[slot].size()) ? change_histw[slot][q] : 0.0;
    }
    bool weno = false;
    bool sweep45_strips = false;  // force the strip kernel of sweep45 (env TTCR_FSM_SWEEP45=strips; tests)
    bool rotated = false;  // 2-D rotated_template: sweep45 after every first-order sweep (ttcr/Grid2Drnfs.h:277-286)
    int ttrp = 0, interp_vel = 0;  // traveltime from raypath (ttcr/Grid3D.h:493-496), processVel
    std::atomic<int> return_rays{0};   // raytrace overloads with r_data (ttcr/Grid3D.h:546-586): rays kept for get_rays
    int fixed_iters = 0, max_batch = 0, use_graph = 1;
    int skip = -1; // persistent kernel: 1 = skip chunks whose read set did not change (exact); 0 = evaluate
                   // every chunk; -1 (default) = on for a lone source with the WENO stage (256^3: 427 -> 368 ms), off otherwise
    int mode = 2;  // 2: persistent kernel, one launch per sweep-iteration, sweeps overlap (default);
                   // 1: persistent kernel, one launch per sweep; 0: one launch per tile wavefront
    Timing timing;
    std::string last_kernel;   // instantiation of the sweep kernel the last solve launched (ttcr_fsm_last_kernel)
    virtual std::string kernel_name() const { return last_kernel; }
    // ttcr_fsm_set_option (a multi-device grid forwards it to its replicas)
    virtual void apply_option(const std::string& k, double value) {
        if (k == "fixed_iters") fixed_iters = (int)value;
        else if (k == "max_batch") max_batch = (int)value;
        else if (k == "use_graph") use_graph = (int)value;
        else if (k == "combine_window_us") combine_window_us = (int)value;
        else if (k == "mode") mode = (int)value;
        else if (k == "skip") skip = (int)value;
        else if (k == "tt_from_rp") ttrp = value != 0;
        else if (k == "interp_vel") interp_vel = value != 0;
        else if (k == "return_rays") return_rays = value != 0;
        else if (k == "pair_sources") pair_by_distance = value != 0;
        else if (k == "pair_layout") {
            if (value != -1 && value != 0 && value != 1) throw ValueError("option 'pair_layout': -1 (default), 0 or 1");
            set_pair_layout((int)value);
        }
        else if (k == "stopping_rule") stopping_rule = (int)value;
        else if (k == "stopping_shortcuts") { if (value < 0 || value > 3 || value != (double)(int)value) throw ValueError("option stopping_shortcuts: 0 ... 3"); stopping_shortcuts = (int)value; }
        else if (k == "prefill") {
            if (value != -1 && value != 0 && value != 1) throw ValueError("option 'prefill': -1 (default), 0 or 1");
            prefill = (int)value;
        } else if (k == "lone_chunk") {
            if (value != 8 && value != 16) throw ValueError("option 'lone_chunk': 8 or 16");
            lone_chunk = (int)value;
        }
        else if (k == "arith") {
            if (value != 0 && value != 1 && value != 2)
                throw ValueError("option 'arith': 0 (the reference's arithmetic), 1 (tolerance-grade fp32 where it stays within 1e-5 s RMS) or 2 (the WENO stage as well)");
            arith = (int)value;
        }
        else throw ValueError("unknown option '" + k + "'");
    }
    virtual void get_niter(int slot, int* it, int* itw) const {
        if (slot < 0 || slot >= n_slots) throw ValueError("Thread number is larger than number of threads");
        if (it) *it = niter[P(slot)];
        if (itw) *itw = niterw[P(slot)];
    }
};

// tile shapes (threads = PJ*PK); see DESIGN.md section 4 for the LDS budget
template <typename T, int DIM> struct TileCfg;
#ifndef FSM_PJ3
#define FSM_PJ3 16
#endif
#ifndef FSM_PK3
#define FSM_PK3 16
#endif
template <> struct TileCfg<float, 3> { static constexpr int PJ = FSM_PJ3, PK = FSM_PK3, BL = 16; };
template <> struct TileCfg<double, 3> { static constexpr int PJ = 16, PK = 8, BL = 16; };
#ifndef FSM_PJ2
#define FSM_PJ2 64
#endif
#ifndef FSM_CHUNK2
#define FSM_CHUNK2 16
#endif
template <> struct TileCfg<float, 2> { static constexpr int PJ = FSM_PJ2, PK = 1, BL = 32; };
template <> struct TileCfg<double, 2> { static constexpr int PJ = FSM_PJ2, PK = 1, BL = 32; };
// chunk length of the persistent kernel
template <typename T, int DIM> struct ChunkCfg { static constexpr int C = DIM == 3 ? FSM_CHUNK3 : FSM_CHUNK2; };

template <typename T>
class GridT : public GridBase {
   public:
    // reference geometry, all in T like the reference members (ttcr/Grid3Drn.h:67-78)
    uint32_t ncx, ncy, ncz;  // cells; 2-D: ncy = 0
    T dx, dz, xmin, ymin, zmin, xmax, ymax, zmax, ox = 0, oy = 0, oz = 0;
    T epsilon;
    int nitermax;
    bool cell, translate;
    bool have_slowness = false;

    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevBuf<T> d_s, d_cells, d_tt, d_rx, d_out;
    // Re-initialisation off the path of a call (reinit, ttcr/Grid3Drnfs.h:92-94: T = max() everywhere -- 32 GB of stores for 64 fields
    // of 512^3 nodes, 6.8 ms at the write bandwidth of the device).  d_tt_alt is a second set of fields: while a solve that restarted
    // every slot runs in d_tt, a low-priority side stream fills d_tt_alt; the next call that restarts every slot swaps the two and
    // finds its fields initialised.  A call that restarts only some slots fills them in place as before (the other slots keep their
    // fields, as the reference's per-thread grids do).  Nothing outside the grid sees the swap except device views, which are valid
    // "until the next raytrace call that solves a source for that slot" (include/ttcr_amd.h).
    DevBuf<T> d_tt_alt;
    bool alt_filled = false;        // a fill of d_tt_alt has been issued on fill_stream (ev_fill marks its end)
    hipStream_t fill_stream = nullptr;
    hipEvent_t ev_fill = nullptr;
    long long prefill_swaps = 0;    // calls that found their fields initialised
    long long prefill_swap_count() const override { return prefill_swaps; }
    void set_pair_layout(int v) override { pair_layout = v; }
    DevBuf<int> d_rslot;
    DevBuf<RaySrc> d_rdesc;
    int weno_ch4_min = 8;  // pair layout: slot groups from which the 3-D WENO stage uses chunks of 4 levels
    int weno_c16_below = 3;  // one field per workgroup: batch entries below which the 3-D WENO stage uses chunks of 16 levels (option lone_chunk = 16)
    int f64_c16_below = 5;   // ... below which the fp64 first-order 3-D sweeps do
    int pre_min = 2;           // 3-D: slot groups in a batch from which the upwind counters are sampled one chunk ahead (PRE)
    // extra (unused) dynamic LDS per workgroup of the whole-iteration launch: caps the resident workgroups per CU.  A lone
    // source is bound by the dependent chain of a marching unit, and a unit that shares its CU's SIMDs with another
    // marching unit runs that chain slower (TTCR_FSM_XS_LDS=bytes, TTCR_FSM_XS_LDS_BELOW=groups; 0: off)
    // Measured, 512^3: one source 8.49 -> 7.37 ms per sweep-iteration with two workgroups per CU instead of four (7.14 with
    // the counters sampled ahead as well), one source pair 10.5 -> 9.96; one per CU 9.95 (too few units in flight);
    // from two slot groups on, and for the 2-D and WENO kernels, the cap does not pay (profiles/r02/occupancy_cap.txt).
    size_t xs_lds_bytes = 40000;
    int xs_lds_below = 2;
    size_t xs_dyn_lds(int batch) const { return (dim == 3 && stage == 0 && batch < xs_lds_below) ? xs_lds_bytes : 0; }
    int time_order_below = 1 << 30; // fewer batch entries (slot groups / slots) than this in a batch: the whole-iteration launch hands its units out in the
                               // order of their expected start times instead of sweep by sweep (build_persistent_lists).  Until round 5: 17 -- with
                               // the kernels of round 5 the start-time order wins at every size measured (512^3, two sweep-iterations: 64 sources
                               // 169.8 -> 166.8 ms, 34 sources 99.6 -> 96.0, 48: 134.4 -> 133.8; rough model, 64 sources to convergence 2 004 ->
                               // 1 917 ms; 256^3 x 24 sources 8.17 -> 7.67 ms per sweep-iteration; 2-D 4096^2 x 64 18.5 -> 18.4;
                               // profiles/r05/experiment_ticket_order.txt)
    DevBuf<T> d_gather;      // scratch for de-interleaving one field
    DevBuf<T> d_rsrc, d_rt0;  // source points / origin times of the source whose rays are traced
    DevBuf<int> d_rstat;
    DevBuf<T> d_ssh;         // sheared copies of the node slowness, one per direction family
    size_t ssh_stride = 0;   // elements per copy: NK planes of shear_plane(geom) elements
    DevBuf<uint32_t> d_mask;
    DevBuf<int> d_bbox, d_slots, d_lmask;
    int* h_lmask = nullptr;  // pinned
    DevBuf<uint32_t> d_tiles;            // per launch w: the patches that have nodes in it
    std::vector<int> tile_off, tile_cnt;  // offsets / counts into d_tiles
    DevBuf<double> d_change;
    DevBuf<unsigned long long> d_prof;  // TTCR_FSM_PROF=1 debug phase timers
    size_t prof_words = 0;
    DevBuf<uint32_t> d_order;  // persistent kernel: patches in ticket order (anti-diagonal major)
    DevBuf<uint32_t> d_order_xs[2][2];  // whole-iteration launch, [stage: first order / WENO][0: sweep by sweep, 1: by expected start time]
    DevBuf<int> d_sync;        // persistent kernel: ticket, abort flag, per (source, patch) progress
    int n_patches = 0;
    int* h_abort = nullptr;    // pinned
    DevBuf<int> d_stamp;       // dirty-brick stamps [n_slots][nbf*nbj*nbk]
    DevBuf<unsigned long long> d_cmap;   // SKIP kernels: per-chunk edge-change flags of every unit of a launch [dir][entry][patch][2][cmap_words]
    int cmap_words = 1;
    DevBuf<unsigned long long> d_sw;   // SKIP kernels: per (global sweep, slot group) units finished | units that changed a node
    int sw_sweeps = 0;
    DevBuf<int> d_iter;        // current iteration index (read by the captured kernels)
    DevBuf<unsigned long long> d_evals;  // [n_slots] node updates actually evaluated
    int* h_iter = nullptr;     // pinned: [0] iteration index, [1] launch epoch
    unsigned launch_epoch = 1;  // number of the next sweep launch of this grid (0: never -- what zeroed memory reads as)
    bool sync_clean = true;     // the last solve ended normally (else the synchronisation words are wiped before the next one)
    size_t sync_words = 0;
    unsigned long long* h_evals = nullptr;  // pinned
    int nbf = 0, nbj = 0, nbk = 0;
    size_t n_bricks = 0;
    DevBuf<InitPoint<T>> d_pts;
    double* h_change = nullptr;  // pinned
    int* h_slots = nullptr;      // pinned
    size_t mask_words = 0;
    SweepGeom geom;
    int n_launch = 0;  // launches per sweep direction
    // one captured launch sequence per stage (first-order / WENO3)
    hipGraph_t graphs[2] = {nullptr, nullptr};
    hipGraphExec_t graph_execs[2] = {nullptr, nullptr};
    int graph_batches[2] = {0, 0}, graph_modes[2] = {-1, -1};

    GridT(int dim_, bool cell_, uint32_t nx, uint32_t ny, uint32_t nz, double ddx, double ddz, double minx,
          double miny, double minz, double eps, int maxit, int nslots, bool translate_, int dev, bool weno_) {
        dim = dim_;
        weno = weno_;
        dtype = sizeof(T) == 4 ? TTCR_F32 : TTCR_F64;
        elem_size = sizeof(T);
        cell = cell_;
        translate = translate_;
        ncx = nx; ncy = ny; ncz = nz;
        dx = (T)ddx;
        dz = (T)ddz;
        xmin = (T)minx; ymin = (T)miny; zmin = (T)minz;
        // xmax(minx+nx*ddx) etc. in T arithmetic (ttcr/Grid3Drn.h:73, ttcr/Grid2Drn.h:62)
        xmax = xmin + (T)nx * dx;
        if (dim == 3) {
            ymax = ymin + (T)ny * dx;
            zmax = zmin + (T)nz * dx;
        } else {
            ymax = ymin;
            zmax = zmin + (T)nz * dz;
        }
        if (translate) {  // buildGridNodes, ttcr/Grid3Drn.h:362-372
            ox = xmin; oy = ymin; oz = zmin;
            xmax -= xmin; ymax -= ymin; zmax -= zmin;
            xmin = 0; ymin = 0; zmin = 0;
        }
        n_nodes = dim == 3 ? (size_t)(nx + 1) * (ny + 1) * (nz + 1) : (size_t)(nx + 1) * (nz + 1);
        n_cells = dim == 3 ? (size_t)nx * ny * nz : (size_t)nx * nz;
        if (n_nodes >= (1ull << 32)) throw ValueError("grid too large: node index must fit uint32 (reference T2)");
        epsilon = (T)eps;
        epsilon *= (T)n_nodes;  // ttcr/Grid3Drnfs.h:49
        nitermax = maxit;
        n_slots = nslots;
        niter.assign(n_slots, 0);
        niterw.assign(n_slots, 0);
        phys.resize(n_slots);
        for (int q = 0; q < n_slots; ++q) phys[q] = q;
        max_batch = n_slots;
        device = dev;
        HIP_CHECK(hipSetDevice(device));
        {
            int cus = 0;
            HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
            persist_wgs = 8 * std::max(cus, 1);
        }
        HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreate(&ev0));
        HIP_CHECK(hipEventCreate(&ev1));
        d_s.reserve(n_nodes, 64);
        // source pairs (fields interleaved, marched together) for the first-order 3-D solver only.  The one-wave 2-D
        // patches issue in order, a second source doubles the instructions of a level and buys nothing (4096^2:
        // 16 sources 28.0 -> 20.8 ms, 64 sources 35.8 -> 28.9 ms, 256 sources 112 -> 87 ms unpaired); the WENO stage
        // is bound by its arithmetic, and pairs cost it a resident wave (256^3: 2 sources 761 -> 483 ms, 8 sources
        // 1208 -> 973 ms, 16 sources 1740 -> 1672 ms, 64 sources 5986 -> 6221 ms).  profiles/r02/pairing.txt
        // Round 4, with exact skipping on by default for batches: on a model where most chunks are skipped, pairs only pay once the
        // chip has more work units than it can keep in flight -- a chunk of a pair is evaluated when EITHER source needs it and a
        // pair's level is the longer chain; on a model where most chunks are evaluated they pay from 4 sources on.
        // 512^3 (1 024 patches), ms per sweep-iteration paired / unpaired, gradient model (66 % of the updates skipped):
        //   2 sources 10.1 / 8.1, 4: 12.0 / 10.6, 8: 17.3 / 16.2, 16: 27.2 / 30.2, 32: 45.9 / 58.4; 256^3: 8 sources 5.7 / 4.5, 16: 6.5 / 6.1
        // ms per solve, random 16^3-block model (8-11 iterations, 75-83 % evaluated):
        //   2 sources 109.9 / 110.5, 4: 191 / 203, 8: 314 / 372, 16: 556 / 710      (profiles/r04/README.md)
        // So: pairs when slots x patches of a sweep exceed pair_units_min = 6 144 (512^3: from 8 sources on, where the smooth model
        // loses 7 % and the rough one gains 18 %; below, the smooth model gains 13-25 % and the rough one loses 0-6 %).
        {
            long long pair_units_min = 6144;
            if (const char* e = std::getenv("TTCR_FSM_PAIR_UNITS")) pair_units_min = std::atoll(e);   // tuning only
            const long long patches = dim == 3 ? (long long)((ny + 1 + TileCfg<T, 3>::PJ - 1) / TileCfg<T, 3>::PJ) * ((nz + 1 + TileCfg<T, 3>::PK - 1) / TileCfg<T, 3>::PK) : 0;
            NS = (n_slots >= 2 && dim == 3 && !weno && (long long)n_slots * patches > pair_units_min) ? 2 : 1;
            // Round 6: between the threshold and 2.5 x the threshold (512^3: 8 ... 15 slots) which layout is faster depends on the MODEL --
            // smooth (most chunks skipped): one field per workgroup on 16-level chunks, 15.2 against 17.2 ms per sweep-iteration for 8 sources;
            // rough (most chunks evaluated): pairs, 311 against 360 ms per solve.  Both layouts take the same memory, so such a grid follows
            // the evaluated fraction of its last call that restarted every slot (choose_layout); option "pair_layout" pins it.
            ns_window = NS == 2 && (long long)n_slots * patches <= (5 * pair_units_min) / 2;
        }
        if (const char* e = std::getenv("TTCR_FSM_PAIR")) { NS = (std::atoi(e) != 0 && n_slots >= 2) ? 2 : 1; ns_window = false; }
        if (const char* e = std::getenv("TTCR_FSM_LAYOUT_LO")) layout_lo = std::atof(e);
        if (const char* e = std::getenv("TTCR_FSM_LAYOUT_HI")) layout_hi = std::atof(e);
        d_tt.reserve(n_nodes * (size_t)(n_slots + (n_slots & 1)), 64);   // (room for either layout)
        mask_words = (n_nodes + 31) / 32;
        d_mask.reserve(mask_words * (size_t)n_slots);
        d_bbox.reserve(6 * (size_t)n_slots);
        d_slots.reserve(n_slots);
        d_lmask.reserve(n_slots);
        HIP_CHECK(hipHostMalloc((void**)&h_lmask, sizeof(int) * n_slots));
        d_change.reserve(n_slots);
        if (std::getenv("TTCR_FSM_PROF")) {   // FSM_ENABLE_PROF builds: 8 phase sums + a 4-word trace entry per work unit
            prof_words = 8 + 4 * (size_t)(1 << 20);
            d_prof.reserve(prof_words);
            HIP_CHECK(hipMemset(d_prof.p, 0, prof_words * sizeof(unsigned long long)));
        }
        HIP_CHECK(hipHostMalloc((void**)&h_change, sizeof(double) * n_slots));
        HIP_CHECK(hipHostMalloc((void**)&h_slots, sizeof(int) * n_slots));
        HIP_CHECK(hipHostMalloc((void**)&h_abort, sizeof(int)));
        *h_abort = 0;
        HIP_CHECK(hipHostMalloc((void**)&h_iter, 2 * sizeof(int)));
        HIP_CHECK(hipHostMalloc((void**)&h_evals, sizeof(unsigned long long) * n_slots));
        HIP_CHECK(hipMemsetAsync(d_tt.p, 0, n_nodes * (size_t)(n_slots + (n_slots & 1)) * sizeof(T), stream));

        if (dim == 3) {
            using C = TileCfg<T, 3>;
            geom.NF = nx + 1; geom.NJ = ny + 1; geom.NK = nz + 1;
            geom.npj = (geom.NJ + C::PJ - 1) / C::PJ;
            geom.npk = (geom.NK + C::PK - 1) / C::PK;
            n_launch = count_launches(C::BL);
        } else {  // F = z (fastest), J = x
            using C = TileCfg<T, 2>;
            geom.NF = nz + 1; geom.NJ = nx + 1; geom.NK = 1;
            geom.npj = (geom.NJ + C::PJ - 1) / C::PJ;
            geom.npk = 1;
            n_launch = count_launches(C::BL);
        }
        geom.n_nodes = (uint32_t)n_nodes;
        geom.M = (std::max(geom.NF, geom.NJ) + 1) & ~1;
        geom.MP = geom.M + FSM_XPAD;
        geom.SR = dim == 3 ? ((geom.NJ + 15) / 16) * 32 : 0;   // 2-D: plain rows (fsm_kernels.h, shear_index)
        ssh_stride = (size_t)geom.NK * shear_plane(geom);
        // (the sweep kernels address a thread's plane of a copy with a 32-bit byte offset from the plane of its patch's first row)
        if ((double)(dim == 3 ? 16 : 1) * (double)shear_plane(geom) * sizeof(T) >= 4294967296.0)
            throw ValueError("grid too large for the slowness offsets of the sweep kernel (16 planes of max(nx, ny) x ny nodes must stay below 4 GiB)");
        d_ssh.reserve(ssh_stride * (dim == 3 ? 4 : 2));
        HIP_CHECK(hipMemsetAsync(d_ssh.p, 0, ssh_stride * (dim == 3 ? 4 : 2) * sizeof(T), stream));   // (entries without a node are read, never used)
        if (dim == 3) build_tile_lists(TileCfg<T, 3>::PJ, TileCfg<T, 3>::PK, TileCfg<T, 3>::BL);
        else build_tile_lists(TileCfg<T, 2>::PJ, TileCfg<T, 2>::PK, TileCfg<T, 2>::BL);
        build_persistent_lists();
        nbf = (geom.NF + FSM_BRICK - 1) / FSM_BRICK;
        nbj = (geom.NJ + FSM_BRICK - 1) / FSM_BRICK;
        nbk = (geom.NK + FSM_BRICK - 1) / FSM_BRICK;
        n_bricks = (size_t)nbf * nbj * nbk;
        d_stamp.reserve(n_bricks * n_slots);  // one set per slot group is used
        {
            // change maps of the SKIP kernels: one bit per chunk and edge; a unit has at most (NF + PJ + PK) / C + 2 chunks
            // (the shortest chunks any instantiation uses are 4 levels long), and reads a little past the end of its
            // upwind units' maps
            const int PJ = dim == 3 ? TileCfg<T, 3>::PJ : TileCfg<T, 2>::PJ, PK = dim == 3 ? TileCfg<T, 3>::PK : 1;
            const int max_chunks = (geom.NF + 2 * (PJ + PK)) / 4 + 4;
            cmap_words = max_chunks / 32 + 1;
            d_cmap.reserve((size_t)n_patches * n_slots * (dim == 3 ? 8 : 4) * 2 * cmap_words);
            HIP_CHECK(hipMemset(d_cmap.p, 0, (size_t)n_patches * n_slots * (dim == 3 ? 8 : 4) * 2 * cmap_words * sizeof(unsigned long long)));
            // whole-sweep tallies: one entry per sweep of a solve (both stages); solves with more sweeps simply stop using them
            sw_sweeps = (dim == 3 ? 8 : 4) * (2 * std::min(nitermax, 256) + 2);
            d_sw.reserve((size_t)sw_sweeps * n_slots);   // (one entry per slot group: at most n_slots of them, whichever the layout)
        }
        d_iter.reserve(2);
        d_evals.reserve(n_slots);
        if (dim == 2)   // the row-parallel sweep45 kernel keeps four rows in (dynamic) LDS
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fsm_sweep45_rows<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        if (const char* e = std::getenv("TTCR_FSM_SWEEP45")) sweep45_strips = std::string(e) == "strips";
        if (const char* e = std::getenv("TTCR_FSM_WENO_CH4_MIN")) weno_ch4_min = std::atoi(e);   // tuning only
        if (const char* e = std::getenv("TTCR_FSM_WENO_C16_BELOW")) weno_c16_below = std::atoi(e);   // tuning only
        if (const char* e = std::getenv("TTCR_FSM_F64_C16_BELOW")) f64_c16_below = std::atoi(e);     // tuning only
        if (const char* e = std::getenv("TTCR_FSM_TIME_ORDER_BELOW")) time_order_below = std::atoi(e);   // tuning only
        if (const char* e = std::getenv("TTCR_FSM_PRE_MIN")) pre_min = std::atoi(e);                     // tuning only
        if (const char* e = std::getenv("TTCR_FSM_SKIP")) skip = std::atoi(e);
        if (const char* e = std::getenv("TTCR_FSM_SKIP_UNITS")) skip_units_min = std::atoi(e);   // tuning only
        if (const char* e = std::getenv("TTCR_FSM_SKIP_PROBE_UNITS")) skip_probe_min = std::atoi(e);   // tuning only
        if (const char* e = std::getenv("TTCR_FSM_RS_FIELDS")) rs_fields_max = (size_t)std::max(0, std::atoi(e));   // tests: batches of the stopping rule's sums (0: strided fields)
        if (const char* e = std::getenv("TTCR_FSM_WGS")) persist_wgs = std::atoi(e);              // tuning only
        if (const char* e = std::getenv("TTCR_FSM_XS_LDS")) xs_lds_bytes = (size_t)std::atol(e);
        if (const char* e = std::getenv("TTCR_FSM_XS_LDS_BELOW")) xs_lds_below = std::atoi(e);
        if (const char* e = std::getenv("TTCR_FSM_MODE")) mode = std::atoi(e);
        if (const char* e = std::getenv("TTCR_FSM_PREFILL")) prefill = std::atoi(e);
        if (const char* e = std::getenv("TTCR_FSM_LONE_CHUNK")) lone_chunk = std::atoi(e) == 8 ? 8 : 16;   // tuning only
        if (const char* e = std::getenv("TTCR_FSM_ARITH")) arith = std::max(0, std::min(2, std::atoi(e)));
    }

    // persistent kernel: ticket order = anti-diagonal m = TJ+TK major (a topological order of the
    // patch dependencies), progress counters per (source slot in batch, patch)
    void build_persistent_lists() {
        std::vector<uint32_t> order;
        for (int m = 0; m <= geom.npj + geom.npk - 2; ++m)
            for (int TK = std::max(0, m - geom.npj + 1); TK <= std::min(m, geom.npk - 1); ++TK)
                order.push_back((uint32_t)(m - TK) | ((uint32_t)TK << 16));
        n_patches = (int)order.size();
        d_order.reserve(order.size());
        HIP_CHECK(hipMemcpy(d_order.p, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        // four ticket counters, the abort word, then one progress word per (direction, slot, patch).  The kernels tag every
        // word with the launch epoch: nothing is reset between the launches of a solve (fsm_kernels.h, "launch epoch")
        sync_words = 8 + (size_t)n_patches * n_slots * (dim == 3 ? 8 : 4);
        d_sync.reserve(sync_words);
        HIP_CHECK(hipMemset(d_sync.p, 0, sync_words * sizeof(int)));
        if (geom.npj >= (1 << 14) || geom.npk >= (1 << 14)) throw ValueError("grid too large for the patch index of the sweep kernel");
        for (int st = 0; st < 2; ++st)
            for (int tm = 0; tm < 2; ++tm) {
                const std::vector<uint32_t> xs = xs_order(order, tm != 0, st == 0 ? 1 : 2);
                d_order_xs[st][tm].reserve(xs.size());
                HIP_CHECK(hipMemcpy(d_order_xs[st][tm].p, xs.data(), xs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            }
    }

    // Ticket order of the whole-iteration launch: every unit (direction d, patch) once, entries TJ | TK << 14 | d << 28.
    // A unit waits for its two upwind patches of the same sweep and for the patches of sweep d-1 that own a column
    // within 2H of its own (fsm_sweep_persistent), so any order that puts those first is deadlock free whatever the
    // number of resident workgroups.  by_time = false: sweep by sweep, anti-diagonals inside a sweep.  by_time = true:
    // by the start time every unit would have on an unbounded machine (upwind patch + one hand-off; previous sweep's
    // patches finished), in chunk times.  A sweep has more patches than the chip has workgroup slots; handed out sweep
    // by sweep, the slots fill up with patches that sit out their turn on the ramp of the patch wavefront while the
    // next sweep, whose first corner is long free, cannot enter.  In start-time order the resident units are the ones
    // that can run next, of whichever sweep.
    std::vector<uint32_t> xs_order(const std::vector<uint32_t>& diag_order, bool by_time, int H) const {
        const int PJ = dim == 3 ? TileCfg<T, 3>::PJ : TileCfg<T, 2>::PJ, PK = dim == 3 ? TileCfg<T, 3>::PK : 1;
        const int C = dim == 3 ? ChunkCfg<T, 3>::C : ChunkCfg<T, 2>::C;
        return xs_order_for(diag_order, by_time, H, PJ, PK, C, geom.npj, n_patches);
    }
    std::vector<uint32_t> xs_order_for(const std::vector<uint32_t>& diag_order, bool by_time, int H, int PJ, int PK, int C, int npj,
                                       int n_patches) const {
        const int ndir = dim == 3 ? 8 : 4;
        std::vector<uint32_t> out;
        out.reserve((size_t)n_patches * ndir);
        if (!by_time) {
            for (int d = 0; d < ndir; ++d)
                for (uint32_t e : diag_order) out.push_back((e & 0xffffu) | ((e >> 16) << 14) | ((uint32_t)d << 28));
            return out;
        }
        auto flags = [&](int d, int& rj, int& rk) {
            if (dim == 3) { rj = (d >> 1) & 1; rk = (d >> 2) & 1; } else { rj = (d == 1) | (d == 2); rk = 0; }
        };
        std::vector<double> ts((size_t)n_patches * ndir, 0.0), tf((size_t)n_patches * ndir, 0.0);
        const double hop_j = (PJ + C - 1.0) / C + 1.0, hop_k = (PK + C - 1.0) / C + 1.0;
        for (int d = 0; d < ndir; ++d) {
            int rj, rk, prj = 0, prk = 0;
            flags(d, rj, rk);
            if (d > 0) flags(d - 1, prj, prk);
            for (uint32_t e : diag_order) {
                const int TJ = e & 0xffffu, TK = e >> 16;
                const size_t me = (size_t)d * n_patches + (size_t)TK * npj + TJ;
                double t = 0.0;
                if (TJ > 0) t = std::max(t, ts[me - 1] + hop_j);
                if (TK > 0) t = std::max(t, ts[me - npj] + hop_k);
                const int j0 = TJ * PJ, k0 = TK * PK;
                const int jm = std::min(j0 + PJ, geom.NJ) - 1, km = std::min(k0 + PK, geom.NK) - 1;
                if (d > 0) {
                    const int ja = std::max(j0 - 2 * H, 0), jb = std::min(jm + 2 * H, geom.NJ - 1);
                    const int ka = std::max(k0 - 2 * H, 0), kb = std::min(km + 2 * H, geom.NK - 1);
                    const int ja2 = rj != prj ? geom.NJ - 1 - jb : ja, jb2 = rj != prj ? geom.NJ - 1 - ja : jb;
                    const int ka2 = rk != prk ? geom.NK - 1 - kb : ka, kb2 = rk != prk ? geom.NK - 1 - ka : kb;
                    for (int tk = ka2 / PK; tk <= kb2 / PK; ++tk)
                        for (int tj = ja2 / PJ; tj <= jb2 / PJ; ++tj) t = std::max(t, tf[(size_t)(d - 1) * n_patches + (size_t)tk * npj + tj]);
                }
                ts[me] = t;
                tf[me] = t + ((jm - j0) + (km - k0) + geom.NF + C - 1) / C + 1.0;
            }
        }
        std::vector<uint32_t> idx((size_t)n_patches * ndir);
        // stable sort by start time; ties keep (direction, anti-diagonal) order
        size_t q = 0;
        std::vector<uint32_t> ent((size_t)n_patches * ndir);
        std::vector<double> key((size_t)n_patches * ndir);
        for (int d = 0; d < ndir; ++d)
            for (uint32_t e : diag_order) {
                const int TJ = e & 0xffffu, TK = e >> 16;
                ent[q] = (uint32_t)TJ | ((uint32_t)TK << 14) | ((uint32_t)d << 28);
                key[q] = ts[(size_t)d * n_patches + (size_t)TK * npj + TJ];
                idx[q] = (uint32_t)q;
                ++q;
            }
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a2, uint32_t b2) { return key[a2] < key[b2]; });
        for (uint32_t i2 : idx) out.push_back(ent[i2]);
        return out;
    }

    // The 3-D WENO stage exists with two chunk lengths: 8 levels (the kernel is latency bound with few
    // sources: fewer chunk boundaries) and 4 levels (157 instead of 208 VGPRs in pair mode, no spills, three
    // waves per SIMD: +14 % throughput once >= 8 slot groups are swept together).  Same results either way.
    template <int DIM, int H>
    void launch_sweeps_persistent(int batch) {
        constexpr int C0 = ChunkCfg<T, DIM>::C;
        // One field per workgroup (lone sources, batches below the pairing threshold): such launches are bound by the chains of their
        // chunks rather than by registers, and chunks of 16 levels halve the staging, the write-back and the hand-offs per level --
        // ms per sweep-iteration with chunks of 8 / 16 levels, 512^3: lone source 7.20 / 6.87, 2 sources 8.11 / 7.15, 4: 10.57 / 9.72,
        // 8 (unpaired): 16.36 / 15.24; 256^3: 1 source 3.29 / 3.15, 2: 3.69 / 3.30, 4: 4.28 / 3.55, 8: 4.49 / 4.04 (chunks of 4: 11.99 /
        // 4.96 for the lone source; PAIRS with chunks of 16: 48 ms instead of 17 for 8 sources, 22 ms with two workgroups per CU and no
        // spills).  fp64: the lone source 8.08 / 7.39 (256^3), 13.03 / 11.40 (384^3), 2 sources 9.01 / 7.62, 4: 10.39 / 9.12, but 8 sources
        // 12.0 / 12.3 and 36.0 / 43.9 (384^3): up to four.  Same partial order, same results, same exact skipping (a chunk is the unit
        // that is skipped: twice as coarse).
        // The WENO stage likewise where it is bound by its chains: 256^3 fp32 1 source 380.6 / 318.3 ms per solve (46 WENO iterations),
        // 2 sources 397 / 371, but 4: 501 / 564, 8: 761 / 981 (bound by its arithmetic there, and the longer chunk costs it registers);
        // fp64 1 source 663.5 / 536.7, 2 sources 757 / 589.  profiles/r05/experiment_chunk_length.txt
        if constexpr (DIM == 3 && C0 == 8) {
            const int below = H == 2 ? weno_c16_below : sizeof(T) == 4 ? std::numeric_limits<int>::max() : f64_c16_below;
            // (the tolerance-grade kernels of one field per workgroup exist with chunks of 16 levels only)
            if (NS == 1 && mode == 2 && (lone_chunk == 16 || fast_now<H>()) && batch < below) {
                launch_sweeps_persistent_ns<DIM, H, 1, 16, true>(batch);
                return;
            }
        }
        // (pair layout only: with one field per slot -- the default of weno grids since round 2 -- the 8-level chunks are
        // 2-4 % faster at every batch size, profiles/r02/weno_chunk.txt)
        if (H == 2 && DIM == 3 && NS == 2 && batch >= weno_ch4_min && C0 == 8) {
            if (NS == 2) launch_sweeps_persistent_ns<DIM, H, 2, (H == 2 && DIM == 3) ? 4 : C0>(batch);
            else launch_sweeps_persistent_ns<DIM, H, 1, (H == 2 && DIM == 3) ? 4 : C0>(batch);
        } else {
            if (NS == 2) launch_sweeps_persistent_ns<DIM, H, 2, C0>(batch); else launch_sweeps_persistent_ns<DIM, H, 1, C0>(batch);
        }
    }

    template <int DIM, int H, int NSV, int CH, bool XS_ONLY = false>   // XS_ONLY: only the whole-iteration launch (mode 2) is instantiated
    void launch_sweeps_persistent_ns(int batch) {
        using C = TileCfg<T, DIM>;
        PersistArgs<T> pa;
        SweepArgs<T>& a = pa.s;
        // (the occupancy cap of the lone source and the upwind counters sampled one chunk ahead (PRE) were tuned for chunks of 8 levels: on
        // first-order chunks of 16 they cost -- lone source 512^3 6.89 -> 6.68 ms per sweep-iteration without both, 256^3 3.19 -> 3.07, fp64
        // 256^3 7.35 -> 7.11; batches without PRE: 512^3 x 2 / 4 sources 7.21 -> 7.09 / 9.69 -> 9.64, 256^3 x 2 / 4 / 8 3.30 -> 3.15 /
        // 3.57 -> 3.47 / 4.04 -> 3.97)
        constexpr bool NO_PRE = CH == 16 && H == 1 && DIM == 3;
        const size_t dyn_lds = NO_PRE ? 0 : xs_dyn_lds(batch);
        a.tt = d_tt.p;
        a.ts = NS;
        a.lmask = d_lmask.p;
        a.frozen = d_mask.p;
        a.bbox = d_bbox.p;
        a.change = d_change.p;
        a.slots = d_slots.p;
        a.tiles = nullptr;
        a.g = geom;
        a.prof = d_prof.p;
        a.mask_words = (uint32_t)mask_words;
        a.dx = dx;
        a.dz = dz;
        a.w = 0;
        a.variant = DIM == 3 ? 0 : (dx == dz ? 1 : 2);
        pa.order = d_order.p;
        pa.sync = d_sync.p;
        pa.n_patches = n_patches;
        pa.batch = batch;
        pa.timeout_ticks = 300000000ull;  // 3 s at 100 MHz
        pa.stamp = d_stamp.p;
        pa.iter_ptr = d_iter.p;
        pa.evals = d_evals.p;
        pa.nbf = nbf; pa.nbj = nbj; pa.nbk = nbk;
        pa.ndir = DIM == 3 ? 8 : 4;
        pa.skip = skip_now(batch) ? (std::getenv("TTCR_FSM_SKIP_ALL_DIRTY") ? 2 : 1) : 0;
        pa.cmap = d_cmap.p;
        pa.cw = cmap_words;
        pa.sw = std::getenv("TTCR_FSM_NO_SW") ? nullptr : d_sw.p;   // (tuning / bisecting: no whole-sweep shortcut)
        pa.n_sw_groups = n_groups();
        pa.n_sw_sweeps = sw_sweeps;

        // first-order 3-D kernels: workgroups take units until the tickets run out -- no more of them than can be resident
        // (8 per CU is more than any instantiation fits; the surplus finds the ticket counter exhausted).  The others (and
        // TTCR_FSM_WGS=0, tuning): one workgroup per unit.
        const size_t wg_cap = (fsm_looped(DIM == 3, H) && persist_wgs > 0) ? (size_t)persist_wgs : ~(size_t)0;
        const dim3 block(C::PJ * C::PK), grid((unsigned)std::min<size_t>((size_t)n_patches * batch, wg_cap));
        const int ndir = DIM == 3 ? 8 : 4;
        {
            const bool pre_ = mode == 2 && !NO_PRE && (DIM == 2 || batch >= pre_min || dyn_lds > 0);
            char nm[160];
            std::snprintf(nm, sizeof nm, "fsm_sweep_persistent<%s,%d,%d,%d,%s,%s,%d,%d,%s,%s>", sizeof(T) == 4 ? "float" : "double", C::PJ, C::PK, CH,
                          DIM == 3 ? "true" : "false", skip_now(batch) ? "true" : "false", H, NSV, mode == 2 ? "true" : "false", pre_ ? "true" : "false");
            last_kernel = nm;
        }
        if (mode == 2) {
            // whole iteration in one launch: tickets direction-major, sweeps overlap at their ends
            pa.ssh = d_ssh.p;
            pa.ssh_stride = ssh_stride;
            pa.dir = 0;
            a.rf = a.rj = a.rk = a.rev = 0;
            a.s_sheared = nullptr;
            pa.timeout_ticks = 1000000000ull;  // 10 s: a unit may wait for most of the previous sweep
            const dim3 gridx((unsigned)std::min<size_t>((size_t)n_patches * batch * ndir, wg_cap));
            pa.order = d_order_xs[H == 2 ? 1 : 0][batch < time_order_below ? 1 : 0].p;
            const bool pre = DIM == 2 || batch >= pre_min || dyn_lds > 0;   // counters sampled one chunk ahead (template PRE)
            if constexpr (std::is_same<T, float>::value) {
                // tolerance-grade arithmetic: the AR = 1 instantiation of the kernel chosen below (fsm_fast.hip); the WENO stage of grids
                // that keep their fields in pairs (TTCR_FSM_PAIR = 1 on a weno grid: tuning only) has none and keeps the exact kernels
                if (fast_now<H>() && (H == 1 || NSV == 1)) {
                    const FastCfg fc{H, DIM, NSV, CH, skip_now(batch), NO_PRE ? false : pre};
                    last_kernel.insert(last_kernel.size() - 1, ",1");
                    const hipError_t e = fsm_fast_launch(pa, fc, gridx.x, dyn_lds, stream);
                    if (e == hipErrorInvalidValue) throw std::logic_error("arith = 1: no such kernel (" + last_kernel + ")");
                    HIP_CHECK(e);
                    return;
                }
            }
            if constexpr (NO_PRE) {
                (void)pre;
                if (skip_now(batch))
                    fsm_sweep_persistent<T, C::PJ, C::PK, CH, DIM == 3, true, H, NSV, true><<<gridx, block, dyn_lds, stream>>>(pa);
                else
                    fsm_sweep_persistent<T, C::PJ, C::PK, CH, DIM == 3, false, H, NSV, true><<<gridx, block, dyn_lds, stream>>>(pa);
            } else {
                if (skip_now(batch) && pre)
                    fsm_sweep_persistent<T, C::PJ, C::PK, CH, DIM == 3, true, H, NSV, true, true><<<gridx, block, dyn_lds, stream>>>(pa);
                else if (skip_now(batch))
                    fsm_sweep_persistent<T, C::PJ, C::PK, CH, DIM == 3, true, H, NSV, true><<<gridx, block, dyn_lds, stream>>>(pa);
                else if (pre)
                    fsm_sweep_persistent<T, C::PJ, C::PK, CH, DIM == 3, false, H, NSV, true, true><<<gridx, block, dyn_lds, stream>>>(pa);
                else
                    fsm_sweep_persistent<T, C::PJ, C::PK, CH, DIM == 3, false, H, NSV, true><<<gridx, block, dyn_lds, stream>>>(pa);
            }
            HIP_CHECK(hipGetLastError());
            return;
        }
        if (fast_now<H>() && H == 1) throw ValueError("option 'arith' = 1 needs whole-iteration launches (option 'mode' = 2)");
        if constexpr (XS_ONLY) throw std::logic_error("launch_sweeps_persistent_ns: whole-iteration launches only");
        else {
        pa.ssh = nullptr;
        pa.ssh_stride = 0;
        static const int RX2[4] = {0, 1, 1, 0}, RZ2[4] = {0, 0, 1, 1};
        for (int d = 0; d < ndir; ++d) {
            int fam;
            if (DIM == 3) {
                a.rf = d & 1; a.rj = (d >> 1) & 1; a.rk = (d >> 2) & 1;
                a.rev = a.rk;
                fam = (a.rf ^ a.rk) | ((a.rj ^ a.rk) << 1);
            } else {
                a.rj = RX2[d]; a.rf = RZ2[d]; a.rk = 0;
                a.rev = a.rj;
                fam = a.rf ^ a.rj;
            }
            a.s_sheared = d_ssh.p + (size_t)fam * ssh_stride;
            pa.dir = d;
            // (no reset of tickets, progress words or change maps: launch epoch pa.iter_ptr[1] + d; the abort word is sticky
            // within an iteration)
            if (skip_now(batch))
                fsm_sweep_persistent<T, C::PJ, C::PK, CH, DIM == 3, true, H, NSV, false><<<grid, block, 0, stream>>>(pa);
            else
                fsm_sweep_persistent<T, C::PJ, C::PK, CH, DIM == 3, false, H, NSV, false><<<grid, block, 0, stream>>>(pa);
        }
        HIP_CHECK(hipGetLastError());
        }
    }

    // For every launch w of a sweep, the patches (TJ,TK) whose level window
    // [BL*w - (TJ+TK)*(BL-1), +BL) contains nodes.  Direction-independent (oriented indices).
    void build_tile_lists(int PJ, int PK, int BL) {
        std::vector<uint32_t> all;
        tile_off.assign(n_launch, 0);
        tile_cnt.assign(n_launch, 0);
        for (int w = 0; w < n_launch; ++w) {
            tile_off[w] = (int)all.size();
            // anti-diagonal order keeps tiles that share halo columns close in the list
            for (int m = 0; m <= geom.npj + geom.npk - 2; ++m) {
                const int L0 = BL * w - m * (BL - 1);
                for (int TK = std::max(0, m - geom.npj + 1); TK <= std::min(m, geom.npk - 1); ++TK) {
                    const int TJ = m - TK;
                    const int j0 = TJ * PJ, k0 = TK * PK;
                    const int jmaxp = std::min(j0 + PJ, geom.NJ) - 1, kmaxp = std::min(k0 + PK, geom.NK) - 1;
                    if (L0 + BL - 1 < j0 + k0 || L0 > jmaxp + kmaxp + geom.NF - 1) continue;
                    all.push_back((uint32_t)TJ | ((uint32_t)TK << 16));
                }
            }
            tile_cnt[w] = (int)all.size() - tile_off[w];
        }
        d_tiles.reserve(std::max<size_t>(all.size(), 1));
        HIP_CHECK(hipMemcpy(d_tiles.p, all.data(), all.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }

    ~GridT() override {
        (void)hipSetDevice(device);
        for (int i = 0; i < 2; ++i) {
            if (graph_execs[i]) (void)hipGraphExecDestroy(graph_execs[i]);
            if (graphs[i]) (void)hipGraphDestroy(graphs[i]);
        }
        if (h_change) (void)hipHostFree(h_change);
        if (h_slots) (void)hipHostFree(h_slots);
        if (h_lmask) (void)hipHostFree(h_lmask);
        if (h_abort) (void)hipHostFree(h_abort);
        if (h_iter) (void)hipHostFree(h_iter);
        if (h_evals) (void)hipHostFree(h_evals);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (fill_stream) { (void)hipStreamSynchronize(fill_stream); (void)hipStreamDestroy(fill_stream); }
        if (ev_fill) (void)hipEventDestroy(ev_fill);
        if (stream) (void)hipStreamDestroy(stream);
    }

    int count_launches(int BL) const {
        const int Lhi = (geom.NJ - 1) + (geom.NK - 1) + geom.NF - 1;
        const int m = geom.npj + geom.npk - 2;
        return (Lhi + m * (BL - 1)) / BL + 1;
    }

    // ---- slowness ---------------------------------------------------------------------
    // c_order (3-D): `s` is the (nx, ny, nz) array in C order; it is uploaded as it lies and permuted to the
    // solver's x-fastest order on the device (the strided host-side flatten('F') of a 512^3 model takes 0.8 s)
    void set_slowness(const void* s, size_t n, bool on_device, bool c_order) override {
        HIP_CHECK(hipSetDevice(device));
        const size_t expect = cell ? n_cells : n_nodes;
        if (n != expect) throw std::length_error("Error: slowness vectors of incompatible size.");
        const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        if (cell) d_cells.reserve(n_cells);
        T* dst = cell ? d_cells.p : d_s.p;
        DevBuf<T> staged;
        if (c_order && dim == 3) {
            const int ax = (int)ncx + (cell ? 0 : 1), ay = (int)ncy + (cell ? 0 : 1), az = (int)ncz + (cell ? 0 : 1);
            staged.reserve(n);
            HIP_CHECK(hipMemcpyAsync(staged.p, s, n * sizeof(T), kind, stream));
            fsm_c_to_x_fastest<T><<<dim3((az + 31) / 32, (ax + 31) / 32, ay), dim3(32, 8), 0, stream>>>(staged.p, dst, ax, ay, az);
            HIP_CHECK(hipGetLastError());
        } else {
            HIP_CHECK(hipMemcpyAsync(dst, s, n * sizeof(T), kind, stream));
        }
        if (cell) {
            const int blocks = (int)std::min<size_t>((n_nodes + 255) / 256, 4096);
            if (dim == 3)
                fsm_cells_to_nodes3d<T><<<blocks, 256, 0, stream>>>(d_cells.p, d_s.p, (int)ncx, (int)ncy, (int)ncz);
            else
                fsm_cells_to_nodes2d<T><<<blocks, 256, 0, stream>>>(d_cells.p, d_s.p, (int)ncx, (int)ncz);
            HIP_CHECK(hipGetLastError());
        }
        // sheared copies (fsm_kernels.h: fsm_shear_slowness): family = F/J flips with K (3-D) or
        // J (2-D) not flipped; a direction and its opposite share one copy
        {
            const int nfam = dim == 3 ? 4 : 2;
            const dim3 grid((geom.MP + 15) / 16, (geom.NJ + 15) / 16, geom.NK);
            const bool lines = grid.y <= 65535 && grid.z <= 65535 && !std::getenv("TTCR_FSM_SHEAR_SCATTER");
            const int blocks = (int)std::min<size_t>((n_nodes + 255) / 256, 8192);
            for (int f = 0; f < nfam; ++f) {
                if (lines) fsm_shear_slowness_lines<T><<<grid, 256, 0, stream>>>(d_s.p, d_ssh.p + (size_t)f * ssh_stride, geom, f & 1, (f >> 1) & 1);
                else fsm_shear_slowness<T><<<blocks, 256, 0, stream>>>(d_s.p, d_ssh.p + (size_t)f * ssh_stride, geom, f & 1, (f >> 1) & 1);
            }
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipStreamSynchronize(stream));
        have_slowness = true;
    }

    void get_slowness(void* out, size_t n) override {
        HIP_CHECK(hipSetDevice(device));
        if (n != n_nodes) throw std::length_error("Error: slowness vectors of incompatible size.");
        HIP_CHECK(hipMemcpyAsync(out, d_s.p, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    void get_tt(int slot, void* out, size_t n) override {
        HIP_CHECK(hipSetDevice(device));
        check_slot(slot);
        slot = P(slot);
        if (n != n_nodes) throw ValueError("traveltime buffer has wrong size");
        if (NS == 1) {
            HIP_CHECK(hipMemcpyAsync(out, tt_ptr(slot), n * sizeof(T), hipMemcpyDeviceToHost, stream));
        } else {
            d_gather.reserve(n_nodes);
            const int blocks = (int)std::min<size_t>((n_nodes + 255) / 256, 8192);
            fsm_gather_field<T><<<blocks, 256, 0, stream>>>(tt_ptr(slot), d_gather.p, n_nodes, NS);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(out, d_gather.p, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        }
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    // The fields of a slot group are interleaved (T[group][node][NS]): a consumer that wants n_nodes contiguous
    // values gets a de-interleaved copy in a scratch buffer of the grid; the zero-copy view comes with its stride.
    void* tt_device(int slot) override {
        HIP_CHECK(hipSetDevice(device));
        check_slot(slot);
        slot = P(slot);
        if (NS == 1) return tt_ptr(slot);
        d_gather.reserve(n_nodes);
        const int blocks = (int)std::min<size_t>((n_nodes + 255) / 256, 8192);
        fsm_gather_field<T><<<blocks, 256, 0, stream>>>(tt_ptr(slot), d_gather.p, n_nodes, NS);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(stream));
        return d_gather.p;
    }
    void* tt_device_view(int slot, size_t* stride) override {
        check_slot(slot);
        if (stride) *stride = (size_t)NS;
        return tt_ptr(P(slot));
    }

    void check_slot(int slot) const {
        if (slot < 0 || slot >= n_slots) throw ValueError("Thread number is larger than number of threads");
    }

    // ---- points -----------------------------------------------------------------------
    int ncoord() const { return dim == 3 ? 3 : 2; }

    // Grid3Drn::checkPts (ttcr/Grid3Drn.h:771-790) / Grid2Drn::checkPts (ttcr/Grid2Drn.h:333-342)
    void check_pts(const T* p, int n) const {
        for (int m = 0; m < n; ++m) {
            if (dim == 3) {
                const T x = p[3 * m], y = p[3 * m + 1], z = p[3 * m + 2];
                if (x < xmin || x > xmax || y < ymin || y > ymax || z < zmin || z > zmax) {
                    std::ostringstream msg;
                    msg << "Error: Point (" << x << ' ' << y << ' ' << z << ") outside grid.";
                    throw std::runtime_error(msg.str());
                }
            } else {
                const T x = p[2 * m], z = p[2 * m + 1];
                if (x < xmin || x > xmax || z < zmin || z > zmax) {
                    std::ostringstream msg;
                    msg << "Error: Point (" << x << ", " << z << ") outside grid.";
                    throw std::runtime_error(msg.str());
                }
            }
        }
    }

    static T node_coord_h(T cmin, uint32_t n, T d) { return cmin + (T)n * d; }

    // first node index whose coordinate is within `small` of v (Node3Dn::operator==,
    // ttcr/Node3Dn.h:147-149, scanned in increasing order like initFSM's linear search)
    int first_match(T cmin, T d, uint32_t nn, T v) const {
        const double small = 1.e-4;
        // candidates around the nearest node; scan a window that covers every possible match
        double est = ((double)v - (double)cmin) / (double)d;
        long c = (long)std::floor(est);
        long span = (long)std::ceil(small / std::fabs((double)d)) + 2;
        long lo = std::max<long>(0, c - span), hi = std::min<long>((long)nn - 1, c + span + 1);
        for (long i = lo; i <= hi; ++i) {
            const T diff = node_coord_h(cmin, (uint32_t)i, d) - v;
            if ((double)(diff < 0 ? -diff : diff) < small) return (int)i;
        }
        return -1;
    }

    InitPoint<T> locate(const T* p, T t0) const {
        InitPoint<T> q;
        std::memset(&q, 0, sizeof(q));
        q.t0 = t0;
        if (dim == 3) {
            q.x = p[0]; q.y = p[1]; q.z = p[2];
            const int fi = first_match(xmin, dx, ncx + 1, q.x);
            const int fj = first_match(ymin, dx, ncy + 1, q.y);
            const int fk = first_match(zmin, dx, ncz + 1, q.z);
            if (fi >= 0 && fj >= 0 && fk >= 0) {
                q.on_node = 1; q.i = fi; q.j = fj; q.k = fk;
            } else {
                // Grid3Drn::getCellNo (ttcr/Grid3Drn.h:207-215) + decomposition (:3530-3534)
                const double small2 = 1.e-4 * 1.e-4;
                const T x = (double)(xmax - q.x) < small2 ? (T)((double)xmax - .5 * (double)dx) : q.x;
                const T y = (double)(ymax - q.y) < small2 ? (T)((double)ymax - .5 * (double)dx) : q.y;
                const T z = (double)(zmax - q.z) < small2 ? (T)((double)zmax - .5 * (double)dx) : q.z;
                const uint32_t nx = (uint32_t)(small2 + (double)((x - xmin) / dx));
                const uint32_t ny = (uint32_t)(small2 + (double)((y - ymin) / dx));
                const uint32_t nz = (uint32_t)(small2 + (double)((z - zmin) / dx));
                const uint32_t cellNo = ny * ncx + nz * (ncx * ncy) + nx;
                const long c = (long)cellNo;
                const long k = c / ((long)ncy * ncx);
                const long j = (c - k * (long)ncy * ncx) / ncx;
                q.i = (int)(c - (k * (long)ncy + j) * ncx);
                q.j = (int)j;
                q.k = (int)k;
            }
        } else {
            q.x = p[0]; q.z = p[1]; q.y = 0;
            const int fi = first_match(xmin, dx, ncx + 1, q.x);
            const int fk = first_match(zmin, dz, ncz + 1, q.z);
            if (fi >= 0 && fk >= 0) {
                q.on_node = 1; q.i = fi; q.k = fk;
            } else {
                // Grid2Drn::getCellNo uses `small` (ttcr/Grid2Drn.h:173-179)
                const double small = 1.e-4;
                const T x = (double)(xmax - q.x) < small ? (T)((double)xmax - .5 * (double)dx) : q.x;
                const T z = (double)(zmax - q.z) < small ? (T)((double)zmax - .5 * (double)dz) : q.z;
                const uint32_t nx = (uint32_t)(small + (double)((x - xmin) / dx));
                const uint32_t nz = (uint32_t)(small + (double)((z - zmin) / dz));
                const long c = (long)(uint32_t)(nx * ncz + nz);
                q.i = (int)(c / ncz);
                q.k = (int)(c - (long)q.i * ncz);
            }
        }
        return q;
    }

    // ---- sweeps -----------------------------------------------------------------------
    template <int DIM>
    void launch_sweeps(int batch) {
        using C = TileCfg<T, DIM>;
        if (fast_now<1>()) throw ValueError("option 'arith' = 1 needs whole-iteration launches (option 'mode' = 2)");
        SweepArgs<T> a;
        a.tt = d_tt.p;
        a.ts = NS;
        a.lmask = nullptr;
        a.frozen = d_mask.p;
        a.bbox = d_bbox.p;
        a.change = d_change.p;
        a.slots = d_slots.p;
        a.g = geom;
        a.prof = d_prof.p;
        a.mask_words = (uint32_t)mask_words;
        a.dx = dx;
        a.dz = dz;
        a.variant = DIM == 3 ? 0 : (dx == dz ? 1 : 2);
        const dim3 block(C::PJ * C::PK);
        const int ndir = DIM == 3 ? 8 : 4;
        // 2-D direction order (i+,j+), (i-,j+), (i-,j-), (i+,j-)  (ttcr/Grid2Drn.h:717-751);
        // here F = z (the reference's j), J = x (the reference's i)
        static const int RX2[4] = {0, 1, 1, 0}, RZ2[4] = {0, 0, 1, 1};
        for (int d = 0; d < ndir; ++d) {
            int fam;
            if (DIM == 3) {
                a.rf = d & 1; a.rj = (d >> 1) & 1; a.rk = (d >> 2) & 1;  // ttcr/Grid3Drn.h:2816-2899
                a.rev = a.rk;
                fam = (a.rf ^ a.rk) | ((a.rj ^ a.rk) << 1);
            } else {
                a.rj = RX2[d]; a.rf = RZ2[d]; a.rk = 0;
                a.rev = a.rj;
                fam = a.rf ^ a.rj;
            }
            a.s_sheared = d_ssh.p + (size_t)fam * ssh_stride;
            for (int w = 0; w < n_launch; ++w) {
                if (tile_cnt[w] == 0) continue;
                a.w = w;
                a.tiles = d_tiles.p + tile_off[w];
                const dim3 grid(tile_cnt[w], 1, batch);
                fsm_sweep_tile<T, C::PJ, C::PK, C::BL, DIM == 3><<<grid, block, 0, stream>>>(a);
            }
        }
        last_kernel = std::string("fsm_sweep_tile<") + (sizeof(T) == 4 ? "float" : "double") + "," + std::to_string(C::PJ) + "," + std::to_string(C::PK) + "," + std::to_string(C::BL) + "," + (DIM == 3 ? "true" : "false") + ">";
        HIP_CHECK(hipGetLastError());
    }

    // Interleaved field layout: traveltime fields are stored in groups of NS sources,
    // T[group][node][NS] (NS = 2 when the grid has at least two slots).  The persistent kernel then
    // marches the NS sources of a group together: one address computation, one 8-byte load/store
    // and one LDS access serve both, and the per-level bookkeeping is shared.
    int NS = 1;
    bool ns_window = false;       // the grid may change its layout between calls (see the constructor)
    int pair_layout = -1;         // option "pair_layout": -1 (default) by the rule above, 0 one field per workgroup, 1 pairs (grids in the window only)
    double last_eval_frac = -1;   // evaluated / all node updates of the last batch that restarted every slot (-1: none yet)
    double layout_lo = 0.55, layout_hi = 0.70;   // pairs -> one field per workgroup below lo, back above hi (TTCR_FSM_LAYOUT_LO / _HI: tests, tuning)
    // Called at the top of a batch that restarts EVERY slot (no field of an earlier call survives it): pairs (NS = 2) or one field per
    // workgroup (NS = 1) for this and the following calls.  Nothing else depends on the layout across calls: stamps, change maps, sweep
    // tallies and snapshots are per solve, the second set of fields is filled with max() whatever its layout, host-side results are per slot.
    void choose_layout() {
        if (!ns_window) return;
        int want = NS;
        if (pair_layout >= 0) want = pair_layout ? 2 : 1;
        else if (last_eval_frac >= 0) want = NS == 2 ? (last_eval_frac < layout_lo ? 1 : 2) : (last_eval_frac > layout_hi ? 2 : 1);
        if (want == NS) return;
        NS = want;
        graph_batches[0] = graph_batches[1] = -1;   // (captured launches hold the layout)
        for (int q = 0; q < n_slots; ++q) phys[q] = q;
    }
    int n_groups() const { return (n_slots + NS - 1) / NS; }
    T* tt_ptr(int slot) const { return d_tt.p + (size_t)(slot / NS) * n_nodes * NS + slot % NS; }
    // Tolerance-grade arithmetic (fp32 grids).  arith = 1: the first-order sweeps of grids WITHOUT the WENO stage -- within north_star's
    // 1e-5 s RMS of the reference.  A grid with the WENO stage keeps the reference's arithmetic in BOTH stages under arith = 1: the WENO
    // iteration amplifies differences of an ulp in its input field to 1e-3 s (measured, profiles/r06/weno_sensitivity.txt: the exact WENO
    // stage behind a tolerance-grade first-order stage ends 4e-5 s RMS / 4e-3 s max away from the reference) -- only the bit-identical
    // first-order field reproduces the reference there.  arith = 2: both stages of such grids as well, outside that bound, for callers
    // who take the WENO stage's own sensitivity as their tolerance.
    template <int H>
    bool fast_now() const { return sizeof(T) == 4 && (arith == 2 || (arith == 1 && !weno)); }
    bool prefill_on() const {
        if (prefill >= 0) return prefill != 0;
        const size_t bytes = n_nodes * (size_t)(n_slots + (n_slots & 1)) * sizeof(T);
        if (bytes < ((size_t)64 << 20)) return false;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
        return d_tt_alt.p || (2 * bytes <= total_b / 2 && bytes + ((size_t)1 << 30) <= free_b);
    }

    int stage = 0;  // 0: first-order sweeps, 1: WENO3 sweeps (persistent kernel only)
    // Exact skipping of chunks / units / sweeps that cannot change a node: on wherever it was measured to pay
    // (profiles/r03/skip_sweep.txt).  What it saves is work, what it costs is a little of every chunk's time: a win once
    // the chip is busy (512^3: from two slot groups on, 1.13x ... 1.7x at 32 groups), a loss of 1-15 % where a 3-D solve is
    // bound by the dependent chain of its units anyway (a lone source or pair; 256^3 and smaller up to 8 sources).  The
    // WENO stage gains at every batch size (its chunks are expensive), and so do the one-wave patches of the 2-D solver
    // (1.03x ... 1.19x from 1 to 64 sources on 1024^2 ... 8192^2 nodes, although the SKIP kernels do not follow the
    // previous sweep up the columns).  TTCR_FSM_SKIP_UNITS: work units per sweep from which the first-order 3-D sweeps skip.
    int skip_units_min = 2048;
    // Round 6: between 1 024 and 2 048 units (a lone 512^3 source, 2 - 4 sources at 384^3 ...) whether skipping pays depends on the MODEL -- 512^3,
    // one source, ms per solve without / with: gradient model 13.3 / 12.8 (arith = 1: 10.9 / 9.6), rough 16^3-block model 65.5 / 68.2 (52.2 / 54.9);
    // 384^3: 9.4 / 9.2 and 32.5 / 35.1 (profiles/r06/lone_skip_models.txt).  Such fp32 launches follow the evaluated fraction of the grid's last
    // skipping solve: on while it stayed below 0.6, off above, and tried again after eight solves without.
    int skip_probe_min = 1024;
    bool probe_skip = true;      // what a launch in that window does now
    int probe_off_calls = 0;     // solves in the window since skipping was switched off
    int persist_wgs = 2048;   // workgroups of a sweep launch (each takes units until none is left): 8 per CU; 0: one per unit
    bool in_probe_window(int entries) const {
        const long long u = (long long)n_patches * entries;
        return dim == 3 && sizeof(T) == 4 && u >= skip_probe_min && u < skip_units_min;
    }
    int skip_default(int entries) const {
        if (dim != 3) return 1;
        if (stage == 1) return 1;
        if (in_probe_window(entries)) return probe_skip ? 1 : 0;
        return (long long)n_patches * entries >= skip_units_min ? 1 : 0;
    }
    bool persistent_now() const { return mode >= 1 || stage == 1; }
    // the rotated-template sweeps change nodes without stamping their bricks: no skipping next to them
    // `entries`: batch entries (slot groups / slots) swept together
    bool skip_now(int entries) const {
        const int on = skip < 0 ? skip_default(entries) : skip;
        // (the SKIP kernels pack a level count and 8 flag bits into one progress word, and keep one mask bit per F brick)
        const bool fits = (long long)geom.NF + geom.NJ + geom.NK < (1ll << 22) && nbf <= 32 * FSM_SLAB_WORDS;
        return on != 0 && fits && !(dim == 2 && rotated && !weno && dx == dz);
    }

    // Grid2Drn::sweep45 for every source of the batch (entries as handed to the sweep kernels)
    void launch_sweep45(int batch) {
        Sweep45Args<T> a;
        a.tt = d_tt.p;
        a.s = d_s.p;
        a.frozen = d_mask.p;
        a.change = d_change.p;
        a.slots = d_slots.p;
        a.lmask = d_lmask.p;
        a.ts = NS;
        a.by_group = persistent_now() ? 1 : 0;
        a.nnx = (int)ncx + 1;
        a.nnz = (int)ncz + 1;
        a.n_nodes = n_nodes;
        a.mask_words = (uint32_t)mask_words;
        a.dx = dx;
        const int blocks = a.by_group ? batch * NS : batch;
        // rows in sequence, a row in parallel: four row buffers of nnz+2 values in LDS (fits up to ~10 000 fp32 /
        // ~5 000 fp64 nodes along z); wider grids take the strip kernel with its level ring
        const size_t smem = 4 * ((size_t)a.nnz + 2) * sizeof(T);
        if (smem <= 156 * 1024 && !sweep45_strips) {
            const int threads = (int)std::min<size_t>(1024, (((size_t)a.nnz + 63) / 64) * 64);
            fsm_sweep45_rows<T><<<blocks, threads, smem, stream>>>(a);
        } else {
            constexpr int NTMAX = sizeof(T) == 4 ? 1024 : 512;
            if (a.nnx + 2 <= 256) fsm_sweep45<T, 256><<<blocks, 256, 0, stream>>>(a);
            else fsm_sweep45<T, NTMAX><<<blocks, NTMAX, 0, stream>>>(a);
        }
        HIP_CHECK(hipGetLastError());
    }

    void issue_sweeps(int batch) {
        issue_sweeps_axis(batch);
        if (dim == 2 && rotated && !weno && dx == dz) launch_sweep45(batch);
    }

    void issue_sweeps_axis(int batch) {
        if (stage == 1) {
            if (dim == 3) launch_sweeps_persistent<3, 2>(batch); else launch_sweeps_persistent<2, 2>(batch);
        } else if (mode >= 1) {
            if (dim == 3) launch_sweeps_persistent<3, 1>(batch); else launch_sweeps_persistent<2, 1>(batch);
        } else {
            if (dim == 3) launch_sweeps<3>(batch); else launch_sweeps<2>(batch);
        }
    }

    void run_iteration(int batch) {
        const int ndir = dim == 3 ? 8 : 4;
        if (use_graph >= 2 || (use_graph == 1 && !persistent_now())) {
            hipGraph_t& graph = graphs[stage];
            hipGraphExec_t& graph_exec = graph_execs[stage];
            const int key = mode * 2 + (skip_now(batch) ? 1 : 0);
            if (!graph_exec || graph_batches[stage] != batch || graph_modes[stage] != key) {
                if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
                if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
                HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                try {
                    issue_sweeps(batch);
                } catch (...) {
                    hipGraph_t junk = nullptr;
                    (void)hipStreamEndCapture(stream, &junk);
                    if (junk) (void)hipGraphDestroy(junk);
                    throw;
                }
                HIP_CHECK(hipStreamEndCapture(stream, &graph));
                HIP_CHECK(hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0));
                graph_batches[stage] = batch;
                graph_modes[stage] = key;
            }
            HIP_CHECK(hipGraphLaunch(graph_exec, stream));
        } else {
            issue_sweeps(batch);
        }
        timing.launches += mode == 2 ? 1 : persistent_now() ? (long long)ndir : (long long)ndir * n_launch;
    }

    // One batch: sources src_ids[b] solved concurrently, source b in slot slot_ids[b].
    // ---- the reference's stopping rule where it matters (option stopping_rule = 1) ---------------------------------
    // Grid3Drnfs::raytrace (ttcr/Grid3Drnfs.h:141-152) sums abs(times[n] - T[n]) over the nodes in order, in T1.  The sweep kernels
    // accumulate the same quantity as an fp64 sum of decreases; the two agree to a few percent near eps * N at 1.3e8 fp32 nodes
    // (profiles/r03/stopping_rule_512.txt), so an iteration whose fp64 change lies within [1/2, 16] x eps * N (fp64 grids: 1e-6
    // either side) is decided by the reference's own sum, computed by fsm_reference_change from a snapshot of the field.  The
    // snapshot of a slot group is taken before an iteration when the change of the iteration before was below 1e4 x 16 x eps * N
    // (consecutive iterations differ by factors of 4 - 40); an iteration that lands in the window without one is decided by the
    // fp64 sum and counted (reference_sums_missed).
    std::map<int, DevBuf<T>> snap;            // slot group -> field(s) before the current iteration
    std::vector<int> snap_iter;               // [group] iteration (stage-local, 1-based) the snapshot belongs to, 0: none
    std::vector<double> prev_change;          // [slot] fp64 change of the iteration before (inf: none yet)
    std::vector<double> prev2_change;         // [slot] ... and of the one before that
    bool snap_always = false;                 // this solve missed a snapshot once: no more predictions
    std::vector<int> snap_T;                  // [group] global iteration index the snapshot is the field in front of (-1: none / not to be trusted)
    bool stamps_ok = false;                   // every iteration of this solve (and stage) so far ran a kernel that keeps the dirty-brick stamps
    int iter_T = 0;                           // global index of the iteration that runs / has just run (sweep numbers of the stamps: ndir * iter_T + 1 ...)
    bool stamps_live_for(int entries) const { return (stopping_shortcuts & 1) && dim == 3 && !weno && persistent_now() && skip_now(entries) && n_nodes * (size_t)NS < ((size_t)1 << 31); }
    // the pass over the fields of group gi and its snapshot (fsm_refsum_terms): terms of the sources asked for (x0 / x1, counts c0 / c1;
    // all nullptr: the snapshot alone); since_T >= 0: only the bricks that changed in iteration since_T or later
    void terms_pass(int gi, T* x0, T* x1, unsigned* c0, unsigned* c1, int since_T) {
        RefTermsArgs<T> ta;
        ta.cur = d_tt.p + (size_t)gi * n_nodes * NS;
        ta.old = snap[gi].p;
        ta.x[0] = x0; ta.x[1] = x1;
        ta.cnt[0] = c0; ta.cnt[1] = c1;
        ta.n_nodes = (uint32_t)n_nodes;
        ta.ns = NS;
        constexpr int V = 16 / (int)sizeof(T);
        const size_t n_el = n_nodes * (size_t)NS;
        const bool vec = n_el % V == 0 && ((uintptr_t)ta.cur | (uintptr_t)ta.old) % 16 == 0;
        const int nodes_per_vec = vec ? std::max(1, V / NS) : 1;
        const bool bricks = since_T >= 0 && geom.NF % nodes_per_vec == 0;
        ta.stamp = bricks ? d_stamp.p + (size_t)gi * n_bricks : nullptr;
        ta.thr = (dim == 3 ? 8 : 4) * since_T + 1;
        ta.NF = geom.NF; ta.NJ = geom.NJ; ta.nbf = nbf; ta.nbj = nbj;
        const unsigned nblk = (unsigned)((n_nodes + FSM_REFSUM_CB - 1) / FSM_REFSUM_CB);
        const unsigned blocks = std::min(nblk, 8192u);
        if (vec) fsm_refsum_terms<T, V><<<blocks, 256, 0, stream>>>(ta);
        else fsm_refsum_terms<T, 1><<<blocks, 256, 0, stream>>>(ta);
        HIP_CHECK(hipGetLastError());
    }
    DevBuf<size_t> d_ref_off;
    DevBuf<T> d_ref_out;
    double window_lo() const { return sizeof(T) == 4 ? 0.5 : 1.0 - 1e-6; }
    double window_hi() const { return sizeof(T) == 4 ? 16.0 : 1.0 + 1e-6; }
    void snapshots_before_iteration(const std::vector<int>& active, int it_next) {
        if (!stopping_rule || fixed_iters > 0) return;
        if ((int)snap_iter.size() != n_groups()) snap_iter.assign(n_groups(), 0);
        // a snapshot costs a copy of the field(s) of a slot group: taken whenever the iteration may land in the window -- the change
        // of the iteration before, continued with the decrease it showed against the one before it (consecutive iterations differ
        // by factors of 4 - 40, a factor changes by less than 8 from one iteration to the next) comes within the window; with one
        // iteration to go by, within 1e3 windows --, before the first iteration of the WENO stage (its `times` is the last
        // first-order field: the change can be anything), and always while the copy is cheap (up to 2^24 nodes)
        const bool cheap = n_nodes * (size_t)NS <= ((size_t)1 << 24);
        std::vector<char> done(n_groups(), 0);
        for (int s2 : active) {
            const int gi = s2 / NS;
            if (done[gi]) continue;
            bool may = prev_change[s2] < 1e3 * window_hi() * (double)epsilon;
            if (may && (int)prev2_change.size() == n_slots && std::isfinite(prev2_change[s2]) && prev2_change[s2] > 0) {
                const double r = std::min(1.0, prev_change[s2] / prev2_change[s2]);
                may = prev_change[s2] * r / 8.0 <= window_hi() * (double)epsilon;
            }
            // (a miss -- an iteration that landed in the window without a snapshot, decided by the fp64 sum -- shows the prediction does
            // not hold for this model: from then on every iteration of the solve is snapshotted)
            if (!(cheap || (stage == 1 && it_next == 1) || may || snap_always)) continue;
            if (it_next == 1 && stage == 0) continue;   // (the first iteration of a solve: its change is infinite -- every node comes down from max())
            done[gi] = 1;
            if (snap_iter[gi] == it_next) continue;     // (brought up to date by the pass that wrote the terms of the last sum: decide_go_on)
            DevBuf<T>& b = snap[gi];
            b.reserve(n_nodes * (size_t)NS);
            if ((int)snap_T.size() != n_groups()) snap_T.assign(n_groups(), -1);
            // (a snapshot that was right in front of an earlier iteration: the bricks that changed since then, by their stamps)
            if (stamps_ok && snap_T[gi] >= 0 && n_nodes * (size_t)NS < ((size_t)1 << 32)) terms_pass(gi, nullptr, nullptr, nullptr, nullptr, snap_T[gi]);
            else HIP_CHECK(hipMemcpyAsync(b.p, d_tt.p + (size_t)gi * n_nodes * NS, n_nodes * (size_t)NS * sizeof(T), hipMemcpyDeviceToDevice, stream));
            snap_iter[gi] = it_next;
            snap_T[gi] = iter_T;
        }
    }
    // The reference's `change` of one field: sum over the nodes, in order, in T1, of abs(times[n] - T[n]).  stopping_rule = 2: the
    // one-chain kernel (fsm_reference_change: 1.3e8 dependent additions for a 512^3 field); otherwise the same sum computed in
    // parallel, exactly (fsm_refsum_*, fsm_kernels.h): rounds of a tile scan over a window of the field + one workgroup that finds
    // where the running sum leaves its binade; the window follows the distance between such places.
    DevBuf<RefSumState> d_rs_state;
    DevBuf<RefSum4<T>> d_rs_tiles;
    DevBuf<T> d_rc_a, d_rc_b;
    size_t rs_fields_max = 16;               // fields of one batch of decide_go_on (0: no room for the compact arrays on this device)
    DevBuf<T> d_rs_x, d_rs_xc;               // [asked field][n_nodes, padded] terms of the sums of one decision; the non-zero ones, in order
    DevBuf<unsigned> d_rs_cnt;               // [asked field][block of FSM_REFSUM_CB nodes] non-zero terms
    DevBuf<unsigned long long> d_rs_off, d_rs_n, d_rs_n2;   // ... exclusive scan of them; [asked field] number of terms
    void reference_change_host(const void* times, const void* field, bool parallel, void* out) override {
        HIP_CHECK(hipSetDevice(device));
        d_rc_a.reserve(n_nodes);
        d_rc_b.reserve(n_nodes);
        HIP_CHECK(hipMemcpyAsync(d_rc_a.p, field, n_nodes * sizeof(T), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rc_b.p, times, n_nodes * sizeof(T), hipMemcpyHostToDevice, stream));
        *(T*)out = reference_change(d_rc_a.p, d_rc_b.p, 1, !parallel);
    }
    T reference_change(const T* cur, const T* old, int stride, bool one_chain) {
        std::vector<const T*> c{cur}, o{old};
        return reference_changes(c, o, stride, one_chain, std::numeric_limits<T>::infinity())[0];
    }
    DevBuf<const T*> d_rs_ptrs;
    DevBuf<unsigned> d_rs_arrived;
    // several fields at once: the rounds of all of them run side by side, enqueued in bunches (the state of every field stays on the
    // device between its rounds: start, sum, window; a field that is done lets its later rounds pass)
    // stop_at: a field is done once its running sum has reached this value (the sum only grows; the value returned is then a lower bound)
    // d_n: [field] number of terms on the device (compacted fields: fsm_refsum_scan's totals); nullptr: n_nodes each
    std::vector<T> reference_changes(const std::vector<const T*>& cur, const std::vector<const T*>& old, int stride, bool one_chain, T stop_at,
                                     const unsigned long long* d_n = nullptr, const std::vector<unsigned long long>* h_n_known = nullptr) {
        const size_t nf = cur.size();
        std::vector<T> out(nf, (T)0);
        if (one_chain) {
            d_ref_off.reserve(1);
            d_ref_out.reserve(1);
            for (size_t f = 0; f < nf; ++f) {
                RefChangeArgs<T> ra;
                ra.cur = cur[f];
                ra.old = old[f];
                const size_t zero = 0;
                HIP_CHECK(hipMemcpyAsync(d_ref_off.p, &zero, sizeof(size_t), hipMemcpyHostToDevice, stream));
                ra.off = d_ref_off.p;
                ra.out = d_ref_out.p;
                ra.n_nodes = n_nodes;
                ra.stride = stride;
                fsm_reference_change<T><<<1, 256, 0, stream>>>(ra);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipMemcpyAsync(&out[f], d_ref_out.p, sizeof(T), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
            }
            return out;
        }
        const size_t tiles_per_field = FSM_REFSUM_WMAX / FSM_REFSUM_TILE;
        d_rs_state.reserve(2 * nf);
        d_rs_tiles.reserve(nf * tiles_per_field);
        d_rs_ptrs.reserve(2 * nf);
        d_rs_arrived.reserve(nf);
        HIP_CHECK(hipMemsetAsync(d_rs_arrived.p, 0, nf * sizeof(unsigned), stream));
        std::vector<RefSumState> st(nf, RefSumState{0ull, 0ull, FSM_REFSUM_WMIN, 0ull});
        std::vector<const T*> ptrs(cur);
        ptrs.insert(ptrs.end(), old.begin(), old.end());
        HIP_CHECK(hipMemcpyAsync(d_rs_state.p, st.data(), nf * sizeof(RefSumState), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rs_ptrs.p, ptrs.data(), 2 * nf * sizeof(const T*), hipMemcpyHostToDevice, stream));
        RefSumArgs<T> ra;
        ra.cur = d_rs_ptrs.p;
        ra.old = d_rs_ptrs.p + nf;
        std::vector<unsigned long long> h_n(nf, (unsigned long long)n_nodes);
        unsigned long long n_max = n_nodes;
        if (!d_n) {
            d_rs_n.reserve(nf);
            HIP_CHECK(hipMemcpyAsync(d_rs_n.p, h_n.data(), nf * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
            d_n = d_rs_n.p;
        } else {
            // compacted fields: their lengths size the launches (windows that always reach to the end of the field were tried: fewer
            // rounds, 176 against 240 per solve of bench.py's heterogeneous leg, but 136 us each against 29 -- the scan is arithmetic)
            if (h_n_known) h_n = *h_n_known;
            else {
                HIP_CHECK(hipMemcpyAsync(h_n.data(), d_n, nf * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
            }
            n_max = 0;
            for (size_t f = 0; f < nf; ++f) n_max = std::max(n_max, h_n[f]);
            if (n_max == 0) return out;   // (no node changed at all: the sum of no terms)
        }
        ra.n = d_n;
        ra.stride = stride;
        ra.st = d_rs_state.p;
        ra.tiles = d_rs_tiles.p;
        ra.arrived = d_rs_arrived.p;
        ra.stop_at = stop_at;
        const unsigned max_tiles = (unsigned)std::min<unsigned long long>(tiles_per_field, (n_max + FSM_REFSUM_TILE - 1) / FSM_REFSUM_TILE);
        ra.round = 0;   // (round r reads buffer r & 1 of the states and writes the other one; the initial states sit in buffer 0)
        fsm_refsum_head<T><<<(unsigned)nf, 256, 0, stream>>>(ra);   // (... written by the pass over the head of every field)
        static const bool trace = std::getenv("TTCR_FSM_REFSUM_TRACE") != nullptr;   // tuning: one round per bunch, its state and wall clock on stderr
        for (int bunch = trace ? 1 : 16;; bunch = trace ? 1 : 8) {   // (rounds are enqueued in bunches: a field that is done lets the rest of its bunch pass)
            const auto t_b = std::chrono::steady_clock::now();
            for (int r = 0; r < bunch; ++r, ++ra.round) fsm_refsum_round<T><<<dim3(std::min(max_tiles, 1024u), (unsigned)nf), 256, 0, stream>>>(ra);
            HIP_CHECK(hipGetLastError());
            refsum_rounds += bunch;
            HIP_CHECK(hipMemcpyAsync(st.data(), d_rs_state.p + (size_t)(ra.round & 1) * nf, nf * sizeof(RefSumState), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            if (trace) {
                std::fprintf(stderr, "[refsum] round %3d  %7.1f us ", ra.round, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_b).count());
                for (size_t f = 0; f < nf; ++f) std::fprintf(stderr, " | start %llu of %llu window %llu", st[f].start, h_n[f], st[f].window);
                std::fprintf(stderr, "\n");
            }
            bool done = true;
            for (size_t f = 0; f < nf; ++f) done = done && st[f].start >= h_n[f];
            if (done) break;
        }
        for (size_t f = 0; f < nf; ++f) {
            if constexpr (sizeof(T) == 4) { const unsigned b = (unsigned)st[f].bits; std::memcpy(&out[f], &b, 4); } else { std::memcpy(&out[f], &st[f].bits, 8); }
        }
        return out;
    }
    // go on after this iteration?  (active slot s2, its fp64 change c, iteration `it` just done)
    std::vector<char> decide_go_on(const std::vector<int>& active, int it) {
        std::vector<char> go(active.size(), 0);
        std::vector<int> ask;
        if ((int)ref_change_last.size() != n_slots) ref_change_last.assign(n_slots, std::nan(""));
        for (int s2 : active) ref_change_last[s2] = std::nan("");
        for (size_t q = 0; q < active.size(); ++q) {
            const int s2 = active[q];
            const double c = h_change[s2];
            go[q] = c >= (double)epsilon;
            if (!stopping_rule || !(c >= window_lo() * (double)epsilon && c <= window_hi() * (double)epsilon)) continue;
            if ((int)snap_iter.size() == n_groups() && snap_iter[s2 / NS] == it) ask.push_back((int)q);
            else { ++reference_sums_missed; snap_always = true; }
        }
        if (ask.empty()) return go;
        std::vector<T> res(ask.size());
        std::vector<char> res_known(ask.size(), 0), res_go(ask.size(), 0);   // the sum itself / the side of eps * N it provably lies on
        if (stopping_rule == 2 || n_nodes * (size_t)NS >= ((size_t)1 << 32)) {
            std::vector<const T*> curs(ask.size()), olds(ask.size());
            for (size_t a = 0; a < ask.size(); ++a) {
                const int s2 = active[ask[a]], gi = s2 / NS;
                curs[a] = d_tt.p + (size_t)gi * n_nodes * NS + s2 % NS;
                olds[a] = snap[gi].p + s2 % NS;
            }
            res = reference_changes(curs, olds, NS, stopping_rule == 2, epsilon);   // (asked: change >= epsilon)
            res_known.assign(ask.size(), 1);
        } else {
            // The terms of the fields that were asked for, compact (fsm_refsum_terms: one pass over a group's fields and its snapshot, which
            // leaves the snapshot equal to the current field -- the one the next iteration would take), then the non-zero ones alone in
            // node order (fsm_refsum_scan / _compact), then the ordered pass over those.  Up to 16 fields at a time (whole groups).
            const size_t pitch = (n_nodes + FSM_REFSUM_CB - 1) / FSM_REFSUM_CB * FSM_REFSUM_CB;
            const unsigned nblk = (unsigned)(pitch / FSM_REFSUM_CB);
            for (size_t a0 = 0; a0 < ask.size();) {
                // (two arrays of n_nodes values per field of a batch: where the device has no room for 16 fields the batches get smaller,
                // and a grid that has none for a pair goes by the strided fields and whole-field snapshots, as in round 5)
                size_t a1 = a0, nf = 0;
                bool room = false;
                while (rs_fields_max >= 2) {
                    a1 = a0;
                    while (a1 < ask.size() && (a1 - a0 + 1 < rs_fields_max || (a1 > a0 && active[ask[a1]] / NS == active[ask[a1 - 1]] / NS))) ++a1;
                    nf = a1 - a0;
                    try {
                        d_rs_x.reserve(nf * pitch);
                        d_rs_xc.reserve(nf * pitch);
                        d_rs_cnt.reserve(nf * (size_t)nblk);
                        d_rs_off.reserve(nf * (size_t)nblk);
                        d_rs_n.reserve(std::max<size_t>(nf, 64));
                        room = true;
                        break;
                    } catch (const DeviceError&) {
                        (void)hipGetLastError();
                        d_rs_x.release();
                        d_rs_xc.release();
                        rs_fields_max = rs_fields_max > 2 ? std::max<size_t>(2, rs_fields_max / 2) : 0;
                    }
                }
                if (!room) {
                    std::vector<const T*> curs(ask.size() - a0), olds(ask.size() - a0);
                    for (size_t a = a0; a < ask.size(); ++a) {
                        const int s2 = active[ask[a]], gi = s2 / NS;
                        curs[a - a0] = d_tt.p + (size_t)gi * n_nodes * NS + s2 % NS;
                        olds[a - a0] = snap[gi].p + s2 % NS;
                    }
                    const std::vector<T> r = reference_changes(curs, olds, NS, false, epsilon);
                    for (size_t a = a0; a < ask.size(); ++a) { res[a] = r[a - a0]; res_known[a] = 1; }
                    break;
                }
                std::vector<const T*> curs(nf), olds(nf, nullptr);
                for (size_t a = a0; a < a1; ++a) {
                    curs[a - a0] = d_rs_xc.p + (a - a0) * pitch;
                    const int gi = active[ask[a]] / NS;
                    if (a > a0 && active[ask[a - 1]] / NS == gi) continue;   // (with the source in front of it: `active` is in slot order)
                    T* xp[2] = {nullptr, nullptr};
                    unsigned* cp[2] = {nullptr, nullptr};
                    for (size_t b = a; b < a1 && active[ask[b]] / NS == gi; ++b) {
                        xp[active[ask[b]] % NS] = d_rs_x.p + (b - a0) * pitch;
                        cp[active[ask[b]] % NS] = d_rs_cnt.p + (b - a0) * (size_t)nblk;
                    }
                    // (the snapshot is the field in front of this iteration: the bricks this iteration changed hold all the terms)
                    const bool bricks = stamps_ok && (int)snap_T.size() == n_groups() && snap_T[gi] == iter_T;
                    terms_pass(gi, xp[0], xp[1], cp[0], cp[1], bricks ? iter_T : -1);
                    snap_iter[gi] = it + 1;
                    if ((int)snap_T.size() != n_groups()) snap_T.assign(n_groups(), -1);
                    snap_T[gi] = iter_T + 1;
                }
                const dim3 cgrid(std::min(nblk, 4096u), (unsigned)nf);
                fsm_refsum_scan<0><<<(unsigned)nf, 1024, 0, stream>>>(d_rs_cnt.p, d_rs_off.p, nblk, d_rs_n.p);
                HIP_CHECK(hipGetLastError());
                // With M non-zero terms the sequential T1 sum lies within gamma = M u / (1 - M u) of the exact one (recursive summation of
                // non-negative terms, u the unit roundoff of T1; zeros are added exactly), and the fp64 sum of decreases the sweep kernels
                // hand over is the exact one to within m: every thread adds the decreases of its nodes of a unit -- at most
                // NF + NJ + NK of them -- in T1 before they go to the fp64 atomics.  Where these bounds put the reference's sum on one
                // side of eps * N, `change >= epsilon` is decided as the reference decides it without computing the sum (option
                // "stopping_shortcuts", bit 1); the ordered pass runs for the rest.
                std::vector<unsigned long long> h_m(nf);
                HIP_CHECK(hipMemcpyAsync(h_m.data(), d_rs_n.p, nf * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                std::vector<size_t> need;
                for (size_t f = 0; f < nf; ++f) {
                    const double u = 0.5 * (double)std::numeric_limits<T>::epsilon(), mu = (double)h_m[f] * u;
                    const double c = h_change[active[ask[a0 + f]]], e = (double)epsilon;
                    if (host_prof) std::fprintf(stderr, "[host]   ask: slot %d iteration %d  fp64 change / (eps N) %.4f  non-zero terms %llu (M u = %.4f)\n", active[ask[a0 + f]], it, c / e, h_m[f], mu);
                    if (h_m[f] == 0ull) { res[a0 + f] = (T)0; res_known[a0 + f] = 1; continue; }   // (the sum of no terms)
                    if ((stopping_shortcuts & 2) && stage == 0 && mu < 0.25) {   // (first-order sweeps: a node only ever comes down)
                        const double g = mu / (1.0 - mu), m = 2.0 * (double)(geom.NF + geom.NJ + geom.NK + 64) * u + 1e-6;
                        if (c * (1.0 - m) * (1.0 - g) >= e) { res_go[a0 + f] = 1; continue; }
                        if (c * (1.0 + m) * (1.0 + g) < e) { res_go[a0 + f] = 0; continue; }
                    }
                    need.push_back(f);
                }
                if (!need.empty()) {
                    fsm_refsum_compact<T><<<cgrid, 256, 0, stream>>>(d_rs_x.p, d_rs_xc.p, pitch, n_nodes, d_rs_cnt.p, d_rs_off.p, nblk);
                    HIP_CHECK(hipGetLastError());
                    std::vector<const T*> c2(need.size()), o2(need.size(), nullptr);
                    std::vector<unsigned long long> m2(need.size());
                    for (size_t q = 0; q < need.size(); ++q) { c2[q] = curs[need[q]]; m2[q] = h_m[need[q]]; }
                    d_rs_n2.reserve(std::max<size_t>(need.size(), 64));
                    HIP_CHECK(hipMemcpyAsync(d_rs_n2.p, m2.data(), need.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
                    const std::vector<T> r = reference_changes(c2, o2, 1, false, epsilon, d_rs_n2.p, &m2);   // (asked: change >= epsilon)
                    for (size_t q = 0; q < need.size(); ++q) { res[a0 + need[q]] = r[q]; res_known[a0 + need[q]] = 1; }
                }
                a0 = a1;
            }
        }
        for (size_t a = 0; a < ask.size(); ++a) {
            go[ask[a]] = res_known[a] ? res[a] >= epsilon : res_go[a] != 0;   // `change >= epsilon`, both T1 (ttcr/Grid3Drnfs.h:153)
            if (res_known[a]) ref_change_last[active[ask[a]]] = (double)res[a];
            ++reference_sums;
        }
        return go;
    }
    std::vector<double> ref_change_last;   // [slot] the reference's sum of the iteration just decided, if it was decided with it (NaN: not)

    // TTCR_FSM_HOST_PROF=1: wall clock of the host-side phases of a call (stderr), tuning only
    bool host_prof = std::getenv("TTCR_FSM_HOST_PROF") != nullptr;
    std::chrono::steady_clock::time_point hp_t = std::chrono::steady_clock::now();
    void hp_mark(const char* what) {
        if (!host_prof) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[host] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - hp_t).count());
        hp_t = now;
    }
    void solve_batch(const std::vector<int>& slot_ids, const std::vector<int>& src_ids, const int* tx_off,
                     const T* tx, const T* t0) {
        const int nb = (int)slot_ids.size();
        hp_mark("(before solve_batch)");
        const int nc = ncoord();
        const long long node_updates_before = timing.node_updates;
        // reinit + initFSM (ttcr/Grid3Drnfs.h:92-100)
        size_t tot_pts = 0;
        for (int b = 0; b < nb; ++b) tot_pts += tx_off[src_ids[b] + 1] - tx_off[src_ids[b]];
        std::vector<InitPoint<T>> pts;
        pts.reserve(tot_pts);
        std::vector<size_t> first(nb);
        for (int b = 0; b < nb; ++b) {
            first[b] = pts.size();
            for (int n = tx_off[src_ids[b]]; n < tx_off[src_ids[b] + 1]; ++n) pts.push_back(locate(tx + (size_t)nc * n, t0[n]));
        }
        d_pts.reserve(std::max<size_t>(tot_pts, 1));
        HIP_CHECK(hipMemcpyAsync(d_pts.p, pts.data(), pts.size() * sizeof(InitPoint<T>), hipMemcpyHostToDevice, stream));
        for (int b = 0; b < nb; ++b)  // dirty-brick stamps: one set per slot group, "never changed"
            HIP_CHECK(hipMemsetAsync(d_stamp.p + (size_t)(slot_ids[b] / NS) * n_bricks, 0xFF, n_bricks * sizeof(int), stream));
        // reinit (Node3Dn::reinit): T = max() everywhere.  When every source of an interleaved group is
        // (re)started in this batch the group is one contiguous fill (full-line stores) instead of NS
        // strided passes.
        std::vector<char> group_filled(n_groups(), 0);
        // every slot of the grid restarted by this batch: take the fields the side stream has initialised since the last such call
        const bool all_slots = nb == n_slots && prefill_on();
        if (all_slots && alt_filled) {
            HIP_CHECK(hipStreamWaitEvent(stream, ev_fill, 0));
            std::swap(d_tt.p, d_tt_alt.p);
            std::swap(d_tt.n, d_tt_alt.n);
            std::swap(d_tt.guard, d_tt_alt.guard);
            alt_filled = false;
            ++prefill_swaps;
            graph_batches[0] = graph_batches[1] = -1;   // (captured launches hold the field pointer)
            std::fill(group_filled.begin(), group_filled.end(), 1);
        }
        if (NS > 1) {
            std::vector<int> cnt(n_groups(), 0);
            for (int b = 0; b < nb; ++b) cnt[slot_ids[b] / NS] += 1;
            for (int gi = 0; gi < n_groups(); ++gi) {
                if (cnt[gi] != NS || group_filled[gi]) continue;   // (group_filled: the fields came initialised from the side stream)
                const size_t n_el = n_nodes * (size_t)NS;
                const int blocks = (int)std::min<size_t>((n_el + 255) / 256, 16384);
                fsm_fill<T><<<blocks, 256, 0, stream>>>(d_tt.p + (size_t)gi * n_el, n_el, real_traits<T>::max(), 1);
                group_filled[gi] = 1;
            }
        }
        for (int b = 0; b < nb; ++b) {
            const int slot = slot_ids[b];
            T* tt = tt_ptr(slot);
            const int blocks = (int)std::min<size_t>((n_nodes + 255) / 256, 8192);
            if (!group_filled[slot / NS]) fsm_fill<T><<<blocks, 256, 0, stream>>>(tt, n_nodes, real_traits<T>::max(), NS);
            HIP_CHECK(hipMemsetAsync(d_mask.p + (size_t)slot * mask_words, 0, mask_words * sizeof(uint32_t), stream));
            InitArgs<T> ia;
            ia.tt = tt;
            ia.ts = NS;
            ia.slowness = d_s.p;
            ia.frozen = d_mask.p + (size_t)slot * mask_words;
            ia.bbox = d_bbox.p + 6 * (size_t)slot;
            ia.pts = d_pts.p + first[b];
            ia.n_pts = tx_off[src_ids[b] + 1] - tx_off[src_ids[b]];
            ia.npts = weno ? 2 : 1;  // frozen box of the WENO solver (ttcr/Grid3Drnfs.h:98-100)
            ia.nnx = ncx + 1;
            ia.nny = dim == 3 ? ncy + 1 : 1;
            ia.nnz = ncz + 1;
            ia.dx = dx; ia.dz = dz; ia.xmin = xmin; ia.ymin = ymin; ia.zmin = zmin;
            ia.dim = dim;
            ia.stamp = d_stamp.p + (size_t)(slot / NS) * n_bricks;
            ia.nbf = nbf; ia.nbj = nbj; ia.nbk = nbk;
            fsm_init_source<T><<<1, 128, 0, stream>>>(ia);
            niter[slot] = 0;
            niterw[slot] = 0;
            if (change_hist.empty()) { change_hist.resize(n_slots); change_histw.resize(n_slots); }
            if (ref_hist.empty()) { ref_hist.resize(n_slots); ref_histw.resize(n_slots); }
            change_hist[slot].clear();
            change_histw[slot].clear();
            ref_hist[slot].clear();
            ref_histw[slot].clear();
        }
        HIP_CHECK(hipGetLastError());
        hp_mark("reinit + initFSM issued");
        HIP_CHECK(hipStreamSynchronize(stream));  // pts vector goes out of scope below; also surfaces errors early
        hp_mark("reinit + initFSM done");

        // driver loop of Grid3Drnfs::raytrace, per source: first-order sweeps until the L1 change
        // drops below eps*N (ttcr/Grid3Drnfs.h:137-153); with weno3 a second loop of WENO sweeps with
        // the change reset (:104-136).  A source that converges leaves the batch (slot -1).
        const int maxit = fixed_iters > 0 ? fixed_iters : nitermax;
        const int ndir = dim == 3 ? 8 : 4;
        if (!sync_clean || launch_epoch > 0xfffff000u) {   // a solve that did not end normally (or 2^32 launches): start over
            HIP_CHECK(hipMemsetAsync(d_sync.p, 0, sync_words * sizeof(int), stream));
            HIP_CHECK(hipMemsetAsync(d_cmap.p, 0, (size_t)n_patches * n_slots * ndir * 2 * cmap_words * sizeof(unsigned long long), stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            launch_epoch = 1;
        }
        sync_clean = false;
        snap_always = false;
        // progress words: wiped once per solve (an ordinary stream operation, like the fills above); between the launches of
        // the solve the 2-bit epoch field tells this launch's values from the previous launch's
        if (persistent_now() || weno) HIP_CHECK(hipMemsetAsync(d_sync.p + 8, 0, (sync_words - 8) * sizeof(int), stream));
        if (persistent_now() || weno) HIP_CHECK(hipMemsetAsync(d_sync.p + 4, 0, sizeof(int), stream));
        HIP_CHECK(hipMemsetAsync(d_evals.p, 0, sizeof(unsigned long long) * n_slots, stream));
        HIP_CHECK(hipMemsetAsync(d_sw.p, 0, sizeof(unsigned long long) * (size_t)sw_sweeps * n_groups(), stream));
        HIP_CHECK(hipEventRecord(ev0, stream));
        int it_total = 0;  // global iteration index: sweep numbers for the dirty-brick stamps
        for (stage = 0; stage < (weno ? 2 : 1); ++stage) {
            std::vector<int> active(slot_ids);
            int it = 0;
            prev_change.assign(n_slots, std::numeric_limits<double>::infinity());   // (the first iteration of a stage always runs: no snapshot)
            prev2_change.assign(n_slots, std::numeric_limits<double>::infinity());
            snap_iter.assign(n_groups(), 0);
            snap_T.assign(n_groups(), -1);
            stamps_ok = false;
            if ((int)ref_change_last.size() != n_slots) ref_change_last.assign(n_slots, std::nan(""));
            // batch entries: slot groups for the persistent kernel (with a lane mask), slots otherwise
            std::vector<int> groups;
            for (int s2 : slot_ids)
                if (groups.empty() || groups.back() != s2 / NS) groups.push_back(s2 / NS);
            while (!active.empty() && it < maxit) {
                int n_entries;
                if (persistent_now()) {
                    n_entries = (int)groups.size();
                    for (int b = 0; b < n_entries; ++b) {
                        int m2 = 0;
                        for (int s2 : active)
                            if (s2 / NS == groups[b]) m2 |= 1 << (s2 % NS);
                        h_slots[b] = m2 ? groups[b] : -1;
                        h_lmask[b] = m2;
                    }
                } else {
                    n_entries = nb;
                    for (int b = 0; b < nb; ++b) { h_slots[b] = b < (int)active.size() ? active[b] : -1; h_lmask[b] = 1; }
                }
                HIP_CHECK(hipMemcpyAsync(d_slots.p, h_slots, sizeof(int) * n_entries, hipMemcpyHostToDevice, stream));
                HIP_CHECK(hipMemcpyAsync(d_lmask.p, h_lmask, sizeof(int) * n_entries, hipMemcpyHostToDevice, stream));
                HIP_CHECK(hipMemsetAsync(d_change.p, 0, sizeof(double) * n_slots, stream));
                iter_T = it_total;
                snapshots_before_iteration(active, it + 1);
                if (host_prof) { HIP_CHECK(hipStreamSynchronize(stream)); hp_mark("snapshots"); }
                stamps_ok = (it == 0 ? true : stamps_ok) && stamps_live_for(n_entries);   // (per stage: the stamps are rewritten between the stages)
                h_iter[0] = it_total;
                h_iter[1] = (int)launch_epoch;   // epoch of this iteration's (first) sweep launch
                launch_epoch += (persistent_now() && mode == 2) ? 1u : (unsigned)ndir;
                HIP_CHECK(hipMemcpyAsync(d_iter.p, h_iter, 2 * sizeof(int), hipMemcpyHostToDevice, stream));
                run_iteration(n_entries);
                HIP_CHECK(hipMemcpyAsync(h_change, d_change.p, sizeof(double) * n_slots, hipMemcpyDeviceToHost, stream));
                if (persistent_now()) HIP_CHECK(hipMemcpyAsync(h_abort, d_sync.p + 4, sizeof(int), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                hp_mark("sweep-iteration");
                if (persistent_now() && *h_abort) {
                    HIP_CHECK(hipMemsetAsync(d_sync.p + 4, 0, sizeof(int), stream));
                    stage = 0;
                    throw DeviceError("persistent sweep kernel: a patch timed out waiting for its upwind neighbour");
                }
                ++it;
                ++it_total;
                std::vector<int> next;
                if (fixed_iters > 0) ref_change_last.assign(n_slots, std::nan(""));
                const std::vector<char> go = fixed_iters > 0 ? std::vector<char>(active.size(), 1) : decide_go_on(active, it);
                hp_mark("stopping rule");
                for (size_t q = 0; q < active.size(); ++q) {
                    const int s2 = active[q];
                    (stage == 0 ? niter : niterw)[s2] = it;
                    (stage == 0 ? change_hist : change_histw)[s2].push_back(h_change[s2]);
                    (stage == 0 ? ref_hist : ref_histw)[s2].push_back((int)ref_change_last.size() == n_slots ? ref_change_last[s2] : std::nan(""));
                    timing.node_updates += (long long)n_nodes * ndir;
                    prev2_change[s2] = prev_change[s2];
                    prev_change[s2] = h_change[s2];
                    if (go[q]) next.push_back(s2);
                }
                active.swap(next);
            }
            timing.iterations = std::max(timing.iterations, it_total);
            // the stamps of the first-order stage say nothing about the WENO stencil: every brick counts as changed in the
            // last sweep, so that the first WENO sweep visits every node once with the new formula
            if (stage == 0 && weno) {
                // (and no whole-sweep shortcut across the stage boundary: a tally of zero finished units never qualifies)
                HIP_CHECK(hipMemsetAsync(d_sw.p, 0, sizeof(unsigned long long) * (size_t)sw_sweeps * n_groups(), stream));
                for (int s2 : slot_ids)
                    HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)(d_stamp.p + (size_t)(s2 / NS) * n_bricks), it_total * (dim == 3 ? 8 : 4),
                                                n_bricks, stream));
            }
        }
        if (std::getenv("TTCR_FSM_DEBUG_SW")) {   // tuning: the whole-sweep tallies of the SKIP kernels
            std::vector<unsigned long long> h((size_t)std::min(sw_sweeps, 40) * n_groups());
            HIP_CHECK(hipMemcpy(h.data(), d_sw.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (size_t q = 0; q < h.size(); ++q) std::fprintf(stderr, "%s%u/%u", q % n_groups() == 0 ? "\n[sw] " : " ", (unsigned)h[q], (unsigned)(h[q] >> 32));
            std::fprintf(stderr, "\n");
        }
        const bool was_persistent = mode >= 1 || weno;
        stage = 0;
        sync_clean = true;
        HIP_CHECK(hipEventRecord(ev1, stream));
        if (all_slots) {
            // the other set of fields -- fresh, or the results of the call before this one, which nothing enqueued after the swap reads --
            // is initialised for the next call BEHIND this call's sweeps (the event just recorded): beside them the fill took
            // workgroup slots and bandwidth from the sweep kernel and was not done when the next call came (512^3 x 64 back to back:
            // 191.8 ms per step instead of 185.7, profiles/r05/README.md); behind them it runs beside the receiver interpolation and
            // whatever the caller does between two calls, and a call that comes at once waits for what is left of it
            const size_t n_el = n_nodes * (size_t)(n_slots + (n_slots & 1));   // (room for either layout)
            if (!fill_stream) {
                int lo = 0, hi = 0;
                HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                HIP_CHECK(hipStreamCreateWithPriority(&fill_stream, hipStreamNonBlocking, lo));
                HIP_CHECK(hipEventCreateWithFlags(&ev_fill, hipEventDisableTiming));
            }
            // (a second set of fields is a convenience: when the device has no room for it -- the snapshots of this solve were not
            // there yet when prefill_on() looked -- the solve that just finished must not fail for it)
            bool have_alt = true;
            try {
                d_tt_alt.reserve(n_el, 64);
            } catch (const DeviceError&) {
                (void)hipGetLastError();
                have_alt = false;
                prefill = 0;
            }
            if (have_alt) {
            HIP_CHECK(hipStreamWaitEvent(fill_stream, ev1, 0));
            const int blocks = (int)std::min<size_t>((n_el + 255) / 256, 16384);
            fsm_fill<T><<<blocks, 256, 0, fill_stream>>>(d_tt_alt.p, n_el, real_traits<T>::max(), 1);
            HIP_CHECK(hipEventRecord(ev_fill, fill_stream));
            alt_filled = true;
            }
        }
        HIP_CHECK(hipEventSynchronize(ev1));
        if (d_prof.p) {
            unsigned long long h[8];
            HIP_CHECK(hipMemcpy(h, d_prof.p, sizeof(h), hipMemcpyDeviceToHost));
            if (const char* tp = std::getenv("TTCR_FSM_PROF_TRACE")) {   // unit trace of the last iteration -> file
                std::vector<unsigned long long> tr(prof_words);
                HIP_CHECK(hipMemcpy(tr.data(), d_prof.p, prof_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                if (FILE* f = std::fopen(tp, "wb")) { std::fwrite(tr.data() + 8, sizeof(unsigned long long), prof_words - 8, f); std::fclose(f); }
            }
            HIP_CHECK(hipMemset(d_prof.p, 0, sizeof(h)));
            if (mode >= 1) {
                const double nb_ = (double)std::max<unsigned long long>(h[7], 1), nu_ = (double)std::max<unsigned long long>(h[6], 1);
                std::fprintf(stderr, "[ttcr_amd prof] units %llu chunks %llu  per unit (us): start %.2f total %.2f | per chunk (us): top %.3f  wait %.3f  stage %.3f  march %.3f  writeback %.3f\n",
                             h[6], h[7], h[5] * 0.01 / nu_, (h[0] + h[1] + h[2] + h[3] + h[4] + h[5]) * 0.01 / nu_, h[0] * 0.01 / nb_,
                             h[1] * 0.01 / nb_, h[2] * 0.01 / nb_, h[3] * 0.01 / nb_, h[4] * 0.01 / nb_);
            } else {
                const double nb_ = (double)std::max<unsigned long long>(h[4], 1);
                std::fprintf(stderr, "[ttcr_amd prof] tiles %llu  per tile (us): setup %.2f  stage %.2f  march %.2f  writeback %.2f\n",
                             h[4], h[0] * 0.01 / nb_, h[1] * 0.01 / nb_, h[2] * 0.01 / nb_, h[3] * 0.01 / nb_);
            }
        }
        hp_mark("after the iterations");
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
        timing.sweep_ms += ms;
        const long long evaluated_before = timing.evaluated_updates;
        if (was_persistent) {
            HIP_CHECK(hipMemcpy(h_evals, d_evals.p, sizeof(unsigned long long) * n_slots, hipMemcpyDeviceToHost));
            for (int s2 : slot_ids) timing.evaluated_updates += (long long)h_evals[s2];
        } else {
            timing.evaluated_updates += timing.node_updates - node_updates_before;   // every update is evaluated
        }
        const bool skipped_now = skip_now(n_groups());
        if (nb == n_slots && timing.node_updates > node_updates_before && skipped_now)   // (what choose_layout goes by; kernels that skip only)
            last_eval_frac = (double)(timing.evaluated_updates - evaluated_before) / (double)(timing.node_updates - node_updates_before);
        if (skip < 0 && !weno && in_probe_window((nb + NS - 1) / NS) && timing.node_updates > node_updates_before) {   // (see skip_probe_min)
            if (skipped_now) {
                probe_skip = (double)(timing.evaluated_updates - evaluated_before) / (double)(timing.node_updates - node_updates_before) < 0.6;
                probe_off_calls = 0;
            } else if (++probe_off_calls >= 8) {
                probe_skip = true;
                probe_off_calls = 0;
            }
        }
    }

    void interp(int slot, int n, const void* pts, void* out) override {
        HIP_CHECK(hipSetDevice(device));
        check_slot(slot);
        if (n <= 0) return;
        const int nc = ncoord();
        std::vector<T> p((const T*)pts, (const T*)pts + (size_t)nc * n);
        if (translate)
            for (int m = 0; m < n; ++m) { p[3 * m] -= ox; p[3 * m + 1] -= oy; p[3 * m + 2] -= oz; }
        check_pts(p.data(), n);
        interp_grid_coords(P(slot), n, p.data(), (T*)out);
    }

    // Grid3Drn::computeSlowness(pt, isTranslated) (ttcr/Grid3Drn.h:2451-2676), Grid2Drn::computeSlowness (ttcr/Grid2Drn.h:262-330)
    void compute_slowness(int n, const void* pts, bool translated, void* out) override {
        HIP_CHECK(hipSetDevice(device));
        if (!have_slowness) throw std::runtime_error("Error: slowness has not been assigned.");
        if (n <= 0) return;
        const int nc = ncoord();
        std::vector<T> p((const T*)pts, (const T*)pts + (size_t)nc * n);
        if (translate && !translated)
            for (int m = 0; m < n; ++m) { p[3 * m] -= ox; p[3 * m + 1] -= oy; p[3 * m + 2] -= oz; }
        check_pts(p.data(), n);   // (the reference indexes its node array unchecked; a point outside is refused here)
        d_rx.reserve((size_t)nc * n);
        d_out.reserve(n);
        HIP_CHECK(hipMemcpyAsync(d_rx.p, p.data(), sizeof(T) * nc * n, hipMemcpyHostToDevice, stream));
        const int blocks = (n + 63) / 64;
        if (dim == 3) {
            RayGeom<T> rg;
            rg.nnx = ncx + 1; rg.nny = ncy + 1; rg.nnz = ncz + 1;
            rg.dx = dx; rg.xmin = xmin; rg.ymin = ymin; rg.zmin = zmin; rg.xmax = xmax; rg.ymax = ymax; rg.zmax = zmax;
            rg.interp_vel = interp_vel;
            fsm_compute_slowness3d<T><<<blocks, 64, 0, stream>>>(rg, d_s.p, d_rx.p, n, d_out.p);
        } else {
            RayGeom2<T> rg2;
            rg2.nnx = ncx + 1; rg2.nnz = ncz + 1;
            rg2.dx = dx; rg2.dz = dz; rg2.xmin = xmin; rg2.zmin = zmin; rg2.xmax = xmax; rg2.zmax = zmax;
            fsm_compute_slowness2d<T><<<blocks, 64, 0, stream>>>(rg2, d_s.p, d_rx.p, n, d_out.p);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out, d_out.p, sizeof(T) * n, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    void interp_grid_coords(int slot, int n, const T* p, T* out) {
        if (n <= 0) return;
        const int nc = ncoord();
        d_rx.reserve((size_t)nc * n);
        d_out.reserve(n);
        HIP_CHECK(hipMemcpyAsync(d_rx.p, p, sizeof(T) * nc * n, hipMemcpyHostToDevice, stream));
        const T* tt = tt_ptr(slot);
        const int blocks = (n + 127) / 128;
        if (dim == 3)
            fsm_interp3d<T><<<blocks, 128, 0, stream>>>(tt, NS, d_rx.p, d_out.p, n, (int)ncx + 1, (int)ncy + 1, (int)ncz + 1, dx, xmin, ymin, zmin);
        else
            fsm_interp2d<T><<<blocks, 128, 0, stream>>>(tt, NS, d_rx.p, d_out.p, n, (int)ncx + 1, (int)ncz + 1, dx, dz, xmin, zmin);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out, d_out.p, sizeof(T) * n, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    // the same for every source of a batch: one upload, one launch, one read-back
    void interp_batch(const std::vector<int>& sl, const std::vector<int>& sr, const int* rx_off, const T* rx, T* tt_out) {
        const int nc = ncoord();
        size_t n = 0;
        for (int s2 : sr) n += (size_t)(rx_off[s2 + 1] - rx_off[s2]);
        if (n == 0) return;
        std::vector<T> p(nc * n), o(n);
        std::vector<int> so(n);
        size_t k = 0;
        for (size_t b = 0; b < sl.size(); ++b) {
            const int m = rx_off[sr[b] + 1] - rx_off[sr[b]];
            std::memcpy(p.data() + nc * k, rx + (size_t)nc * rx_off[sr[b]], sizeof(T) * nc * m);
            std::fill(so.begin() + k, so.begin() + k + m, sl[b]);
            k += m;
        }
        d_rx.reserve(nc * n);
        d_out.reserve(n);
        d_rslot.reserve(n);
        HIP_CHECK(hipMemcpyAsync(d_rx.p, p.data(), sizeof(T) * nc * n, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rslot.p, so.data(), sizeof(int) * n, hipMemcpyHostToDevice, stream));
        const int blocks = (int)((n + 127) / 128);
        if (dim == 3)
            fsm_interp3d_batch<T><<<blocks, 128, 0, stream>>>(d_tt.p, NS, n_nodes, d_rslot.p, d_rx.p, d_out.p, (int)n, (int)ncx + 1,
                                                              (int)ncy + 1, (int)ncz + 1, dx, xmin, ymin, zmin);
        else
            fsm_interp2d_batch<T><<<blocks, 128, 0, stream>>>(d_tt.p, NS, n_nodes, d_rslot.p, d_rx.p, d_out.p, (int)n, (int)ncx + 1, (int)ncz + 1, dx,
                                                              dz, xmin, zmin);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(o.data(), d_out.p, sizeof(T) * n, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        k = 0;
        for (size_t b = 0; b < sl.size(); ++b) {
            const int m = rx_off[sr[b] + 1] - rx_off[sr[b]];
            std::memcpy(tt_out + rx_off[sr[b]], o.data() + k, sizeof(T) * m);
            k += m;
        }
    }

    // a walk that failed on the device -> the reference's exception (ttcr/Grid3Drn.h:1170-1181); status 2 is our
    // step limit, where the reference would not return
    [[noreturn]] void throw_walk_error(int st, const T* rx, const T* tx, long max_steps) const {
        const int nc = ncoord();
        std::ostringstream msg;
        auto pt = [&](const T* v) { for (int c = 0; c < nc; ++c) msg << (c ? " " : "") << v[c]; };
        if (st == 1) {
            msg << "Error while computing raypaths: going outside grid \n                Rx: ";
            pt(rx);
            msg << "\n                Tx: ";
            pt(tx);
            msg << "\n";
        } else {
            msg << "Error while computing raypaths: ray from Rx ";
            pt(rx);
            msg << " did not reach the source within " << max_steps << " steps";
        }
        throw std::runtime_error(msg.str());
    }

    // getTraveltimeFromRaypath for the receivers of every source of a batch in ONE launch (one thread per receiver:
    // a single source's few hundred receivers leave the device empty); errors as in raypath_batch_rays
    void raypath_batch(const std::vector<int>& sl, const std::vector<int>& sr, const int* tx_off, const T* tx, const T* t0,
                       const int* rx_off, const T* rx, T* tt_out) {
        const int nc = ncoord();
        size_t n = 0;
        for (int s2 : sr) n += (size_t)(rx_off[s2 + 1] - rx_off[s2]);
        if (n == 0) return;
        std::vector<T> p(nc * n), o(n), txb;
        std::vector<T> t0b;
        std::vector<int> so(n), st(n);
        std::vector<RaySrc> desc(sl.size());
        size_t k = 0;
        for (size_t b = 0; b < sl.size(); ++b) {
            const int src = sr[b], m = rx_off[src + 1] - rx_off[src];
            desc[b].tt_off = (long long)(tt_ptr(sl[b]) - d_tt.p);
            desc[b].tx_off = (int)t0b.size();
            desc[b].n_tx = tx_off[src + 1] - tx_off[src];
            txb.insert(txb.end(), tx + (size_t)nc * tx_off[src], tx + (size_t)nc * tx_off[src + 1]);
            t0b.insert(t0b.end(), t0 + tx_off[src], t0 + tx_off[src + 1]);
            std::memcpy(p.data() + nc * k, rx + (size_t)nc * rx_off[src], sizeof(T) * nc * m);
            std::fill(so.begin() + k, so.begin() + k + m, (int)b);
            k += m;
        }
        d_rsrc.reserve(txb.size());
        d_rt0.reserve(t0b.size());
        d_rx.reserve(nc * n);
        d_out.reserve(n);
        d_rstat.reserve(n);
        d_rslot.reserve(n);
        d_rdesc.reserve(desc.size());
        HIP_CHECK(hipMemcpyAsync(d_rsrc.p, txb.data(), sizeof(T) * txb.size(), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rt0.p, t0b.data(), sizeof(T) * t0b.size(), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rx.p, p.data(), sizeof(T) * nc * n, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rslot.p, so.data(), sizeof(int) * n, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rdesc.p, desc.data(), sizeof(RaySrc) * desc.size(), hipMemcpyHostToDevice, stream));
        const long max_steps = walk_step_limit;
        const dim3 rgrid((unsigned)((n + 63) / 64)), rblock(64);
        if (dim == 3) {
            RayGeom<T> rg;
            rg.nnx = ncx + 1; rg.nny = ncy + 1; rg.nnz = ncz + 1;
            rg.dx = dx; rg.xmin = xmin; rg.ymin = ymin; rg.zmin = zmin; rg.xmax = xmax; rg.ymax = ymax; rg.zmax = zmax;
            rg.interp_vel = interp_vel;
            fsm_raypath3d<T, false><<<rgrid, rblock, 0, stream>>>(d_tt.p, NS, d_s.p, rg, 0, d_rsrc.p, d_rt0.p, d_rx.p, (int)n, d_out.p,
                                                                  d_rstat.p, max_steps, nullptr, 0, nullptr, d_rdesc.p, d_rslot.p);
        } else {
            RayGeom2<T> rg2;
            rg2.nnx = ncx + 1; rg2.nnz = ncz + 1;
            rg2.dx = dx; rg2.dz = dz; rg2.xmin = xmin; rg2.zmin = zmin; rg2.xmax = xmax; rg2.zmax = zmax;
            const T* cell_s = cell ? d_cells.p : nullptr;
            fsm_raypath2d<T, false><<<rgrid, rblock, 0, stream>>>(d_tt.p, NS, d_s.p, cell_s, rg2, 0, d_rsrc.p, d_rt0.p, d_rx.p, (int)n,
                                                                  d_out.p, d_rstat.p, max_steps, nullptr, 0, nullptr, d_rdesc.p, d_rslot.p);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(o.data(), d_out.p, sizeof(T) * n, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(st.data(), d_rstat.p, sizeof(int) * n, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        for (size_t q = 0; q < n; ++q)
            if (st[q] != 0) throw_walk_error(st[q], p.data() + (size_t)nc * q, txb.data() + (size_t)nc * desc[so[q]].tx_off, max_steps);
        k = 0;
        for (size_t b = 0; b < sl.size(); ++b) {
            const int m = rx_off[sr[b] + 1] - rx_off[sr[b]];
            std::memcpy(tt_out + rx_off[sr[b]], o.data() + k, sizeof(T) * m);
            k += m;
        }
    }

    // getRaypath (the r_data overloads, ttcr/Grid3D.h:546-586 / Grid2D) for the receivers of every source of a batch in ONE launch
    // per chunk of rows: a walk is a chain of dependent loads (≈ 10 ms for ANY number of receivers of a 256^3 source), so one
    // launch per source made a call with return_rays cost more than its solves.  Rows in batch order; recording rows hold the
    // length of an ordinary ray, a longer one is traced again alone; the recording buffer stays below rays_buffer_bytes.
    size_t rays_buffer_bytes = (size_t)4 << 30;
    void raypath_batch_rays(const std::vector<int>& sl, const std::vector<int>& sr, const int* tx_off, const T* tx, const T* t0,
                            const int* rx_off, const T* rx, T* tt_out, std::vector<std::vector<long long>>& src_ray_len,
                            std::vector<std::vector<T>>& src_ray_pts) {
        const int nc = ncoord();
        size_t n = 0;
        for (int s2 : sr) n += (size_t)(rx_off[s2 + 1] - rx_off[s2]);
        if (n == 0) return;
        std::vector<T> p(nc * n), o(n), txb, t0b;
        std::vector<int> so(n), st(n), np(n);
        std::vector<RaySrc> desc(sl.size());
        size_t k = 0;
        for (size_t b = 0; b < sl.size(); ++b) {
            const int src = sr[b], m = rx_off[src + 1] - rx_off[src];
            desc[b].tt_off = (long long)(tt_ptr(sl[b]) - d_tt.p);
            desc[b].tx_off = (int)t0b.size();
            desc[b].n_tx = tx_off[src + 1] - tx_off[src];
            txb.insert(txb.end(), tx + (size_t)nc * tx_off[src], tx + (size_t)nc * tx_off[src + 1]);
            t0b.insert(t0b.end(), t0 + tx_off[src], t0 + tx_off[src + 1]);
            std::memcpy(p.data() + nc * k, rx + (size_t)nc * rx_off[src], sizeof(T) * nc * m);
            std::fill(so.begin() + k, so.begin() + k + m, (int)b);
            k += m;
        }
        d_rsrc.reserve(txb.size());
        d_rt0.reserve(t0b.size());
        d_rx.reserve(nc * n);
        d_out.reserve(n);
        d_rstat.reserve(n);
        d_rslot.reserve(n);
        d_raynp.reserve(n);
        d_rdesc.reserve(desc.size());
        HIP_CHECK(hipMemcpyAsync(d_rsrc.p, txb.data(), sizeof(T) * txb.size(), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rt0.p, t0b.data(), sizeof(T) * t0b.size(), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rx.p, p.data(), sizeof(T) * nc * n, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rslot.p, so.data(), sizeof(int) * n, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rdesc.p, desc.data(), sizeof(RaySrc) * desc.size(), hipMemcpyHostToDevice, stream));
        RayGeom<T> rg;
        rg.nnx = ncx + 1; rg.nny = ncy + 1; rg.nnz = ncz + 1;
        rg.dx = dx; rg.xmin = xmin; rg.ymin = ymin; rg.zmin = zmin; rg.xmax = xmax; rg.ymax = ymax; rg.zmax = zmax;
        rg.interp_vel = interp_vel;
        RayGeom2<T> rg2;
        rg2.nnx = ncx + 1; rg2.nnz = ncz + 1;
        rg2.dx = dx; rg2.dz = dz; rg2.xmin = xmin; rg2.zmin = zmin; rg2.xmax = xmax; rg2.zmax = zmax;
        const T* cell_s = (dim == 2 && cell) ? d_cells.p : nullptr;
        // The reference's walk has no step limit: next to a corner it can go back and forth between two planes for tens of thousands
        // of steps before it drifts away (41 512 points for a receiver 1.7e-4 inside the far corner of a 2-D cell grid,
        // tests/test_parity_gpu.py).  The limit only turns a walk that would never end into an error.
        const long max_steps = walk_step_limit;
        const long cap = std::min<long>(max_steps, 8L * ((long)ncx + ncy + ncz + 3)) + 3;   // Rx, one point per step, <= two per source point
        // (the recording rows of a launch: at most rays_buffer_bytes, and at most a quarter of what the device has free -- a replica that is
        // full of slots, or a smaller device, records in more launches instead of failing its allocation)
        size_t rays_budget = rays_buffer_bytes, free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) rays_budget = std::min(rays_budget, std::max<size_t>(free_b / 4, (size_t)64 << 20));
        const size_t chunk = std::max<size_t>(1, std::min<size_t>(n, rays_budget / (sizeof(T) * nc * cap)));
        auto launch = [&](size_t row0, int m, T* pts, long rcap) {   // rows [row0, row0 + m) of the batch
            const dim3 rgrid((unsigned)((m + 63) / 64)), rblock(64);
            if (dim == 3)
                fsm_raypath3d<T, true><<<rgrid, rblock, 0, stream>>>(d_tt.p, NS, d_s.p, rg, 0, d_rsrc.p, d_rt0.p, d_rx.p + (size_t)nc * row0, m,
                                                                     d_out.p + row0, d_rstat.p + row0, max_steps, pts, rcap, d_raynp.p + row0,
                                                                     d_rdesc.p, d_rslot.p + row0);
            else
                fsm_raypath2d<T, true><<<rgrid, rblock, 0, stream>>>(d_tt.p, NS, d_s.p, cell_s, rg2, 0, d_rsrc.p, d_rt0.p, d_rx.p + (size_t)nc * row0, m,
                                                                     d_out.p + row0, d_rstat.p + row0, max_steps, pts, rcap, d_raynp.p + row0,
                                                                     d_rdesc.p, d_rslot.p + row0);
            HIP_CHECK(hipGetLastError());
        };
        auto compact = [&](const T* pts, long rcap, const long long* d_off, int m) {
            if (dim == 3)
                fsm_compact_rays<T><<<m, 128, 0, stream>>>(pts, rcap, d_off, d_raydense.p, shift_rays() ? ox : (T)0, shift_rays() ? oy : (T)0,
                                                           shift_rays() ? oz : (T)0);
            else
                fsm_compact_rays2<T><<<m, 128, 0, stream>>>(pts, rcap, d_off, d_raydense.p);
            HIP_CHECK(hipGetLastError());
        };
        std::vector<long long> off;
        std::vector<T> dense;
        for (size_t c0 = 0; c0 < n; c0 += chunk) {
            const int m = (int)std::min(chunk, n - c0);
            d_raypts.reserve((size_t)m * cap * nc);
            launch(c0, m, d_raypts.p, cap);
            HIP_CHECK(hipMemcpyAsync(o.data() + c0, d_out.p + c0, sizeof(T) * m, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(st.data() + c0, d_rstat.p + c0, sizeof(int) * m, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(np.data() + c0, d_raynp.p + c0, sizeof(int) * m, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            for (int q = 0; q < m; ++q)
                if (st[c0 + q] != 0 && st[c0 + q] != 3)
                    throw_walk_error(st[c0 + q], p.data() + (size_t)nc * (c0 + q), txb.data() + (size_t)nc * desc[so[c0 + q]].tx_off, max_steps);
            off.assign((size_t)m + 1, 0);
            for (int q = 0; q < m; ++q) off[q + 1] = off[q] + np[c0 + q];
            const long long tot = off[m];
            d_rayoff.reserve((size_t)m + 1);
            d_raydense.reserve((size_t)std::max<long long>(tot, 1) * nc);
            HIP_CHECK(hipMemcpyAsync(d_rayoff.p, off.data(), sizeof(long long) * ((size_t)m + 1), hipMemcpyHostToDevice, stream));
            compact(d_raypts.p, cap, d_rayoff.p, m);
            for (int q = 0; q < m; ++q) {   // rays longer than a row (status 3): once more, alone, with room
                if (st[c0 + q] != 3) continue;
                const long need = (long)np[c0 + q] + 1;
                d_raylong.reserve((size_t)need * nc);
                const long long off2[2] = {off[q], off[q + 1]};
                d_rayoff2.reserve(2);
                HIP_CHECK(hipMemcpyAsync(d_rayoff2.p, off2, sizeof(off2), hipMemcpyHostToDevice, stream));
                launch(c0 + q, 1, d_raylong.p, need);
                compact(d_raylong.p, need, d_rayoff2.p, 1);
                int st2 = 0, np2 = 0;
                HIP_CHECK(hipMemcpyAsync(&st2, d_rstat.p + c0 + q, sizeof(int), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipMemcpyAsync(&np2, d_raynp.p + c0 + q, sizeof(int), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipMemcpyAsync(o.data() + c0 + q, d_out.p + c0 + q, sizeof(T), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));   // (also: off2 goes out of scope)
                if (st2 != 0 || np2 != np[c0 + q]) throw DeviceError("raypath: a long ray did not retrace to the same length");
            }
            dense.resize((size_t)tot * nc);
            if (tot > 0) HIP_CHECK(hipMemcpyAsync(dense.data(), d_raydense.p, sizeof(T) * nc * tot, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            for (int q = 0; q < m; ++q) {   // hand the rays to their sources (rows of a source are consecutive)
                const int src = sr[so[c0 + q]];
                src_ray_len[src].push_back((long long)np[c0 + q]);
                src_ray_pts[src].insert(src_ray_pts[src].end(), dense.begin() + (size_t)off[q] * nc, dense.begin() + (size_t)off[q + 1] * nc);
            }
        }
        k = 0;
        for (size_t b = 0; b < sl.size(); ++b) {
            const int m = rx_off[sr[b] + 1] - rx_off[sr[b]];
            std::memcpy(tt_out + rx_off[sr[b]], o.data() + k, sizeof(T) * m);
            k += m;
        }
    }

    // rays of the last raytrace call with return_rays on, in the order of the receiver rows of that call
    std::vector<long long> rays_off{0};
    std::vector<T> rays_pts;
    DevBuf<T> d_raypts, d_raydense, d_raylong;
    DevBuf<long long> d_rayoff2;
    long walk_step_limit = 1000000;   // steps after which a ray walk is declared endless (the oracle uses the same number)
    DevBuf<int> d_raynp;
    DevBuf<long long> d_rayoff;

    bool shift_rays() const { return translate; }

    // ---- matrix M (the raytrace overloads with m_data: ttcr/Grid3D.h:743-772 -> Grid3Drn::getRaypath(Tx, t0, Rx, m_data, RxNo, tt,
    // threadNo), ttcr/Grid3Drn.h:1503-1800; with r_data as well, what ttcrpy calls for compute_M with return_rays:
    // ttcr/Grid3D.h:646-680 -> getRaypath(Tx, t0, Rx, r_data, m_data, RxNo, tt, threadNo), ttcr/Grid3Drn.h:2144-2470).  The walk
    // of either overload is a kernel of its own (fsm_raypath3d_m) that leaves one record (mid-point, length, slowness at the
    // mid-point) per term block of the reference; the host adds -s^2 * ds * w at the eight nodes around each mid-point.
    // Restated as it stands (weights without xmin, node indices that may lie one past the grid -- the Python layer drops those;
    // the zero-length segments of the m_data-only overload give signed zeros), entries merged by node in push order; a receiver
    // on the source: no entries and tt = 0, not t0.
    std::vector<std::vector<long long>> slot_m_off, slot_m_j;
    std::vector<std::vector<T>> slot_m_v;
    DevBuf<T> d_msegs, d_mlong;
    DevBuf<int> d_mnseg;
    // records of every receiver's walk: seg_off[q] .. seg_off[q+1] in segs (5 values each), traveltimes of the overload in out
    void walk_m(int slot, int n_tx, const T* txp, const T* t0p, int n, const T* p, T* out, bool both, std::vector<long long>& seg_off,
                std::vector<T>& segs) {
        seg_off.assign(1, 0);
        segs.clear();
        if (n <= 0) return;
        d_rsrc.reserve((size_t)3 * n_tx);
        d_rt0.reserve(n_tx);
        HIP_CHECK(hipMemcpyAsync(d_rsrc.p, txp, sizeof(T) * 3 * n_tx, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rt0.p, t0p, sizeof(T) * n_tx, hipMemcpyHostToDevice, stream));
        RayGeom<T> rg;
        rg.nnx = ncx + 1; rg.nny = ncy + 1; rg.nnz = ncz + 1;
        rg.dx = dx; rg.xmin = xmin; rg.ymin = ymin; rg.zmin = zmin; rg.xmax = xmax; rg.ymax = ymax; rg.zmax = zmax;
        rg.interp_vel = interp_vel;
        const long max_steps = walk_step_limit;
        // one record per step, up to two per source point at the end; a longer walk is done again with the room it asked for
        const long cap = std::min<long>(max_steps, 8L * ((long)ncx + ncy + ncz + 3)) + 2L * n_tx + 1;
        const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)256 << 20) / (sizeof(T) * 5 * cap)));
        std::vector<int> st(chunk), ns(chunk);
        std::vector<T> rows;
        for (int c0 = 0; c0 < n; c0 += chunk) {
            const int m = std::min(chunk, n - c0);
            const T* pc = p + (size_t)3 * c0;
            d_rx.reserve((size_t)3 * m);
            d_out.reserve(m);
            d_rstat.reserve(m);
            d_msegs.reserve((size_t)m * cap * 5);
            d_mnseg.reserve(m);
            HIP_CHECK(hipMemcpyAsync(d_rx.p, pc, sizeof(T) * 3 * m, hipMemcpyHostToDevice, stream));
            const dim3 rgrid((m + 63) / 64), rblock(64);
            if (both)
                fsm_raypath3d_m<T, true><<<rgrid, rblock, 0, stream>>>(tt_ptr(slot), NS, d_s.p, rg, n_tx, d_rsrc.p, d_rt0.p, d_rx.p, m, d_out.p,
                                                                       d_rstat.p, max_steps, d_msegs.p, cap, d_mnseg.p);
            else
                fsm_raypath3d_m<T, false><<<rgrid, rblock, 0, stream>>>(tt_ptr(slot), NS, d_s.p, rg, n_tx, d_rsrc.p, d_rt0.p, d_rx.p, m, d_out.p,
                                                                        d_rstat.p, max_steps, d_msegs.p, cap, d_mnseg.p);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(out + c0, d_out.p, sizeof(T) * m, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(st.data(), d_rstat.p, sizeof(int) * m, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(ns.data(), d_mnseg.p, sizeof(int) * m, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            for (int q = 0; q < m; ++q)
                if (st[q] != 0 && st[q] != 3) throw_walk_error(st[q], pc + (size_t)3 * q, txp, max_steps);
            size_t base = segs.size(), tot = 0;
            for (int q = 0; q < m; ++q) tot += (size_t)5 * ns[q];
            segs.resize(base + tot);   // (once per chunk: the copies below are asynchronous)
            for (int q = 0; q < m; ++q) {
                if (st[q] == 0) {
                    if (ns[q] > 0)
                        HIP_CHECK(hipMemcpyAsync(segs.data() + base, d_msegs.p + (size_t)q * cap * 5, sizeof(T) * 5 * ns[q], hipMemcpyDeviceToHost, stream));
                } else {   // status 3: once more, alone, with room
                    const long need = (long)ns[q] + 1;
                    d_mlong.reserve((size_t)need * 5);
                    if (both)
                        fsm_raypath3d_m<T, true><<<1, 64, 0, stream>>>(tt_ptr(slot), NS, d_s.p, rg, n_tx, d_rsrc.p, d_rt0.p, d_rx.p + (size_t)3 * q, 1,
                                                                       d_out.p + q, d_rstat.p + q, max_steps, d_mlong.p, need, d_mnseg.p + q);
                    else
                        fsm_raypath3d_m<T, false><<<1, 64, 0, stream>>>(tt_ptr(slot), NS, d_s.p, rg, n_tx, d_rsrc.p, d_rt0.p, d_rx.p + (size_t)3 * q, 1,
                                                                        d_out.p + q, d_rstat.p + q, max_steps, d_mlong.p, need, d_mnseg.p + q);
                    HIP_CHECK(hipGetLastError());
                    int st2 = 0, ns2 = 0;
                    HIP_CHECK(hipMemcpyAsync(&st2, d_rstat.p + q, sizeof(int), hipMemcpyDeviceToHost, stream));
                    HIP_CHECK(hipMemcpyAsync(&ns2, d_mnseg.p + q, sizeof(int), hipMemcpyDeviceToHost, stream));
                    HIP_CHECK(hipMemcpyAsync(segs.data() + base, d_mlong.p, sizeof(T) * 5 * ns[q], hipMemcpyDeviceToHost, stream));
                    HIP_CHECK(hipStreamSynchronize(stream));
                    if (st2 != 0 || ns2 != ns[q]) throw DeviceError("compute_M: a long walk did not retrace to the same length");
                }
                base += (size_t)5 * ns[q];
                seg_off.push_back(seg_off.back() + ns[q]);
            }
            HIP_CHECK(hipStreamSynchronize(stream));
        }
    }
    // rows of M from the records of walk_m: appended to (mo, mj, mv); mo ends with the running entry count
    void assemble_m(int n_rx, const std::vector<long long>& seg_off, const std::vector<T>& segs, std::vector<long long>& mo,
                    std::vector<long long>& mj, std::vector<T>& mv) const {
        const size_t nnx = ncx + 1, nny = ncy + 1;
        for (int r = 0; r < n_rx; ++r) {
            const size_t row0 = mj.size();
            for (long long q = seg_off[r]; q < seg_off[r + 1]; ++q) {
                const T* sg = segs.data() + 5 * (size_t)q;
                const T ds = sg[3];
                T sq = sg[4];
                sq *= sq;
                const size_t ix = (size_t)((sg[0] - xmin) / dx), iy = (size_t)((sg[1] - ymin) / dx), iz = (size_t)((sg[2] - zmin) / dx);
                for (size_t ii = 0; ii < 2; ++ii)
                    for (size_t jj = 0; jj < 2; ++jj)
                        for (size_t kk = 0; kk < 2; ++kk) {
                            const size_t iv = ix + ii, jv = iy + jj, kv = iz + kk;
                            const T dvdv = (T)((1. - std::abs(sg[0] - iv * dx) / dx) * (1. - std::abs(sg[1] - jv * dx) / dx) *
                                               (1. - std::abs(sg[2] - kv * dx) / dx));
                            const long long j = (long long)((kv * nny + jv) * nnx + iv);
                            const T v = -sq * ds * dvdv;
                            size_t e = row0;
                            for (; e < mj.size(); ++e)
                                if (mj[e] == j) { mv[e] += v; break; }
                            if (e == mj.size()) { mj.push_back(j); mv.push_back(v); }
                        }
            }
            mo.push_back((long long)mj.size());
        }
    }
    // the m_data overloads for every source of a call: the fields are solved in batches like raytrace_multi solves them, the walks
    // follow each batch (Grid3D's multi-source overloads with m_data, ttcr/Grid3D.h:896-1000, which run the single-source overload
    // per source on host threads).  One CSR over all receiver rows of the call, in row order.
    bool m_walk_mode = false, m_walk_both = false;
    std::vector<std::vector<long long>> m_seg_off;
    std::vector<std::vector<T>> m_segs;
    std::vector<long long> multi_m_off{0}, multi_m_j;
    std::vector<T> multi_m_v;
    void raytrace_multi_m(int n_src, const int* tx_off, const void* tx_v, const void* t0_v, const int* rx_off, const void* rx_v,
                          void* tt_out_v, bool both) override {
        if (dim != 3) throw Unsupported("compute_M is implemented for 3-D grids only");
        if (cell) throw Unsupported("compute_M not defined for grids with slowness defined for cells");
        multi_m_off.assign(1, 0); multi_m_j.clear(); multi_m_v.clear();
        if (n_src <= 0) return;
        m_seg_off.assign(n_src, std::vector<long long>{0});
        m_segs.assign(n_src, std::vector<T>());
        m_walk_mode = true; m_walk_both = both;
        try {
            raytrace_multi(n_src, tx_off, tx_v, t0_v, rx_off, rx_v, tt_out_v, -1, nullptr, both);
        } catch (...) { m_walk_mode = false; throw; }
        m_walk_mode = false;
        for (int n = 0; n < n_src; ++n) {
            assemble_m(rx_off[n + 1] - rx_off[n], m_seg_off[n], m_segs[n], multi_m_off, multi_m_j, multi_m_v);
            std::vector<T>().swap(m_segs[n]);
        }
    }
    void multi_m_size(size_t* n_rows, size_t* nnz) const override { *n_rows = multi_m_off.size() - 1; *nnz = multi_m_j.size(); }
    void get_multi_m(long long* row_off, long long* j, void* v) const override {
        std::memcpy(row_off, multi_m_off.data(), multi_m_off.size() * sizeof(long long));
        if (!multi_m_j.empty()) {
            std::memcpy(j, multi_m_j.data(), multi_m_j.size() * sizeof(long long));
            std::memcpy(v, multi_m_v.data(), multi_m_v.size() * sizeof(T));
        }
    }
    void raytrace_m(int slot, int n_tx, const void* tx_v, const void* t0_v, int n_rx, const void* rx_v, void* tt_out_v, bool both) override {
        check_slot(slot);
        if (dim != 3) throw Unsupported("compute_M is implemented for 3-D grids only");
        if (cell) throw Unsupported("compute_M not defined for grids with slowness defined for cells");
        if (n_tx < 1) throw ValueError("every source needs at least one point");
        const int tx_off[2] = {0, n_tx}, rx_off[2] = {0, n_rx};
        // the field of the source (and, for the overload that keeps them, the rays: the points of that overload are those of
        // the r_data overload); the receivers' traveltimes are then replaced by those of the m_data walk
        raytrace_multi(1, tx_off, tx_v, t0_v, rx_off, rx_v, tt_out_v, slot, nullptr, both);
        T* tt_out = (T*)tt_out_v;
        std::vector<T> txs((const T*)tx_v, (const T*)tx_v + 3 * (size_t)n_tx), rxs((const T*)rx_v, (const T*)rx_v + 3 * (size_t)n_rx);
        if (translate) {
            for (int q = 0; q < n_tx; ++q) { txs[3 * q] -= ox; txs[3 * q + 1] -= oy; txs[3 * q + 2] -= oz; }
            for (int q = 0; q < n_rx; ++q) { rxs[3 * q] -= ox; rxs[3 * q + 1] -= oy; rxs[3 * q + 2] -= oz; }
        }
        std::vector<long long> seg_off;
        std::vector<T> segs;
        // (the source was solved in the PHYSICAL slot P(slot) -- pair_sources may have permuted the map in an earlier batched call)
        walk_m(P(slot), n_tx, txs.data(), (const T*)t0_v, n_rx, rxs.data(), tt_out, both, seg_off, segs);
        if (slot_m_off.empty()) { slot_m_off.resize(n_slots); slot_m_j.resize(n_slots); slot_m_v.resize(n_slots); }
        slot_m_off[slot].assign(1, 0); slot_m_j[slot].clear(); slot_m_v[slot].clear();
        assemble_m(n_rx, seg_off, segs, slot_m_off[slot], slot_m_j[slot], slot_m_v[slot]);
        // the rays of the overload that keeps them (already shifted back by the origin of a translated grid)
        if (slot_rays_off.empty()) { slot_rays_off.assign(n_slots, std::vector<long long>{0}); slot_rays_pts.resize(n_slots); }
        if (both) {
            slot_rays_off[slot] = rays_off;
            slot_rays_pts[slot] = rays_pts;
        } else {
            slot_rays_off[slot].assign(1, 0);
            slot_rays_pts[slot].clear();
        }
        rays_off.assign(1, 0);
        rays_pts.clear();
    }
    void slot_m_size(int slot, size_t* n_rows, size_t* nnz) const override {
        check_slot(slot);
        if (slot_m_off.empty() || slot_m_off[slot].empty()) { *n_rows = 0; *nnz = 0; return; }
        *n_rows = slot_m_off[slot].size() - 1;
        *nnz = slot_m_j[slot].size();
    }
    void get_slot_m(int slot, long long* row_off, long long* j, void* v) const override {
        check_slot(slot);
        if (slot_m_off.empty() || slot_m_off[slot].empty()) { row_off[0] = 0; return; }
        std::memcpy(row_off, slot_m_off[slot].data(), slot_m_off[slot].size() * sizeof(long long));
        if (!slot_m_j[slot].empty()) {
            std::memcpy(j, slot_m_j[slot].data(), slot_m_j[slot].size() * sizeof(long long));
            std::memcpy(v, slot_m_v[slot].data(), slot_m_v[slot].size() * sizeof(T));
        }
    }

    // ---- matrix L (the raytrace overloads with l_data, ttcr/Grid2D.h:583-640 -> Grid2Drn::getRaypath(Tx, t0, Rx, [r_data,] l_data,
    // tt, threadNo), ttcr/Grid2Drn.h:1852-2190): per receiver the (cell, length) entries of its ray, sorted by cell with the
    // reference's comparator (CompareSiv_i, ttcr/ttcr_t.h:417-422, std::sort -- entries of one cell keep whatever order that
    // gives them, like in the reference).  The walk is a kernel of its own (fsm_raypath2d_l).
    std::vector<std::vector<long long>> slot_l_off, slot_l_cell;
    std::vector<std::vector<T>> slot_l_val;
    DevBuf<uint32_t> d_lcell;
    DevBuf<T> d_lval;
    DevBuf<int> d_lnum;
    // the walk of the l_data overloads for the receivers of the source whose field lies in `slot`: (cell, length) entries per receiver,
    // sorted like the reference sorts them; the rays of the overload that keeps them
    void walk_l(int slot, int n_tx, const void* tx_v, const void* t0_v, int n_rx, const void* rx_v, void* tt_out_v, bool with_rays,
                std::vector<long long>& loff, std::vector<long long>& lcell, std::vector<T>& lval, std::vector<long long>& roff,
                std::vector<T>& rpts) {
        const int ps = slot;   // (a PHYSICAL slot, like walk_m's: the callers translate)
        loff.assign(1, 0); lcell.clear(); lval.clear();
        roff.assign(1, 0); rpts.clear();
        if (n_rx <= 0) return;
        check_pts((const T*)rx_v, n_rx);
        T* tt_out = (T*)tt_out_v;
        d_rsrc.reserve(2 * (size_t)n_tx);
        d_rt0.reserve(n_tx);
        HIP_CHECK(hipMemcpyAsync(d_rsrc.p, tx_v, sizeof(T) * 2 * n_tx, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(d_rt0.p, t0_v, sizeof(T) * n_tx, hipMemcpyHostToDevice, stream));
        RayGeom2<T> rg2;
        rg2.nnx = ncx + 1; rg2.nnz = ncz + 1;
        rg2.dx = dx; rg2.dz = dz; rg2.xmin = xmin; rg2.zmin = zmin; rg2.xmax = xmax; rg2.zmax = zmax;
        const long max_steps = walk_step_limit;
        long cap = std::min<long>(max_steps, 8L * ((long)ncx + ncz + 3)) + 4;
        const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_rx, ((size_t)256 << 20) / (sizeof(T) * 4 * cap)));
        std::vector<int> st(chunk), np(chunk), nl(chunk);
        std::vector<uint32_t> hc;
        std::vector<T> hv, hp;
        for (int c0 = 0; c0 < n_rx; c0 += chunk) {
            const int m = std::min(chunk, n_rx - c0);
            const T* pc = (const T*)rx_v + 2 * (size_t)c0;
            d_rx.reserve(2 * (size_t)m); d_out.reserve(m); d_rstat.reserve(m); d_raynp.reserve(m); d_lnum.reserve(m);
            HIP_CHECK(hipMemcpyAsync(d_rx.p, pc, sizeof(T) * 2 * m, hipMemcpyHostToDevice, stream));
            for (int attempt = 0;; ++attempt) {
                if (with_rays) d_raypts.reserve((size_t)m * cap * 2);
                d_lcell.reserve((size_t)m * cap); d_lval.reserve((size_t)m * cap);
                const dim3 rgrid((m + 63) / 64), rblock(64);
                if (with_rays)
                    fsm_raypath2d_l<T, true><<<rgrid, rblock, 0, stream>>>(tt_ptr(ps), NS, d_s.p, d_cells.p, rg2, n_tx, d_rsrc.p, d_rt0.p, d_rx.p, m, d_out.p,
                                                                           d_rstat.p, max_steps, d_raypts.p, cap, d_raynp.p, d_lcell.p, d_lval.p, cap, d_lnum.p);
                else
                    fsm_raypath2d_l<T, false><<<rgrid, rblock, 0, stream>>>(tt_ptr(ps), NS, d_s.p, d_cells.p, rg2, n_tx, d_rsrc.p, d_rt0.p, d_rx.p, m, d_out.p,
                                                                            d_rstat.p, max_steps, nullptr, 0, d_raynp.p, d_lcell.p, d_lval.p, cap, d_lnum.p);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipMemcpyAsync(tt_out + c0, d_out.p, sizeof(T) * m, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipMemcpyAsync(st.data(), d_rstat.p, sizeof(int) * m, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipMemcpyAsync(nl.data(), d_lnum.p, sizeof(int) * m, hipMemcpyDeviceToHost, stream));
                if (with_rays) HIP_CHECK(hipMemcpyAsync(np.data(), d_raynp.p, sizeof(int) * m, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                long need = 0;
                for (int q = 0; q < m; ++q) {
                    if (st[q] == 1 || st[q] == 2) throw_walk_error(st[q], pc + 2 * (size_t)q, (const T*)tx_v, max_steps);
                    if (st[q] == 3) need = std::max<long>(need, std::max<long>(nl[q], with_rays ? np[q] : 0) + 1);
                }
                if (need == 0) break;
                if (attempt > 2) throw DeviceError("raypath: a long ray did not retrace to the same length");
                cap = need;   // a ray longer than a row: the chunk once more with the room it asked for
            }
            hc.resize((size_t)m * cap); hv.resize((size_t)m * cap);
            HIP_CHECK(hipMemcpyAsync(hc.data(), d_lcell.p, sizeof(uint32_t) * (size_t)m * cap, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipMemcpyAsync(hv.data(), d_lval.p, sizeof(T) * (size_t)m * cap, hipMemcpyDeviceToHost, stream));
            if (with_rays) { hp.resize((size_t)m * cap * 2); HIP_CHECK(hipMemcpyAsync(hp.data(), d_raypts.p, sizeof(T) * (size_t)m * cap * 2, hipMemcpyDeviceToHost, stream)); }
            HIP_CHECK(hipStreamSynchronize(stream));
            struct Siv { size_t i; T v; };
            std::vector<Siv> row;
            for (int q = 0; q < m; ++q) {
                row.resize(nl[q]);
                for (int e = 0; e < nl[q]; ++e) { row[e].i = hc[(size_t)q * cap + e]; row[e].v = hv[(size_t)q * cap + e]; }
                std::sort(row.begin(), row.end(), [](const Siv n1, const Siv n2) { return n1.i < n2.i; });   // CompareSiv_i
                for (const Siv& e : row) { lcell.push_back((long long)e.i); lval.push_back(e.v); }
                loff.push_back((long long)lcell.size());
                if (with_rays) {
                    rpts.insert(rpts.end(), hp.begin() + (size_t)q * cap * 2, hp.begin() + ((size_t)q * cap + np[q]) * 2);
                    roff.push_back(roff.back() + np[q]);
                }
            }
        }
    }
    void raytrace_l(int slot, int n_tx, const void* tx_v, const void* t0_v, int n_rx, const void* rx_v, void* tt_out_v, bool with_rays) override {
        HIP_CHECK(hipSetDevice(device));
        check_slot(slot);
        if (dim != 2) throw Unsupported("compute_L defined for the FSM");   // (3-D: ttcrpy itself raises, rgrid.pyx:916-917)
        if (!cell) throw Unsupported("compute_L defined only for grids with slowness defined for cells");
        if (n_tx < 1) throw ValueError("every source needs at least one point");
        // the solve (Grid2D::raytrace(Tx, t0, Rx, threadNo)), without receivers: the walk below gives the traveltimes
        {
            const int tx_off[2] = {0, n_tx}, rx_off[2] = {0, 0};
            const int keep_ttrp = ttrp;
            ttrp = 0;
            try { raytrace_multi(1, tx_off, tx_v, t0_v, rx_off, rx_v, tt_out_v, slot); } catch (...) { ttrp = keep_ttrp; throw; }
            ttrp = keep_ttrp;
        }
        if (slot_l_off.empty()) { slot_l_off.assign(n_slots, std::vector<long long>{0}); slot_l_cell.resize(n_slots); slot_l_val.resize(n_slots); }
        if (slot_rays_off.empty()) { slot_rays_off.assign(n_slots, std::vector<long long>{0}); slot_rays_pts.resize(n_slots); }
        walk_l(P(slot), n_tx, tx_v, t0_v, n_rx, rx_v, tt_out_v, with_rays, slot_l_off[slot], slot_l_cell[slot], slot_l_val[slot], slot_rays_off[slot],
               slot_rays_pts[slot]);
    }
    // the l_data overloads for every source of a call (Grid2D's multi-source overloads with l_data run the single-source overload per
    // source on host threads, ttcr/Grid2D.h; ttcrpy: compute_L with several events): batched solves, the walks follow each batch.
    // One CSR over all receiver rows of the call, in row order; the rays (with_rays) as the rays of the call.
    bool l_walk_mode = false, l_walk_rays = false;
    std::vector<std::vector<long long>> l_src_off, l_src_cell, l_src_roff;
    std::vector<std::vector<T>> l_src_val, l_src_rpts;
    std::vector<long long> multi_l_off{0}, multi_l_cell;
    std::vector<T> multi_l_val;
    void raytrace_multi_l(int n_src, const int* tx_off, const void* tx_v, const void* t0_v, const int* rx_off, const void* rx_v,
                          void* tt_out_v, bool with_rays) override {
        if (dim != 2) throw Unsupported("compute_L defined for the FSM");
        if (!cell) throw Unsupported("compute_L defined only for grids with slowness defined for cells");
        multi_l_off.assign(1, 0); multi_l_cell.clear(); multi_l_val.clear();
        rays_off.assign(1, 0); rays_pts.clear();
        if (n_src <= 0) return;
        l_src_off.assign(n_src, {}); l_src_cell.assign(n_src, {}); l_src_roff.assign(n_src, {}); l_src_val.assign(n_src, {}); l_src_rpts.assign(n_src, {});
        const int keep_ttrp = ttrp;
        ttrp = 0;
        l_walk_mode = true; l_walk_rays = with_rays;
        try {
            raytrace_multi(n_src, tx_off, tx_v, t0_v, rx_off, rx_v, tt_out_v, -1);
        } catch (...) { l_walk_mode = false; ttrp = keep_ttrp; throw; }
        l_walk_mode = false; ttrp = keep_ttrp;
        rays_off.assign(1, 0); rays_pts.clear();
        for (int n = 0; n < n_src; ++n) {
            const long long lbase = (long long)multi_l_cell.size();
            for (size_t r = 1; r < l_src_off[n].size(); ++r) multi_l_off.push_back(lbase + l_src_off[n][r]);
            multi_l_cell.insert(multi_l_cell.end(), l_src_cell[n].begin(), l_src_cell[n].end());
            multi_l_val.insert(multi_l_val.end(), l_src_val[n].begin(), l_src_val[n].end());
            if (with_rays) {
                const long long base = rays_off.back();
                for (size_t r = 1; r < l_src_roff[n].size(); ++r) rays_off.push_back(base + l_src_roff[n][r]);
                rays_pts.insert(rays_pts.end(), l_src_rpts[n].begin(), l_src_rpts[n].end());
            }
        }
    }
    void multi_l_size(size_t* n_rows, size_t* nnz) const override { *n_rows = multi_l_off.size() - 1; *nnz = multi_l_cell.size(); }
    void get_multi_l(long long* row_off, long long* cellv, void* v) const override {
        std::memcpy(row_off, multi_l_off.data(), multi_l_off.size() * sizeof(long long));
        if (!multi_l_cell.empty()) {
            std::memcpy(cellv, multi_l_cell.data(), multi_l_cell.size() * sizeof(long long));
            std::memcpy(v, multi_l_val.data(), multi_l_val.size() * sizeof(T));
        }
    }
    void slot_l_size(int slot, size_t* n_rows, size_t* nnz) const override {
        check_slot(slot);
        if (slot_l_off.empty()) { *n_rows = 0; *nnz = 0; return; }
        *n_rows = slot_l_off[slot].size() - 1;
        *nnz = slot_l_cell[slot].size();
    }
    void get_slot_l(int slot, long long* row_off, long long* cellno, void* v) const override {
        check_slot(slot);
        if (slot_l_off.empty()) { row_off[0] = 0; return; }
        std::memcpy(row_off, slot_l_off[slot].data(), slot_l_off[slot].size() * sizeof(long long));
        if (!slot_l_cell[slot].empty()) {
            std::memcpy(cellno, slot_l_cell[slot].data(), slot_l_cell[slot].size() * sizeof(long long));
            std::memcpy(v, slot_l_val[slot].data(), slot_l_val[slot].size() * sizeof(T));
        }
    }

    // rays of the last raytrace_rays call of every slot
    std::vector<std::vector<long long>> slot_rays_off;
    std::vector<std::vector<T>> slot_rays_pts;
    void raytrace_rays(int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx, void* tt_out) override {
        check_slot(slot);
        const int tx_off[2] = {0, n_tx}, rx_off[2] = {0, n_rx};
        raytrace_multi(1, tx_off, tx, t0, rx_off, rx, tt_out, slot, nullptr, true);
        if (slot_rays_off.empty()) { slot_rays_off.assign(n_slots, std::vector<long long>{0}); slot_rays_pts.resize(n_slots); }
        slot_rays_off[slot] = rays_off;
        slot_rays_pts[slot].swap(rays_pts);
        if (!return_rays) { rays_off.assign(1, 0); rays_pts.clear(); } else rays_pts = slot_rays_pts[slot];
    }
    void slot_rays_size(int slot, size_t* n_rays, size_t* n_points) const override {
        check_slot(slot);
        if (slot_rays_off.empty()) { *n_rays = 0; *n_points = 0; return; }
        *n_rays = slot_rays_off[slot].size() - 1;
        *n_points = (size_t)slot_rays_off[slot].back();
    }
    void get_slot_rays(int slot, long long* offsets, void* pts) const override {
        check_slot(slot);
        if (slot_rays_off.empty()) { offsets[0] = 0; return; }
        std::memcpy(offsets, slot_rays_off[slot].data(), slot_rays_off[slot].size() * sizeof(long long));
        if (!slot_rays_pts[slot].empty()) std::memcpy(pts, slot_rays_pts[slot].data(), slot_rays_pts[slot].size() * sizeof(T));
    }
    void rays_size(size_t* n_rays, size_t* n_points) const override {
        *n_rays = rays_off.size() - 1;
        *n_points = (size_t)rays_off.back();
    }
    void get_rays(long long* offsets, void* pts) const override {
        std::memcpy(offsets, rays_off.data(), rays_off.size() * sizeof(long long));
        if (!rays_pts.empty()) std::memcpy(pts, rays_pts.data(), rays_pts.size() * sizeof(T));
    }

    // Grid3D::raytrace multi-source overload (ttcr/Grid3D.h:810-853)
    void validate_points(int n_tx, const void* tx_v, int n_rx, const void* rx_v) override {
        const int nc = ncoord();
        if (n_tx <= 0) throw ValueError("every source needs at least one point");
        std::vector<T> tx((const T*)tx_v, (const T*)tx_v + (size_t)nc * n_tx), rx((const T*)rx_v, (const T*)rx_v + (size_t)nc * std::max(n_rx, 0));
        if (translate) {
            for (int m = 0; m < n_tx; ++m) { tx[3 * m] -= ox; tx[3 * m + 1] -= oy; tx[3 * m + 2] -= oz; }
            for (int m = 0; m < n_rx; ++m) { rx[3 * m] -= ox; rx[3 * m + 1] -= oy; rx[3 * m + 2] -= oz; }
        }
        check_pts(tx.data(), n_tx);
        check_pts(rx.data(), n_rx);
    }

    // Source pairs (NS == 2) share the control flow of the sweep kernel: with exact skipping a chunk is evaluated when EITHER
    // source of the pair needs it, so two sources that lie close to each other -- whose fronts reach the same regions in the
    // same sweeps -- cost fewer evaluations than two that do not (64 random sources on 512^3 nodes: 40.5 % -> 33.9 % of the
    // node updates evaluated, 204 -> 175 ms of sweeps).  The results do not depend on who shares a pair with whom.  Within
    // the batch (logical slots sl[b], sources sr[b]) the physical slots are handed out again: greedy nearest-neighbour pairs
    // (first source point) go to the slot pairs the batch owns completely, the rest to its other slots; `phys` records it.
    void pair_sources(const std::vector<int>& sl, const std::vector<int>& sr, const int* tx_off, const T* tx) {
        const int nb = (int)sl.size();
        if (NS != 2 || !pair_by_distance || nb < 3) return;
        const int nc = ncoord();
        std::vector<int> ps(nb);
        for (int b = 0; b < nb; ++b) ps[b] = P(sl[b]);
        std::sort(ps.begin(), ps.end());
        std::vector<int> pair_slots, single_slots;   // physical slots of complete pairs (2g, 2g+1 both in the batch), the others
        for (int b = 0; b < nb;) {
            if (b + 1 < nb && (ps[b] & 1) == 0 && ps[b + 1] == ps[b] + 1) { pair_slots.push_back(ps[b]); pair_slots.push_back(ps[b + 1]); b += 2; }
            else { single_slots.push_back(ps[b]); b += 1; }
        }
        if (pair_slots.empty()) return;
        auto dist2 = [&](int a, int b) {
            double d = 0;
            for (int c = 0; c < nc; ++c) { const double v = (double)tx[(size_t)nc * tx_off[sr[a]] + c] - (double)tx[(size_t)nc * tx_off[sr[b]] + c]; d += v * v; }
            return d;
        };
        std::vector<char> used(nb, 0);
        std::vector<int> order;   // batch entries: pairs first, then what is left
        order.reserve(nb);
        int n_left = nb;
        for (size_t k = 0; k + 1 < pair_slots.size(); k += 2) {
            int a = 0;
            while (used[a]) ++a;
            used[a] = 1;
            int best = -1;
            double bd = 0;
            for (int b = 0; b < nb; ++b)
                if (!used[b]) { const double d = dist2(a, b); if (best < 0 || d < bd) { best = b; bd = d; } }
            used[best] = 1;
            order.push_back(a); order.push_back(best);
            n_left -= 2;
        }
        for (int b = 0; b < nb && n_left > 0; ++b) if (!used[b]) { order.push_back(b); --n_left; }
        std::vector<int> target(pair_slots);
        target.insert(target.end(), single_slots.begin(), single_slots.end());
        for (int k = 0; k < nb; ++k) phys[sl[order[k]]] = target[k];
    }

    void raytrace_multi(int n_src, const int* tx_off, const void* tx_v, const void* t0_v, const int* rx_off,
                        const void* rx_v, void* tt_out_v, int forced_slot, const int* explicit_slots = nullptr,
                        bool force_rays = false) override {
        HIP_CHECK(hipSetDevice(device));
        if (host_prof) { hp_t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[host] ---- raytrace_multi\n"); }
        const int return_rays = (this->return_rays.load() || force_rays) ? 1 : 0;   // (shadows the option for this call)
        const auto wall0 = std::chrono::steady_clock::now();
        timing = Timing();
        timing.n_sources = n_src;
        rays_off.assign(1, 0);
        rays_pts.clear();
        if (!have_slowness) throw std::runtime_error("Error: slowness has not been assigned.");
        if (n_src <= 0) return;
        if (forced_slot >= 0) {
            check_slot(forced_slot);
            if (n_src != 1) throw ValueError("a thread number can only be given for a single source");
        }
        const int nc = ncoord();
        const int n_tx = tx_off[n_src], n_rx = rx_off[n_src];
        std::vector<T> tx((const T*)tx_v, (const T*)tx_v + (size_t)nc * n_tx);
        std::vector<T> rx((const T*)rx_v, (const T*)rx_v + (size_t)nc * n_rx);
        const T* t0 = (const T*)t0_v;
        T* tt_out = (T*)tt_out_v;
        if (translate) {  // Grid3D::raytrace subtracts the origin (ttcr/Grid3D.h:478-485)
            for (int m = 0; m < n_tx; ++m) { tx[3 * m] -= ox; tx[3 * m + 1] -= oy; tx[3 * m + 2] -= oz; }
            for (int m = 0; m < n_rx; ++m) { rx[3 * m] -= ox; rx[3 * m + 1] -= oy; rx[3 * m + 2] -= oz; }
        }
        for (int n = 0; n < n_src; ++n) {
            if (tx_off[n + 1] <= tx_off[n]) throw ValueError("every source needs at least one point");
            check_pts(tx.data() + (size_t)nc * tx_off[n], tx_off[n + 1] - tx_off[n]);
            check_pts(rx.data() + (size_t)nc * rx_off[n], rx_off[n + 1] - rx_off[n]);
        }
        if (explicit_slots)
            for (int n = 0; n < n_src; ++n) {
                check_slot(explicit_slots[n]);
                if (n > 0 && explicit_slots[n] <= explicit_slots[n - 1]) throw ValueError("explicit slots must be ascending and distinct");
            }
        // block distribution of the sources over the slots: get_blk_size (ttcr/Grid3D.h:451-465)
        const int n_blk = explicit_slots ? n_src : std::min(n_slots, n_src);
        std::vector<int> blk(n_blk, 0);
        for (int n = 0; n < n_src; ++n) blk[n % n_blk] += 1;
        std::vector<int> start(n_blk, 0);
        for (int b = 1; b < n_blk; ++b) start[b] = start[b - 1] + blk[b - 1];
        const int rounds = *std::max_element(blk.begin(), blk.end());
        const int mb = std::max(1, std::min(max_batch > 0 ? max_batch : n_slots, n_slots));
        std::vector<std::vector<long long>> src_ray_len(return_rays ? n_src : 0);
        std::vector<std::vector<T>> src_ray_pts(return_rays ? n_src : 0);
        for (int r = 0; r < rounds; ++r) {
            std::vector<int> slots, srcs;
            for (int b = 0; b < n_blk; ++b)
                if (r < blk[b]) { slots.push_back(forced_slot >= 0 ? forced_slot : (explicit_slots ? explicit_slots[b] : b)); srcs.push_back(start[b] + r); }
            for (size_t c0 = 0; c0 < slots.size(); c0 += mb) {
                const size_t c1 = std::min(slots.size(), c0 + mb);
                std::vector<int> sl(slots.begin() + c0, slots.begin() + c1), sr(srcs.begin() + c0, srcs.begin() + c1);
                // logical -> physical slots; the sources of a block-distributed batch are first paired by distance
                // (also when the caller names the slots -- the replicas of a multi-device grid, single-source calls of several host
                // threads that went to the device together: the permutation stays among the slots of the batch)
                if ((int)sl.size() == n_slots) choose_layout();   // (every slot restarts: the layout may follow the model, see the constructor)
                if (forced_slot < 0) pair_sources(sl, sr, tx_off, tx.data());
                for (int& q : sl) q = P(q);
                {   // (the batch driver takes its entries in ascending slot order)
                    std::vector<int> idx(sl.size());
                    for (size_t q = 0; q < idx.size(); ++q) idx[q] = (int)q;
                    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return sl[a] < sl[b]; });
                    std::vector<int> sl2(sl.size()), sr2(sr.size());
                    for (size_t q = 0; q < idx.size(); ++q) { sl2[q] = sl[idx[q]]; sr2[q] = sr[idx[q]]; }
                    sl.swap(sl2); sr.swap(sr2);
                }
                solve_batch(sl, sr, tx_off, tx.data(), t0);
                hp_mark("solve_batch tail");
                if (l_walk_mode) {
                    // the l_data overloads for every source of the call (raytrace_multi_l): the walk gives traveltimes, entries and rays
                    for (size_t b = 0; b < sl.size(); ++b) {
                        const int n = sr[b];
                        walk_l(sl[b], tx_off[n + 1] - tx_off[n], tx.data() + (size_t)2 * tx_off[n], t0 + tx_off[n], rx_off[n + 1] - rx_off[n],
                               rx.data() + (size_t)2 * rx_off[n], tt_out + rx_off[n], l_walk_rays, l_src_off[n], l_src_cell[n], l_src_val[n],
                               l_src_roff[n], l_src_rpts[n]);
                    }
                } else if (m_walk_mode) {
                    // the m_data overloads for every source of the call (raytrace_multi_m): the rays of the overload that keeps them,
                    // then the walk that leaves the terms of M -- its traveltimes are the call's
                    if (return_rays) raypath_batch_rays(sl, sr, tx_off, tx.data(), t0, rx_off, rx.data(), tt_out, src_ray_len, src_ray_pts);
                    for (size_t b = 0; b < sl.size(); ++b) {
                        const int n = sr[b];
                        walk_m(sl[b], tx_off[n + 1] - tx_off[n], tx.data() + (size_t)3 * tx_off[n], t0 + tx_off[n], rx_off[n + 1] - rx_off[n],
                               rx.data() + (size_t)3 * rx_off[n], tt_out + rx_off[n], m_walk_both, m_seg_off[n], m_segs[n]);
                    }
                } else if (!(ttrp || return_rays)) {
                    interp_batch(sl, sr, rx_off, rx.data(), tt_out);
                    hp_mark("receivers");
                } else if (!return_rays) {
                    raypath_batch(sl, sr, tx_off, tx.data(), t0, rx_off, rx.data(), tt_out);
                } else {
                    raypath_batch_rays(sl, sr, tx_off, tx.data(), t0, rx_off, rx.data(), tt_out, src_ray_len, src_ray_pts);
                }
            }
        }
        if (return_rays) {   // one ray per receiver row, in row order (rows of source 0, then source 1, ...)
            size_t tot = 0;
            for (int n = 0; n < n_src; ++n) tot += src_ray_pts[n].size();
            rays_pts.reserve(tot);
            for (int n = 0; n < n_src; ++n) {
                for (long long len : src_ray_len[n]) rays_off.push_back(rays_off.back() + len);
                rays_pts.insert(rays_pts.end(), src_ray_pts[n].begin(), src_ray_pts[n].end());
                std::vector<T>().swap(src_ray_pts[n]);
            }
        }
        timing.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    }
};

// ---- one grid on several devices ---------------------------------------------------------------------------
// Replaces, for more than one GPU, what Grid3D's multi-source overload does with host threads (ttcr/Grid3D.h:810-853;
// the reference's OpenCL backend keeps one solver per thread slot in one process, ttcr/Grid3Drnfs_OpenCL.h:172-193):
// one replica of the grid per device (slowness replicated, traveltime slots divided), the sources of a call block-
// distributed over the slots exactly like get_blk_size (ttcr/Grid3D.h:451-465), slot s living on device s / spd
// (spd = slots per device), one host thread per device driving its replica, no exchange between devices during a
// solve.  Everything else (getTT, receivers, iteration counts, rays) is forwarded to the replica that owns the slot.
class MultiGrid : public GridBase {
   public:
    std::vector<std::unique_ptr<GridBase>> rep;
    int spd = 1;   // slots per replica (the last one may hold fewer)
    std::vector<long long> rays_off{0};
    std::vector<char> rays_pts;
    size_t pt_bytes = 12;

    template <typename Make>
    MultiGrid(int n_slots_, const std::vector<int>& devs, Make&& make) {
        n_slots = n_slots_;
        const int nd = std::min<int>((int)devs.size(), n_slots);
        spd = (n_slots + nd - 1) / nd;
        for (int r = 0; r * spd < n_slots; ++r) rep.emplace_back(make(std::min(spd, n_slots - r * spd), devs[r]));
        const GridBase& g0 = *rep[0];
        dim = g0.dim; dtype = g0.dtype; device = g0.device; n_nodes = g0.n_nodes; n_cells = g0.n_cells;
        elem_size = g0.elem_size; weno = g0.weno;
        pt_bytes = elem_size * (dim == 3 ? 3 : 2);
        niter.assign(n_slots, 0);
        niterw.assign(n_slots, 0);
        // the slowness of one device is handed to the others device to device
        for (size_t a = 0; a < rep.size(); ++a)
            for (size_t b = 0; b < rep.size(); ++b)
                if (rep[a]->device != rep[b]->device) {
                    (void)hipSetDevice(rep[a]->device);
                    (void)hipDeviceEnablePeerAccess(rep[b]->device, 0);   // (already enabled / not possible: the copy is staged)
                    (void)hipGetLastError();
                }
    }
    GridBase& of(int slot, int& local) const {
        if (slot < 0 || slot >= n_slots) throw ValueError("Thread number is larger than number of threads");
        local = slot % spd;
        return *rep[slot / spd];
    }
    void apply_option(const std::string& k, double value) override {
        GridBase::apply_option(k, value);
        for (auto& r : rep) r->apply_option(k, value);
    }
    void set_slowness(const void* s, size_t n, bool on_device, bool c_order = false) override {
        for (auto& r : rep) r->set_slowness(s, n, on_device, c_order);
    }
    void get_slowness(void* out, size_t n) override { rep[0]->get_slowness(out, n); }
    void validate_points(int n_tx, const void* tx, int n_rx, const void* rx) override { rep[0]->validate_points(n_tx, tx, n_rx, rx); }
    void get_tt(int slot, void* out, size_t n) override { int l; GridBase& g = of(slot, l); g.get_tt(l, out, n); }
    void* tt_device(int slot) override { int l; GridBase& g = of(slot, l); return g.tt_device(l); }
    void* tt_device_view(int slot, size_t* stride) override { int l; GridBase& g = of(slot, l); return g.tt_device_view(l, stride); }
    void interp(int slot, int n, const void* pts, void* out) override { int l; GridBase& g = of(slot, l); g.interp(l, n, pts, out); }
    void compute_slowness(int n, const void* pts, bool translated, void* out) override { rep[0]->compute_slowness(n, pts, translated, out); }
    void get_niter(int slot, int* it, int* itw) const override { int l; GridBase& g = of(slot, l); g.get_niter(l, it, itw); }
    std::string kernel_name() const override { return rep[0]->kernel_name(); }
    long long prefill_swap_count() const override { long long a = 0; for (const auto& r : rep) a += r->prefill_swap_count(); return a; }
    void stopping_stats(long long* sums, long long* missed, long long* rounds) const override {
        long long a = 0, b = 0, c = 0;
        for (const auto& r : rep) { long long x = 0, y = 0, z = 0; r->stopping_stats(&x, &y, &z); a += x; b += y; c += z; }
        if (sums) *sums = a;
        if (missed) *missed = b;
        if (rounds) *rounds = c;
    }
    void reference_change_host(const void* times, const void* field, bool parallel, void* out) override { rep[0]->reference_change_host(times, field, parallel, out); }
    void get_changes(int slot, double* first, int n_first, double* wen, int n_weno) const override {
        int l; GridBase& g = of(slot, l); g.get_changes(l, first, n_first, wen, n_weno);
    }
    void get_reference_changes(int slot, double* first, int n_first, double* wen, int n_weno) const override {
        int l; GridBase& g = of(slot, l); g.get_reference_changes(l, first, n_first, wen, n_weno);
    }
    void rays_size(size_t* n_rays, size_t* n_points) const override { *n_rays = rays_off.size() - 1; *n_points = (size_t)rays_off.back(); }
    void get_rays(long long* offsets, void* pts) const override {
        std::memcpy(offsets, rays_off.data(), rays_off.size() * sizeof(long long));
        if (!rays_pts.empty()) std::memcpy(pts, rays_pts.data(), rays_pts.size());
    }
    void raytrace_rays(int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx, void* tt_out) override {
        int l; GridBase& g = of(slot, l);
        g.raytrace_rays(l, n_tx, tx, t0, n_rx, rx, tt_out);
        timing = g.timing;
    }
    void raytrace_m(int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx, void* tt_out, bool both) override {
        int l; GridBase& g = of(slot, l);
        g.raytrace_m(l, n_tx, tx, t0, n_rx, rx, tt_out, both);
        timing = g.timing;
    }
    void slot_m_size(int slot, size_t* n_rows, size_t* nnz) const override { int l; GridBase& g = of(slot, l); g.slot_m_size(l, n_rows, nnz); }
    // The batched m_data / l_data calls (Grid3D::raytrace with m_data, Grid2D::raytrace with l_data, for every event of a call): the
    // sources go to the replicas in contiguous blocks, in proportion to their slots (every replica then distributes its block over
    // its own slots like a one-device grid), the replicas work side by side, and the CSR rows, rays and traveltimes of the blocks are
    // put together in source order.
    std::vector<long long> mm_off{0}, mm_idx;
    std::vector<char> mm_val;
    template <typename Call, typename Size, typename Get>
    void sharded_matrix_call(int n_src, const int* tx_off, const void* tx_v, const void* t0_v, const int* rx_off, const void* rx_v, void* tt_out_v,
                             Call&& call, Size&& size_of, Get&& get_of) {
        const auto wall0 = std::chrono::steady_clock::now();
        timing = Timing();
        timing.n_sources = n_src;
        mm_off.assign(1, 0); mm_idx.clear(); mm_val.clear();
        rays_off.assign(1, 0); rays_pts.clear();
        if (n_src <= 0) return;
        const char* tx = (const char*)tx_v; const char* t0 = (const char*)t0_v; const char* rx = (const char*)rx_v;
        char* tt_out = (char*)tt_out_v;
        const int nd = (int)rep.size();
        std::vector<int> first(nd + 1, 0);   // sources [first[d], first[d+1]) go to replica d
        {
            long long before = 0;
            for (int d = 0; d < nd; ++d) {
                first[d] = (int)((long long)n_src * before / n_slots);
                before += rep[d]->n_slots;
            }
            first[nd] = n_src;
        }
        struct Part { std::vector<long long> off, idx, roff; std::vector<char> val, rpts; Timing t; };
        std::vector<Part> part(nd);
        std::vector<std::exception_ptr> errs(nd);
        auto work = [&](int d) {
            try {
                const int s0 = first[d], m = first[d + 1] - first[d];
                if (m <= 0) return;
                GridBase& g = *rep[d];
                std::vector<int> to(m + 1), ro(m + 1);
                for (int q = 0; q <= m; ++q) { to[q] = tx_off[s0 + q] - tx_off[s0]; ro[q] = rx_off[s0 + q] - rx_off[s0]; }
                std::vector<char> stt(elem_size * (size_t)std::max(ro[m], 1));
                call(g, m, to.data(), tx + pt_bytes * tx_off[s0], t0 + elem_size * tx_off[s0], ro.data(), rx + pt_bytes * rx_off[s0], stt.data());
                std::memcpy(tt_out + elem_size * rx_off[s0], stt.data(), elem_size * (size_t)ro[m]);
                Part& p = part[d];
                p.t = g.timing;
                size_t nr = 0, nnz = 0;
                size_of(g, &nr, &nnz);
                p.off.assign(nr + 1, 0); p.idx.resize(nnz); p.val.resize(nnz * elem_size);
                if (nr > 0) get_of(g, p.off.data(), p.idx.data(), p.val.data());
                size_t rr = 0, rp = 0;
                g.rays_size(&rr, &rp);
                p.roff.assign(rr + 1, 0); p.rpts.resize(rp * pt_bytes);
                if (rr > 0) g.get_rays(p.roff.data(), p.rpts.data());
            } catch (...) { errs[d] = std::current_exception(); }
        };
        std::vector<std::thread> th;
        for (int d = 1; d < nd; ++d) th.emplace_back(work, d);
        work(0);
        for (auto& t : th) t.join();
        for (auto& e : errs)
            if (e) std::rethrow_exception(e);
        for (int d = 0; d < nd; ++d) {
            const Part& p = part[d];
            if (p.off.empty()) continue;
            const long long base = mm_off.back();
            for (size_t r = 1; r < p.off.size(); ++r) mm_off.push_back(base + p.off[r]);
            mm_idx.insert(mm_idx.end(), p.idx.begin(), p.idx.end());
            mm_val.insert(mm_val.end(), p.val.begin(), p.val.end());
            const long long rbase = rays_off.back();
            for (size_t r = 1; r < p.roff.size(); ++r) rays_off.push_back(rbase + p.roff[r]);
            rays_pts.insert(rays_pts.end(), p.rpts.begin(), p.rpts.end());
            timing.sweep_ms = std::max(timing.sweep_ms, p.t.sweep_ms);   // (the devices run side by side)
            timing.launches += p.t.launches;
            timing.node_updates += p.t.node_updates;
            timing.evaluated_updates += p.t.evaluated_updates;
            timing.iterations = std::max(timing.iterations, p.t.iterations);
        }
        // iteration counts: per slot, as after any call (the last source a slot solved)
        for (int s2 = 0; s2 < n_slots; ++s2) { int l; GridBase& g = of(s2, l); g.get_niter(l, &niter[s2], &niterw[s2]); }
        timing.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    }
    void raytrace_multi_m(int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off, const void* rx, void* tt_out,
                          bool both) override {
        sharded_matrix_call(
            n_src, tx_off, tx, t0, rx_off, rx, tt_out,
            [&](GridBase& g, int m, const int* to, const void* stx, const void* st0, const int* ro, const void* srx, void* stt) {
                g.raytrace_multi_m(m, to, stx, st0, ro, srx, stt, both);
            },
            [](GridBase& g, size_t* nr, size_t* nnz) { g.multi_m_size(nr, nnz); },
            [](GridBase& g, long long* off, long long* idx, void* v) { g.get_multi_m(off, idx, v); });
    }
    void raytrace_multi_l(int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off, const void* rx, void* tt_out,
                          bool with_rays) override {
        sharded_matrix_call(
            n_src, tx_off, tx, t0, rx_off, rx, tt_out,
            [&](GridBase& g, int m, const int* to, const void* stx, const void* st0, const int* ro, const void* srx, void* stt) {
                g.raytrace_multi_l(m, to, stx, st0, ro, srx, stt, with_rays);
            },
            [](GridBase& g, size_t* nr, size_t* nnz) { g.multi_l_size(nr, nnz); },
            [](GridBase& g, long long* off, long long* idx, void* v) { g.get_multi_l(off, idx, v); });
    }
    void multi_l_size(size_t* n_rows, size_t* nnz) const override { *n_rows = mm_off.size() - 1; *nnz = mm_idx.size(); }
    void get_multi_l(long long* row_off, long long* cell, void* v) const override { get_multi_m(row_off, cell, v); }
    void multi_m_size(size_t* n_rows, size_t* nnz) const override { *n_rows = mm_off.size() - 1; *nnz = mm_idx.size(); }
    void get_multi_m(long long* row_off, long long* j, void* v) const override {
        std::memcpy(row_off, mm_off.data(), mm_off.size() * sizeof(long long));
        if (!mm_idx.empty()) {
            std::memcpy(j, mm_idx.data(), mm_idx.size() * sizeof(long long));
            std::memcpy(v, mm_val.data(), mm_val.size());
        }
    }
    void raytrace_l(int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx, void* tt_out, bool with_rays) override {
        int l; GridBase& g = of(slot, l);
        g.raytrace_l(l, n_tx, tx, t0, n_rx, rx, tt_out, with_rays);
        timing = g.timing;
    }
    void slot_l_size(int slot, size_t* n_rows, size_t* nnz) const override { int l; GridBase& g = of(slot, l); g.slot_l_size(l, n_rows, nnz); }
    void get_slot_l(int slot, long long* row_off, long long* cellno, void* v) const override { int l; GridBase& g = of(slot, l); g.get_slot_l(l, row_off, cellno, v); }
    void get_slot_m(int slot, long long* row_off, long long* j, void* v) const override { int l; GridBase& g = of(slot, l); g.get_slot_m(l, row_off, j, v); }
    void slot_rays_size(int slot, size_t* n_rays, size_t* n_points) const override { int l; GridBase& g = of(slot, l); g.slot_rays_size(l, n_rays, n_points); }
    void get_slot_rays(int slot, long long* offsets, void* pts) const override { int l; GridBase& g = of(slot, l); g.get_slot_rays(l, offsets, pts); }

    void raytrace_multi(int n_src, const int* tx_off, const void* tx_v, const void* t0_v, const int* rx_off, const void* rx_v,
                        void* tt_out_v, int forced_slot, const int* explicit_slots = nullptr, bool force_rays = false) override {
        const auto wall0 = std::chrono::steady_clock::now();
        timing = Timing();
        timing.n_sources = n_src;
        rays_off.assign(1, 0);
        rays_pts.clear();
        if (n_src <= 0) return;
        const bool want_rays = return_rays.load() || force_rays;
        const char* tx = (const char*)tx_v; const char* t0 = (const char*)t0_v; const char* rx = (const char*)rx_v;
        char* tt_out = (char*)tt_out_v;
        if (forced_slot >= 0) {
            if (n_src != 1) throw ValueError("a thread number can only be given for a single source");
            int l; GridBase& g = of(forced_slot, l);
            g.raytrace_multi(n_src, tx_off, tx_v, t0_v, rx_off, rx_v, tt_out_v, l, nullptr, force_rays);
            timing = g.timing;
            if (want_rays) {
                size_t nr = 0, np = 0;
                g.rays_size(&nr, &np);
                rays_off.resize(nr + 1);
                rays_pts.resize(np * pt_bytes);
                g.get_rays(rays_off.data(), rays_pts.data());
            }
            return;
        }
        // every point of the call is checked before anything is solved, like on one device
        for (int n = 0; n < n_src; ++n) {
            if (tx_off[n + 1] <= tx_off[n]) throw ValueError("every source needs at least one point");
            rep[0]->validate_points(tx_off[n + 1] - tx_off[n], tx + pt_bytes * tx_off[n], rx_off[n + 1] - rx_off[n], rx + pt_bytes * rx_off[n]);
        }
        // (source, slot, round): the block distribution of get_blk_size over ALL slots, or the slots the caller names
        struct Item { int src, slot, round; };
        std::vector<Item> items;
        if (explicit_slots) {
            for (int n = 0; n < n_src; ++n) {
                if (explicit_slots[n] < 0 || explicit_slots[n] >= n_slots) throw ValueError("Thread number is larger than number of threads");
                if (n > 0 && explicit_slots[n] <= explicit_slots[n - 1]) throw ValueError("explicit slots must be ascending and distinct");
                items.push_back({n, explicit_slots[n], 0});
            }
        } else {
            const int n_blk = std::min(n_slots, n_src);
            std::vector<int> blk(n_blk, 0), start(n_blk, 0);
            for (int n = 0; n < n_src; ++n) blk[n % n_blk] += 1;
            for (int b = 1; b < n_blk; ++b) start[b] = start[b - 1] + blk[b - 1];
            for (int b = 0; b < n_blk; ++b)
                for (int r = 0; r < blk[b]; ++r) items.push_back({start[b] + r, b, r});
        }
        int rounds = 0;
        for (const Item& it : items) rounds = std::max(rounds, it.round + 1);
        std::vector<std::vector<long long>> src_ray_len(want_rays ? n_src : 0);
        std::vector<std::vector<char>> src_ray_pts(want_rays ? n_src : 0);
        std::vector<std::exception_ptr> errs(rep.size());
        std::vector<Timing> dev_t(rep.size());
        auto work = [&](int d) {
            try {
                GridBase& g = *rep[d];
                for (int r = 0; r < rounds; ++r) {
                    std::vector<Item> mine;
                    for (const Item& it : items)
                        if (it.round == r && it.slot / spd == d) mine.push_back(it);
                    if (mine.empty()) continue;
                    std::sort(mine.begin(), mine.end(), [](const Item& a, const Item& b) { return a.slot < b.slot; });
                    const int m = (int)mine.size();
                    std::vector<int> to(m + 1, 0), ro(m + 1, 0), ls(m);
                    for (int q = 0; q < m; ++q) {
                        to[q + 1] = to[q] + tx_off[mine[q].src + 1] - tx_off[mine[q].src];
                        ro[q + 1] = ro[q] + rx_off[mine[q].src + 1] - rx_off[mine[q].src];
                        ls[q] = mine[q].slot % spd;
                    }
                    std::vector<char> stx(pt_bytes * to[m]), st0(elem_size * to[m]), srx(pt_bytes * std::max(ro[m], 1)), stt(elem_size * std::max(ro[m], 1));
                    for (int q = 0; q < m; ++q) {
                        const int n = mine[q].src;
                        std::memcpy(stx.data() + pt_bytes * to[q], tx + pt_bytes * tx_off[n], pt_bytes * (to[q + 1] - to[q]));
                        std::memcpy(st0.data() + elem_size * to[q], t0 + elem_size * tx_off[n], elem_size * (to[q + 1] - to[q]));
                        std::memcpy(srx.data() + pt_bytes * ro[q], rx + pt_bytes * rx_off[n], pt_bytes * (ro[q + 1] - ro[q]));
                    }
                    g.raytrace_multi(m, to.data(), stx.data(), st0.data(), ro.data(), srx.data(), stt.data(), -1, ls.data(), want_rays);
                    dev_t[d].sweep_ms += g.timing.sweep_ms;
                    dev_t[d].launches += g.timing.launches;
                    dev_t[d].node_updates += g.timing.node_updates;
                    dev_t[d].evaluated_updates += g.timing.evaluated_updates;
                    dev_t[d].iterations = std::max(dev_t[d].iterations, g.timing.iterations);
                    for (int q = 0; q < m; ++q)
                        std::memcpy(tt_out + elem_size * rx_off[mine[q].src], stt.data() + elem_size * ro[q], elem_size * (ro[q + 1] - ro[q]));
                    if (want_rays) {   // the replica's rays are in the row order of ITS call
                        size_t nr = 0, np = 0;
                        g.rays_size(&nr, &np);
                        std::vector<long long> off(nr + 1);
                        std::vector<char> pts(np * pt_bytes);
                        g.get_rays(off.data(), pts.data());
                        for (int q = 0; q < m; ++q) {
                            const int n = mine[q].src;
                            for (int k = ro[q]; k < ro[q + 1]; ++k) src_ray_len[n].push_back(off[k + 1] - off[k]);
                            src_ray_pts[n].assign(pts.begin() + off[ro[q]] * pt_bytes, pts.begin() + off[ro[q + 1]] * pt_bytes);
                        }
                    }
                }
            } catch (...) { errs[d] = std::current_exception(); }
        };
        std::vector<std::thread> th;
        for (size_t d = 1; d < rep.size(); ++d) th.emplace_back(work, (int)d);
        work(0);
        for (auto& t : th) t.join();
        for (auto& e : errs)
            if (e) std::rethrow_exception(e);
        for (int s2 = 0; s2 < n_slots; ++s2) { int l; GridBase& g = of(s2, l); g.get_niter(l, &niter[s2], &niterw[s2]); }
        for (const Timing& t : dev_t) {   // the devices run side by side: times are the slowest one's, counts add up
            timing.sweep_ms = std::max(timing.sweep_ms, t.sweep_ms);
            timing.launches += t.launches;
            timing.node_updates += t.node_updates;
            timing.evaluated_updates += t.evaluated_updates;
            timing.iterations = std::max(timing.iterations, t.iterations);
        }
        if (want_rays)
            for (int n = 0; n < n_src; ++n) {
                for (long long len : src_ray_len[n]) rays_off.push_back(rays_off.back() + len);
                rays_pts.insert(rays_pts.end(), src_ray_pts[n].begin(), src_ray_pts[n].end());
            }
        timing.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    }
};

}  // namespace ttcr_amd

// =============================================================================== C ABI
using namespace ttcr_amd;

struct ttcr_fsm_grid {
    std::unique_ptr<GridBase> impl;
};

template <typename F>
static int guarded(F&& f) {
    try {
        f();
        return TTCR_OK;
    } catch (const ValueError& e) {
        g_last_error = e.what();
        return TTCR_ERR_VALUE;
    } catch (const DeviceError& e) {
        g_last_error = e.what();
        return TTCR_ERR_DEVICE;
    } catch (const Unsupported& e) {
        g_last_error = e.what();
        return TTCR_ERR_UNSUPPORTED;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return TTCR_ERR_RUNTIME;
    }
}

// entry points that touch a grid: serialised per handle (ttcrpy's raytrace(..., thread_no=k) lets several host
// threads work on one grid; here the slots share one stream, one graph and the pinned / scratch buffers)
template <typename G, typename F>
static int guarded_on(G* g, F&& f) {
    if (!g) {
        g_last_error = "null grid handle";
        return TTCR_ERR_VALUE;
    }
    std::lock_guard<std::mutex> lock(g->impl->mu);
    return guarded(std::forward<F>(f));
}

static int pick_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        throw DeviceError("no HIP device available: the MI355X FSM backend has no CPU fallback");
    if (device < 0) {
        int cur = 0;
        HIP_CHECK(hipGetDevice(&cur));
        return cur;
    }
    if (device >= n) throw ValueError("device ordinal out of range");
    return device;
}

// device list of a grid: the caller's, else TTCR_AMD_DEVICES ("0,1,2,3") when the caller asks for "the current device"
// (an unmodified ttcrpy script is spread over the GPUs of a node by setting the variable), else the one device
static std::vector<int> device_list(int device, const int* devices, int n_devices) {
    std::vector<int> devs;
    if (devices && n_devices > 0) {
        for (int q = 0; q < n_devices; ++q) devs.push_back(pick_device(devices[q]));
    } else if (device < 0 && std::getenv("TTCR_AMD_DEVICES") && *std::getenv("TTCR_AMD_DEVICES")) {
        std::stringstream ss(std::getenv("TTCR_AMD_DEVICES"));
        std::string tok;
        while (std::getline(ss, tok, ',')) {
            if (tok.empty()) continue;
            char* endp = nullptr;
            const long v = std::strtol(tok.c_str(), &endp, 10);
            if (*endp) throw ValueError("TTCR_AMD_DEVICES: expected a comma-separated list of device ordinals");
            devs.push_back(pick_device((int)v));
        }
    }
    if (devs.empty()) devs.push_back(pick_device(device));
    return devs;
}

template <typename Make>
static std::unique_ptr<GridBase> make_grid(int n_slots, const std::vector<int>& devs, Make&& make) {
    if (devs.size() == 1 || n_slots == 1) return std::unique_ptr<GridBase>(make(n_slots, devs[0]));
    return std::unique_ptr<GridBase>(new MultiGrid(n_slots, devs, make));
}

static void create3d(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncy, uint32_t ncz, double dx,
                     double xmin, double ymin, double zmin, double eps, int maxit, int weno, int translate_origin, int n_slots,
                     int device, const int* devices, int n_devices) {
    if (!out) throw ValueError("null output handle");
    *out = nullptr;
    if (dtype != TTCR_F32 && dtype != TTCR_F64) throw ValueError("dtype must be TTCR_F32 or TTCR_F64");
    if (ncx < 1 || ncy < 1 || ncz < 1) throw ValueError("grid needs at least one cell per axis");
    if (n_slots < 1) throw ValueError("n_slots must be >= 1");
    if (!(dx > 0)) throw ValueError("dx must be positive");
    if (weno && (ncx < 3 || ncy < 3 || ncz < 3)) throw ValueError("weno=True needs at least 3 cells per axis");
    const std::vector<int> devs = device_list(device, devices, n_devices);
    auto g = std::make_unique<ttcr_fsm_grid>();
    auto make = [&](int ns, int dev) -> GridBase* {
        GridBase* r;
        if (dtype == TTCR_F32)
            r = new GridT<float>(3, cell_slowness != 0, ncx, ncy, ncz, dx, dx, xmin, ymin, zmin, eps, maxit, ns, translate_origin != 0, dev, weno != 0);
        else
#ifdef FSM_DEV_F32_ONLY   // tuning builds: half the instantiations
            throw ValueError("tuning build without double grids");
#else
            r = new GridT<double>(3, cell_slowness != 0, ncx, ncy, ncz, dx, dx, xmin, ymin, zmin, eps, maxit, ns, translate_origin != 0, dev, weno != 0);
#endif
        r->weno = weno != 0;
        return r;
    };
    g->impl = make_grid(n_slots, devs, make);
    g->impl->weno = weno != 0;
    *out = g.release();
}

static void create2d(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncz, double dx, double dz,
                     double xmin, double zmin, double eps, int maxit, int weno, int rotated_template, int n_slots, int device,
                     const int* devices, int n_devices) {
    if (!out) throw ValueError("null output handle");
    *out = nullptr;
    if (dtype != TTCR_F32 && dtype != TTCR_F64) throw ValueError("dtype must be TTCR_F32 or TTCR_F64");
    if (ncx < 1 || ncz < 1) throw ValueError("grid needs at least one cell per axis");
    if (n_slots < 1) throw ValueError("n_slots must be >= 1");
    if (!(dx > 0) || !(dz > 0)) throw ValueError("dx and dz must be positive");
    if (weno && (ncx < 3 || ncz < 3)) throw ValueError("weno=True needs at least 3 cells per axis");
    const std::vector<int> devs = device_list(device, devices, n_devices);
    auto g = std::make_unique<ttcr_fsm_grid>();
    auto make = [&](int ns, int dev) -> GridBase* {
        GridBase* r;
        if (dtype == TTCR_F32)
            r = new GridT<float>(2, cell_slowness != 0, ncx, 0, ncz, dx, dz, xmin, 0.0, zmin, eps, maxit, ns, false, dev, weno != 0);
        else
#ifdef FSM_DEV_F32_ONLY
            throw ValueError("tuning build without double grids");
#else
            r = new GridT<double>(2, cell_slowness != 0, ncx, 0, ncz, dx, dz, xmin, 0.0, zmin, eps, maxit, ns, false, dev, weno != 0);
#endif
        r->weno = weno != 0;
        r->rotated = rotated_template != 0;
        return r;
    };
    g->impl = make_grid(n_slots, devs, make);
    g->impl->weno = weno != 0;
    g->impl->rotated = rotated_template != 0;
    *out = g.release();
}

extern "C" {

int ttcr_fsm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ttcr_fsm_last_error(void) { return g_last_error.c_str(); }

int ttcr_fsm3d_create(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncy, uint32_t ncz, double dx,
                      double xmin, double ymin, double zmin, double eps, int maxit, int weno, int n_slots, int translate_origin,
                      int device) {
    return guarded([&] { create3d(out, dtype, cell_slowness, ncx, ncy, ncz, dx, xmin, ymin, zmin, eps, maxit, weno, translate_origin, n_slots, device, nullptr, 0); });
}
int ttcr_fsm3d_create_multi(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncy, uint32_t ncz, double dx,
                            double xmin, double ymin, double zmin, double eps, int maxit, int weno, int n_slots, int translate_origin,
                            const int* devices, int n_devices) {
    return guarded([&] {
        if (!devices || n_devices < 1) throw ValueError("device list is empty");
        create3d(out, dtype, cell_slowness, ncx, ncy, ncz, dx, xmin, ymin, zmin, eps, maxit, weno, translate_origin, n_slots, -1, devices, n_devices);
    });
}

int ttcr_fsm2d_create(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncz, double dx, double dz,
                      double xmin, double zmin, double eps, int maxit, int weno, int rotated_template, int n_slots, int device) {
    return guarded([&] { create2d(out, dtype, cell_slowness, ncx, ncz, dx, dz, xmin, zmin, eps, maxit, weno, rotated_template, n_slots, device, nullptr, 0); });
}
int ttcr_fsm2d_create_multi(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncz, double dx, double dz,
                            double xmin, double zmin, double eps, int maxit, int weno, int rotated_template, int n_slots,
                            const int* devices, int n_devices) {
    return guarded([&] {
        if (!devices || n_devices < 1) throw ValueError("device list is empty");
        create2d(out, dtype, cell_slowness, ncx, ncz, dx, dz, xmin, zmin, eps, maxit, weno, rotated_template, n_slots, -1, devices, n_devices);
    });
}
int ttcr_fsm_n_devices(const ttcr_fsm_grid* g) {
    if (!g) return 0;
    const MultiGrid* m = dynamic_cast<const MultiGrid*>(g->impl.get());
    return m ? (int)m->rep.size() : 1;
}

void ttcr_fsm_destroy(ttcr_fsm_grid* g) { delete g; }

int ttcr_fsm_set_slowness(ttcr_fsm_grid* g, const void* s, size_t n) {
    return guarded_on(g, [&] { g->impl->set_slowness(s, n, false); });
}
int ttcr_fsm_set_slowness_c_order(ttcr_fsm_grid* g, const void* s, size_t n) {
    return guarded_on(g, [&] { g->impl->set_slowness(s, n, false, true); });
}
int ttcr_fsm_set_slowness_device(ttcr_fsm_grid* g, const void* d_s, size_t n) {
    return guarded_on(g, [&] { g->impl->set_slowness(d_s, n, true); });
}
int ttcr_fsm_get_slowness(ttcr_fsm_grid* g, void* out, size_t n) {
    return guarded_on(g, [&] { g->impl->get_slowness(out, n); });
}

// One batch of combined requests (distinct slots): every request is validated on its own (a point outside the grid fails
// that call only), the valid ones go to the device as ONE multi-source solve with their slots, results are copied back.
static void run_combined(GridBase* gb, std::vector<GridBase::Request*>& batch) {
    std::vector<GridBase::Request*> ok;
    for (auto* r : batch) {
        try {
            if (r->slot < 0 || r->slot >= gb->n_slots) throw ValueError("Thread number is larger than number of threads");
            gb->validate_points(r->n_tx, r->tx, r->n_rx, r->rx);
            ok.push_back(r);
        } catch (const ValueError& e) { r->status = TTCR_ERR_VALUE; r->err = e.what();
        } catch (const std::exception& e) { r->status = TTCR_ERR_RUNTIME; r->err = e.what(); }
    }
    if (ok.empty()) return;
    std::sort(ok.begin(), ok.end(), [](const GridBase::Request* a, const GridBase::Request* b) { return a->slot < b->slot; });
    const size_t es = gb->elem_size, nc = gb->dim == 3 ? 3 : 2;
    std::vector<int> tx_off(ok.size() + 1, 0), rx_off(ok.size() + 1, 0), slots(ok.size());
    for (size_t n = 0; n < ok.size(); ++n) {
        tx_off[n + 1] = tx_off[n] + ok[n]->n_tx;
        rx_off[n + 1] = rx_off[n] + std::max(ok[n]->n_rx, 0);
        slots[n] = ok[n]->slot;
    }
    std::vector<char> tx(es * nc * tx_off.back()), t0(es * tx_off.back()), rx(es * nc * std::max(rx_off.back(), 1)), tt(es * std::max(rx_off.back(), 1));
    for (size_t n = 0; n < ok.size(); ++n) {
        std::memcpy(tx.data() + es * nc * tx_off[n], ok[n]->tx, es * nc * ok[n]->n_tx);
        std::memcpy(t0.data() + es * tx_off[n], ok[n]->t0, es * ok[n]->n_tx);
        if (ok[n]->n_rx > 0) std::memcpy(rx.data() + es * nc * rx_off[n], ok[n]->rx, es * nc * ok[n]->n_rx);
    }
    int st = TTCR_OK;
    std::string msg;
    try {
        gb->raytrace_multi((int)ok.size(), tx_off.data(), tx.data(), t0.data(), rx_off.data(), rx.data(), tt.data(), -1, slots.data());
    } catch (const ValueError& e) { st = TTCR_ERR_VALUE; msg = e.what();
    } catch (const DeviceError& e) { st = TTCR_ERR_DEVICE; msg = e.what();
    } catch (const std::exception& e) { st = TTCR_ERR_RUNTIME; msg = e.what(); }
    if (st != TTCR_OK && st != TTCR_ERR_DEVICE && ok.size() > 1) {
        // something only the solve itself finds out (a ray that leaves the grid, ...) failed the batch: the calls are
        // independent in the reference, so each one is redone on its own and only the offending call reports the error
        for (auto* r : ok) {
            const int to[2] = {0, r->n_tx}, ro[2] = {0, std::max(r->n_rx, 0)};
            try {
                gb->raytrace_multi(1, to, r->tx, r->t0, ro, r->rx, r->tt, r->slot);
                r->status = TTCR_OK;
            } catch (const ValueError& e) { r->status = TTCR_ERR_VALUE; r->err = e.what();
            } catch (const DeviceError& e) { r->status = TTCR_ERR_DEVICE; r->err = e.what();
            } catch (const std::exception& e) { r->status = TTCR_ERR_RUNTIME; r->err = e.what(); }
        }
        return;
    }
    for (size_t n = 0; n < ok.size(); ++n) {
        ok[n]->status = st;
        ok[n]->err = msg;
        if (st == TTCR_OK && ok[n]->n_rx > 0) std::memcpy(ok[n]->tt, tt.data() + es * rx_off[n], es * ok[n]->n_rx);
    }
}

int ttcr_fsm_raytrace(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                      void* tt_out) {
    if (!g) { g_last_error = "null grid handle"; return TTCR_ERR_VALUE; }
    GridBase* gb = g->impl.get();
    auto alone = [&] {
        return guarded_on(g, [&] {
            if (slot < 0 || slot >= g->impl->n_slots) throw ValueError("Thread number is larger than number of threads");
            const int tx_off[2] = {0, n_tx}, rx_off[2] = {0, n_rx};
            g->impl->raytrace_multi(1, tx_off, tx, t0, rx_off, rx, tt_out, slot);
        });
    };
    if (gb->n_slots <= 1 || gb->combine_window_us.load() <= 0 || gb->return_rays.load()) return alone();   // (rays belong to "the last call": no company)
    // Grid3D's multi-source overload (ttcr/Grid3D.h:810-853) reaches a backend as nt host threads, each calling the
    // single-source raytrace with its own threadNo.  Calls that arrive within a short window are gathered by the first
    // one (the leader) and go to the device as ONE batch -- the sources then run side by side like in
    // ttcr_fsm_raytrace_multi instead of one after the other.  A handle that has only ever seen one call at a time (a
    // plain loop over sources) never waits for company.
    GridBase::Request req{slot, n_tx, n_rx, tx, t0, rx, tt_out};
    try {
        std::unique_lock<std::mutex> lk(gb->q_mu);
        if (!gb->queue.empty() || gb->leader_active) gb->had_company = true;
        gb->queue.push_back(&req);
        gb->q_cv.notify_all();
        while (!req.done) {
            if (gb->leader_active) { gb->q_cv.wait(lk); continue; }
            gb->leader_active = true;
            std::vector<GridBase::Request*> batch;
            try {
                if (gb->had_company) {
                    const auto window = std::chrono::microseconds(gb->combine_window_us.load());
                    size_t seen = gb->queue.size();
                    while ((int)gb->queue.size() < gb->n_slots) {   // wait for company until nothing new arrives for one window
                        gb->q_cv.wait_for(lk, window);
                        if (gb->queue.size() == seen) break;
                        seen = gb->queue.size();
                    }
                }
                std::vector<GridBase::Request*> rest;
                std::vector<char> taken(gb->n_slots > 0 ? gb->n_slots : 1, 0);
                for (auto* r : gb->queue) {   // one request per slot and batch; a second one for the same slot waits for the next round
                    const bool dup = r->slot >= 0 && r->slot < gb->n_slots && taken[r->slot];
                    if (dup) { rest.push_back(r); continue; }
                    if (r->slot >= 0 && r->slot < gb->n_slots) taken[r->slot] = 1;
                    batch.push_back(r);
                }
                gb->queue.swap(rest);
                lk.unlock();
                try {
                    std::lock_guard<std::mutex> hl(gb->mu);
                    run_combined(gb, batch);
                } catch (...) { lk.lock(); throw; }
                lk.lock();
            } catch (...) {
                // (an allocation failed somewhere above: nobody may be left waiting for a leader that is gone)
                if (batch.empty()) { batch.swap(gb->queue); }
                // (thrown while the batch was being picked: its requests are still queued -- they are marked done below and their
                // callers return, so the queue must not keep pointers to them; remove / erase do not allocate)
                gb->queue.erase(std::remove_if(gb->queue.begin(), gb->queue.end(),
                                               [&](GridBase::Request* q) { return std::find(batch.begin(), batch.end(), q) != batch.end(); }),
                                gb->queue.end());
                for (auto* r : batch)
                    if (r->status == TTCR_OK) { r->status = TTCR_ERR_RUNTIME; r->err = "out of memory while combining raytrace calls"; }
            }
            for (auto* r : batch) r->done = true;
            gb->leader_active = false;
            gb->q_cv.notify_all();
        }
    } catch (...) {
        g_last_error = "out of memory while queueing a raytrace call";
        return TTCR_ERR_RUNTIME;
    }
    if (req.status != TTCR_OK) g_last_error = req.err;
    return req.status;
}

int ttcr_fsm_raytrace_rays(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                           void* tt_out) {
    return guarded_on(g, [&] { g->impl->raytrace_rays(slot, n_tx, tx, t0, n_rx, rx, tt_out); });
}
int ttcr_fsm_raytrace_m(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                        void* tt_out) {
    return guarded_on(g, [&] { g->impl->raytrace_m(slot, n_tx, tx, t0, n_rx, rx, tt_out, false); });
}
int ttcr_fsm_raytrace_rm(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                        void* tt_out) {
    return guarded_on(g, [&] { g->impl->raytrace_m(slot, n_tx, tx, t0, n_rx, rx, tt_out, true); });
}
int ttcr_fsm_raytrace_multi_m(ttcr_fsm_grid* g, int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off,
                              const void* rx, void* tt_out, int with_rays) {
    return guarded_on(g, [&] { g->impl->raytrace_multi_m(n_src, tx_off, tx, t0, rx_off, rx, tt_out, with_rays != 0); });
}
int ttcr_fsm_raytrace_multi_l(ttcr_fsm_grid* g, int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off,
                              const void* rx, void* tt_out, int with_rays) {
    return guarded_on(g, [&] { g->impl->raytrace_multi_l(n_src, tx_off, tx, t0, rx_off, rx, tt_out, with_rays != 0); });
}
int ttcr_fsm_multi_l_size(const ttcr_fsm_grid* g, size_t* n_rows, size_t* nnz) {
    return guarded_on(g, [&] {
        if (!n_rows || !nnz) throw ValueError("null output pointer");
        g->impl->multi_l_size(n_rows, nnz);
    });
}
int ttcr_fsm_get_multi_l(const ttcr_fsm_grid* g, long long* row_off, long long* cell, void* v) {
    return guarded_on(g, [&] { g->impl->get_multi_l(row_off, cell, v); });
}
int ttcr_fsm_multi_m_size(const ttcr_fsm_grid* g, size_t* n_rows, size_t* nnz) {
    return guarded_on(g, [&] {
        if (!n_rows || !nnz) throw ValueError("null output pointer");
        g->impl->multi_m_size(n_rows, nnz);
    });
}
int ttcr_fsm_get_multi_m(const ttcr_fsm_grid* g, long long* row_off, long long* j, void* v) {
    return guarded_on(g, [&] { g->impl->get_multi_m(row_off, j, v); });
}
int ttcr_fsm_slot_m_size(const ttcr_fsm_grid* g, int slot, size_t* n_rows, size_t* nnz) {
    return guarded_on(g, [&] {
        if (!n_rows || !nnz) throw ValueError("null output pointer");
        g->impl->slot_m_size(slot, n_rows, nnz);
    });
}
int ttcr_fsm_get_slot_m(const ttcr_fsm_grid* g, int slot, long long* row_off, long long* j, void* v) {
    return guarded_on(g, [&] { g->impl->get_slot_m(slot, row_off, j, v); });
}
int ttcr_fsm_raytrace_l(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                        void* tt_out, int with_rays) {
    return guarded_on(g, [&] { g->impl->raytrace_l(slot, n_tx, tx, t0, n_rx, rx, tt_out, with_rays != 0); });
}
int ttcr_fsm_slot_l_size(const ttcr_fsm_grid* g, int slot, size_t* n_rows, size_t* nnz) {
    return guarded_on(g, [&] {
        if (!n_rows || !nnz) throw ValueError("null output pointer");
        g->impl->slot_l_size(slot, n_rows, nnz);
    });
}
int ttcr_fsm_get_slot_l(const ttcr_fsm_grid* g, int slot, long long* row_off, long long* cell, void* v) {
    return guarded_on(g, [&] { g->impl->get_slot_l(slot, row_off, cell, v); });
}
int ttcr_fsm_slot_rays_size(const ttcr_fsm_grid* g, int slot, size_t* n_rays, size_t* n_points) {
    return guarded_on(g, [&] {
        if (!n_rays || !n_points) throw ValueError("null output pointer");
        g->impl->slot_rays_size(slot, n_rays, n_points);
    });
}
int ttcr_fsm_get_slot_rays(const ttcr_fsm_grid* g, int slot, long long* offsets, void* pts) {
    return guarded_on(g, [&] { g->impl->get_slot_rays(slot, offsets, pts); });
}

int ttcr_fsm_raytrace_multi(ttcr_fsm_grid* g, int n_src, const int* tx_off, const void* tx, const void* t0,
                            const int* rx_off, const void* rx, void* tt_out) {
    return guarded_on(g, [&] { g->impl->raytrace_multi(n_src, tx_off, tx, t0, rx_off, rx, tt_out, -1); });
}

int ttcr_fsm_get_tt(ttcr_fsm_grid* g, int slot, void* out, size_t n) {
    return guarded_on(g, [&] { g->impl->get_tt(slot, out, n); });
}
int ttcr_fsm_get_tt_device(ttcr_fsm_grid* g, int slot, void** d_ptr) {
    return guarded_on(g, [&] {
        if (!d_ptr) throw ValueError("null output pointer");
        *d_ptr = g->impl->tt_device(slot);
    });
}
int ttcr_fsm_get_tt_device_view(ttcr_fsm_grid* g, int slot, void** d_ptr, size_t* stride) {
    return guarded_on(g, [&] {
        if (!d_ptr || !stride) throw ValueError("null output pointer");
        *d_ptr = g->impl->tt_device_view(slot, stride);
    });
}
int ttcr_fsm_interp(ttcr_fsm_grid* g, int slot, int n_pts, const void* pts, void* tt_out) {
    return guarded_on(g, [&] { g->impl->interp(slot, n_pts, pts, tt_out); });
}
int ttcr_fsm_compute_slowness(ttcr_fsm_grid* g, int n_pts, const void* pts, int translated, void* out) {
    return guarded_on(g, [&] { g->impl->compute_slowness(n_pts, pts, translated != 0, out); });
}
int ttcr_fsm_get_niter(ttcr_fsm_grid* g, int slot, int* niter, int* niterw) {
    return guarded_on(g, [&] { g->impl->get_niter(slot, niter, niterw); });
}
int ttcr_fsm_get_changes(ttcr_fsm_grid* g, int slot, double* first_order, int n_first, double* weno, int n_weno) {
    return guarded_on(g, [&] {
        if ((n_first > 0 && !first_order) || (n_weno > 0 && !weno) || n_first < 0 || n_weno < 0) throw ValueError("bad output buffers");
        g->impl->get_changes(slot, first_order, n_first, weno, n_weno);
    });
}
int ttcr_fsm_get_reference_changes(ttcr_fsm_grid* g, int slot, double* first_order, int n_first, double* weno, int n_weno) {
    return guarded_on(g, [&] {
        if ((n_first > 0 && !first_order) || (n_weno > 0 && !weno) || n_first < 0 || n_weno < 0) throw ValueError("bad output buffers");
        g->impl->get_reference_changes(slot, first_order, n_first, weno, n_weno);
    });
}
int ttcr_fsm_n_slots(const ttcr_fsm_grid* g) { return g->impl->n_slots; }
size_t ttcr_fsm_n_nodes(const ttcr_fsm_grid* g) { return g->impl->n_nodes; }
size_t ttcr_fsm_n_cells(const ttcr_fsm_grid* g) { return g->impl->n_cells; }

int ttcr_fsm_set_option(ttcr_fsm_grid* g, const char* key, double value) {
    return guarded_on(g, [&] { g->impl->apply_option(std::string(key ? key : ""), value); });
}

int ttcr_fsm_rays_size(const ttcr_fsm_grid* g, size_t* n_rays, size_t* n_points) {
    return guarded_on(g, [&] { g->impl->rays_size(n_rays, n_points); });
}
int ttcr_fsm_get_rays(const ttcr_fsm_grid* g, long long* offsets, void* pts) {
    return guarded_on(g, [&] { g->impl->get_rays(offsets, pts); });
}

int ttcr_fsm_last_timing(const ttcr_fsm_grid* g, ttcr_fsm_timing* out) {
    return guarded_on(g, [&] {
        const Timing& t = g->impl->timing;
        out->sweep_ms = t.sweep_ms;
        out->total_ms = t.total_ms;
        out->kernel_launches = t.launches;
        out->node_updates = t.node_updates;
        out->evaluated_updates = t.evaluated_updates;
        out->iterations = t.iterations;
        out->n_sources = t.n_sources;
    });
}

int ttcr_fsm_stopping_stats(const ttcr_fsm_grid* g, long long* reference_sums, long long* reference_sums_missed, long long* rounds) {
    return guarded_on(g, [&] { g->impl->stopping_stats(reference_sums, reference_sums_missed, rounds); });
}
int ttcr_fsm_prefill_swaps(const ttcr_fsm_grid* g, long long* swaps) {
    return guarded_on(g, [&] { if (swaps) *swaps = g->impl->prefill_swap_count(); });
}
int ttcr_fsm_reference_change(ttcr_fsm_grid* g, const void* times, const void* field, int parallel, void* out) {
    return guarded_on(g, [&] {
        if (!times || !field || !out) throw ValueError("ttcr_fsm_reference_change: null argument");
        g->impl->reference_change_host(times, field, parallel != 0, out);
    });
}

int ttcr_fsm_last_kernel(const ttcr_fsm_grid* g, char* buf, size_t n) {
    return guarded_on(g, [&] {
        if (!buf || n == 0) throw ValueError("ttcr_fsm_last_kernel: no buffer");
        const std::string k = g->impl->kernel_name();
        std::snprintf(buf, n, "%s", k.c_str());
    });
}

#ifndef TTCR_BUILD_ID
#define TTCR_BUILD_ID "unknown"
#endif
const char* ttcr_fsm_build_id(void) { return TTCR_BUILD_ID; }

}  // extern "C"
