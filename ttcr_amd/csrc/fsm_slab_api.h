// ttcr_amd/csrc/fsm_slab_api.h -- what the host side (fsm_capi.hip) sees of the slab sweep kernel (fsm_slab_kernels.h).  The kernel
// lives in a translation unit of its own (fsm_slab.hip): it is compiled with the max-ILP machine scheduler, which interleaves the
// independent node updates of a level (the default scheduler leaves them one after the other: one dependent chain).
#pragma once
#include "fsm_kernels.h"

namespace ttcr_amd {

struct SlabArgs {
    float* tt;                // [n_slots][n_nodes] traveltime fields, natural order (x fastest); >= 16 guard elements either side
    const float* ssh;         // sheared slowness copies [4][ssh_stride]
    size_t ssh_stride;
    const uint32_t* frozen;   // [n_slots][mask_words]
    const int* bbox;          // [n_slots][6]
    double* change;           // [n_slots]
    const int* slots;         // [batch] slot solved by batch entry z, -1: converged
    unsigned long long* evals;  // [n_slots]
    const uint32_t* order;    // units (TJ | TK << 14 | dir << 28) in ticket order
    int* sync;                // [0..3] ticket counters, [4] abort, [8..] progress words (fsm_kernels.h, "launch epoch"): per
                              // (direction, batch entry, patch) NW slab words and one "patch finished" word
    const int* iter_ptr;      // [1]: launch sequence number
    SweepGeom g;              // NF, NJ, NK, M, SR, n_nodes (npj / npk of the 16 x 16 patches: unused here)
    int npj, npk, n_patches, batch;
    uint32_t mask_words;
    float dx;
    unsigned long long timeout_ticks;
    unsigned long long* prof;   // debug (-DFSM_SLAB_PROF=1 builds, TTCR_FSM_PROF=1): per-phase wall-clock sums, else nullptr
};

constexpr int FSM_SLAB_DONE = 0x3fffffff;
constexpr int FSM_SLAB_GUARD = 16;       // elements the host keeps allocated in front of / behind the fields
constexpr int FSM_SLAB_C = 8;            // levels per chunk = elements per piece
constexpr int FSM_SLAB_COLB = 128;       // bytes of a ring column: 32 levels
constexpr int FSM_SLAB_ROWB = 66 * 128;  // bytes of a ring row: J-upwind halo column, 64 lanes, J-downwind halo column
__host__ __device__ constexpr int fsm_slab_rows(int pkr, int nw) { return 1 + nw * (pkr + 1); }   // K-upwind halo row, then per slab its rows + the row below
__host__ __device__ constexpr size_t fsm_slab_lds_bytes(int pkr, int nw) { return (size_t)fsm_slab_rows(pkr, nw) * FSM_SLAB_ROWB + 1024; }

// fsm_slab.hip.  Shapes: (rows per wavefront) x (wavefronts per workgroup) = 2x4, 1x4, 2x2, 4x2; false: no such instantiation
bool fsm_slab_shape_ok(int pkr, int nw);
// launches fsm_sweep_slab<pkr, nw> with `wgs` workgroups on `stream` (device: the current device, for the per-device function attribute)
hipError_t fsm_slab_launch(int pkr, int nw, const SlabArgs& a, unsigned wgs, hipStream_t stream, int device);

}  // namespace ttcr_amd
