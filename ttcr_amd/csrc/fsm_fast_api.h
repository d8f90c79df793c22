// ttcr_amd/csrc/fsm_fast_api.h -- what the host side (fsm_capi.hip) sees of the sweep kernels with tolerance-grade arithmetic
// (option "arith" = 1: update3_fast / update2_fast / weno_axis_fast of fsm_kernels.h instead of the reference's fp64 arithmetic).  They are the
// AR = 1 instantiations of fsm_sweep_persistent -- same arguments, synchronisation words, ticket lists and launch geometry as the
// exact kernels the host would launch otherwise -- compiled in a translation unit of their own (fsm_fast.hip).
#pragma once
#include "fsm_kernels.h"

namespace ttcr_amd {

struct FastCfg {
    int h;        // 1: first-order stage, 2: WENO stage
    int dim;      // 3: patches of 16 x 16 columns; 2: one-wave patches of 64 columns
    int ns;       // fields marched per workgroup (1, or 2: source pairs, 3-D only)
    int chunk;    // levels per chunk (3-D: 8 or 16 with ns == 1, 8 with ns == 2; 2-D: 16)
    bool skip;    // exact skipping (template SKIP)
    bool pre;     // upwind counters sampled one chunk ahead (template PRE)
};
// whole-iteration launch (template XS) of the instantiation `c` names; hipErrorInvalidValue when there is none
hipError_t fsm_fast_launch(const PersistArgs<float>& pa, const FastCfg& c, unsigned wgs, size_t dyn_lds, hipStream_t stream);

}  // namespace ttcr_amd
