// ttcr_amd/csrc/fsm_piped_api.h -- what the host side (fsm_capi.hip) sees of the pipelined sweep kernel (fsm_piped_kernels.h): same
// arguments, same synchronisation words and ticket lists as fsm_sweep_persistent<float,16,16,8,true,false,1,1,true,*>; workgroups of
// 384 threads (four march wavefronts + two staging wavefronts).  The kernel lives in a translation unit of its own (fsm_piped.hip).
#pragma once
#include "fsm_kernels.h"

namespace ttcr_amd {

constexpr int FSM_PIPED_THREADS = 384;
// launches fsm_sweep_piped with `wgs` workgroups and `dyn_lds` bytes of unused dynamic LDS (occupancy cap) on `stream`
hipError_t fsm_piped_launch(const PersistArgs<float>& pa, unsigned wgs, size_t dyn_lds, hipStream_t stream, int device);

}  // namespace ttcr_amd
