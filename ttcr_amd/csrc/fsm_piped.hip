// ttcr_amd/csrc/fsm_piped.hip -- translation unit of the pipelined sweep kernel (fsm_piped_kernels.h); see fsm_piped_api.h.
#include "fsm_piped_kernels.h"
#include "fsm_piped_api.h"

namespace ttcr_amd {

hipError_t fsm_piped_launch(const PersistArgs<float>& pa, unsigned wgs, size_t dyn_lds, hipStream_t stream, int device) {
    static size_t attr_set[64] = {};   // (per device: the attribute belongs to the function on that device)
    if (dyn_lds > attr_set[device & 63]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fsm_sweep_piped), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);
        if (e != hipSuccess) return e;
        attr_set[device & 63] = dyn_lds;
    }
    fsm_sweep_piped<<<dim3(wgs), dim3(FSM_PIPED_THREADS), dyn_lds, stream>>>(pa);
    return hipGetLastError();
}

}  // namespace ttcr_amd
