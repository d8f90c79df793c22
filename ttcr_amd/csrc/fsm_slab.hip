// ttcr_amd/csrc/fsm_slab.hip -- translation unit of the slab sweep kernel (fsm_slab_kernels.h); see fsm_slab_api.h.
// Built with -mllvm -amdgpu-sched-strategy=max-ilp (ttcr_amd/build.py).
#include "fsm_slab_kernels.h"

namespace ttcr_amd {

bool fsm_slab_shape_ok(int pkr, int nw) { return (pkr == 2 && nw == 4) || (pkr == 1 && nw == 4) || (pkr == 2 && nw == 2) || (pkr == 4 && nw == 2); }

template <int PKR, int NW>
static hipError_t launch(const SlabArgs& a, unsigned wgs, hipStream_t stream, int device) {
    static bool attr_set[64] = {};   // (per device: the attribute belongs to the function on that device)
    constexpr size_t lds = fsm_slab_lds_bytes(PKR, NW);
    if (!attr_set[device & 63]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fsm_sweep_slab<PKR, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[device & 63] = true;
    }
    fsm_sweep_slab<PKR, NW><<<dim3(wgs), dim3(64 * NW), lds, stream>>>(a);
    return hipGetLastError();
}

hipError_t fsm_slab_launch(int pkr, int nw, const SlabArgs& a, unsigned wgs, hipStream_t stream, int device) {
    if (pkr == 2 && nw == 4) return launch<2, 4>(a, wgs, stream, device);
    if (pkr == 1 && nw == 4) return launch<1, 4>(a, wgs, stream, device);
    if (pkr == 2 && nw == 2) return launch<2, 2>(a, wgs, stream, device);
    if (pkr == 4 && nw == 2) return launch<4, 2>(a, wgs, stream, device);
    return hipErrorInvalidValue;
}

}  // namespace ttcr_amd
