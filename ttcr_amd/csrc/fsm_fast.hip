// ttcr_amd/csrc/fsm_fast.hip -- translation unit of the sweep kernels with tolerance-grade arithmetic; see fsm_fast_api.h.
#include "fsm_fast_api.h"

namespace ttcr_amd {

template <int PJ, int PK, int C, bool IS3D, bool SKIP, int NS, bool PRE, int H = 1>
static hipError_t launch_one(const PersistArgs<float>& pa, unsigned wgs, size_t dyn_lds, hipStream_t stream) {
    fsm_sweep_persistent<float, PJ, PK, C, IS3D, SKIP, H, NS, true, PRE, 1><<<dim3(wgs), dim3(PJ * PK), dyn_lds, stream>>>(pa);
    return hipGetLastError();
}

hipError_t fsm_fast_launch(const PersistArgs<float>& pa, const FastCfg& c, unsigned wgs, size_t dyn_lds, hipStream_t stream) {
    if (c.h == 2) {   // WENO stage: one field per workgroup
#ifndef FSM_FAST_MIN
        if (c.ns != 1) return hipErrorInvalidValue;
        if (c.dim == 3 && c.chunk == 16) {
            if (c.skip) return c.pre ? launch_one<16, 16, 16, true, true, 1, true, 2>(pa, wgs, dyn_lds, stream)
                                     : launch_one<16, 16, 16, true, true, 1, false, 2>(pa, wgs, dyn_lds, stream);
            return c.pre ? launch_one<16, 16, 16, true, false, 1, true, 2>(pa, wgs, dyn_lds, stream)
                         : launch_one<16, 16, 16, true, false, 1, false, 2>(pa, wgs, dyn_lds, stream);
        }
        if (c.dim == 3 && c.chunk == 8) {
            if (c.skip) return c.pre ? launch_one<16, 16, 8, true, true, 1, true, 2>(pa, wgs, dyn_lds, stream)
                                     : launch_one<16, 16, 8, true, true, 1, false, 2>(pa, wgs, dyn_lds, stream);
            return c.pre ? launch_one<16, 16, 8, true, false, 1, true, 2>(pa, wgs, dyn_lds, stream)
                         : launch_one<16, 16, 8, true, false, 1, false, 2>(pa, wgs, dyn_lds, stream);
        }
        if (c.dim == 2 && c.chunk == 16 && c.pre)
            return c.skip ? launch_one<64, 1, 16, false, true, 1, true, 2>(pa, wgs, dyn_lds, stream)
                          : launch_one<64, 1, 16, false, false, 1, true, 2>(pa, wgs, dyn_lds, stream);
#endif
        return hipErrorInvalidValue;
    }
    if (c.dim == 3 && c.ns == 1 && c.chunk == 16 && !c.pre)
        return c.skip ? launch_one<16, 16, 16, true, true, 1, false>(pa, wgs, dyn_lds, stream)
                      : launch_one<16, 16, 16, true, false, 1, false>(pa, wgs, dyn_lds, stream);
#ifndef FSM_FAST_MIN   // (tuning builds: the lone-source kernels only)
    if (c.dim == 3 && c.ns == 2 && c.chunk == 8) {
        if (c.skip) return c.pre ? launch_one<16, 16, 8, true, true, 2, true>(pa, wgs, dyn_lds, stream)
                                 : launch_one<16, 16, 8, true, true, 2, false>(pa, wgs, dyn_lds, stream);
        return c.pre ? launch_one<16, 16, 8, true, false, 2, true>(pa, wgs, dyn_lds, stream)
                     : launch_one<16, 16, 8, true, false, 2, false>(pa, wgs, dyn_lds, stream);
    }
    if (c.dim == 2 && c.ns == 1 && c.chunk == 16 && c.pre)
        return c.skip ? launch_one<64, 1, 16, false, true, 1, true>(pa, wgs, dyn_lds, stream)
                      : launch_one<64, 1, 16, false, false, 1, true>(pa, wgs, dyn_lds, stream);
#endif
    return hipErrorInvalidValue;
}

}  // namespace ttcr_amd
