# distutils: language = c++
# cython: language_level=3
# integration/adapter_probe.pyx -- what Grid3d_d.__cinit__ / raytrace do with the adapter (src/ttcrpy/rgrid.pyx:208-282,
# :1040-1090): `new Grid3Drnfs_amd[double,uint32_t](...)`, setSlowness from a numpy array, raytrace, exceptions through
# `except +`.  Cythonized + compiled by tests/test_integration.py against the reference headers (build container).
from libc.stdint cimport uint32_t
from libcpp.vector cimport vector
from ttcr_amd_adapters cimport Grid3Drnfs_amd, sxyz


def solve(uint32_t n, double dx, slowness, src, rcv):
    """one source, receivers rcv -> traveltimes; raises RuntimeError with the library's message when there is no GPU"""
    cdef Grid3Drnfs_amd[double, uint32_t]* g = new Grid3Drnfs_amd[double, uint32_t](False, n, n, n, dx, 0.0, 0.0, 0.0, 1e-5, 50,
                                                                                   False, False, False, 1, False)
    cdef vector[double] s, t0, tt
    cdef vector[sxyz[double]] Tx, Rx
    try:
        for v in slowness:
            s.push_back(v)
        g.setSlowness(s)
        Tx.push_back(sxyz[double](src[0], src[1], src[2]))
        t0.push_back(0.0)
        for r in rcv:
            Rx.push_back(sxyz[double](r[0], r[1], r[2]))
        g.raytrace(Tx, t0, Rx, tt, 0)
        return [tt[i] for i in range(tt.size())]
    finally:
        del g
