# distutils: language = c++
# cython: language_level=3
# integration/abi_probe.pyx -- a module that uses the C ABI through integration/ttcr_amd.pxd, exactly the way rgrid.pyx
# would (cimport, cdef handle, status -> exception).  Cythonized and compiled by tests/test_integration.py.
from ttcr_amd cimport (ttcr_fsm_grid, ttcr_fsm3d_create, ttcr_fsm_destroy, ttcr_fsm_last_error, ttcr_fsm_device_count,
                       ttcr_fsm_n_nodes, ttcr_fsm_set_slowness, ttcr_fsm_raytrace, ttcr_fsm_get_niter, TTCR_F64, TTCR_OK)
from libc.stdint cimport uint32_t


def device_count():
    return ttcr_fsm_device_count()


def create_and_destroy(uint32_t n):
    """returns (status, message, n_nodes): on a box without a GPU the status is TTCR_ERR_DEVICE and nothing is created"""
    cdef ttcr_fsm_grid* g = NULL
    cdef int st = ttcr_fsm3d_create(&g, TTCR_F64, 0, n, n, n, 1.0, 0.0, 0.0, 0.0, 1e-5, 50, 0, 1, 0, -1)
    if st != TTCR_OK:
        return st, ttcr_fsm_last_error().decode(), 0
    cdef size_t nn = ttcr_fsm_n_nodes(g)
    ttcr_fsm_destroy(g)
    return st, "", nn
