# integration/ttcr_amd.pxd -- Cython declarations of the C ABI (include/ttcr_amd.h); the two adapter classes a ttcrpy
# maintainer adds to src/ttcrpy/rgrid.pxd are in ttcr_amd_adapters.pxd (they need the reference headers).  tests/test_integration.py cythonizes a module that cimports this file
# (Cython 3 is in the image), so the stub is checked by the compiler, not by eye.
from libc.stdint cimport uint32_t
from libcpp cimport bool
from libcpp.vector cimport vector

cdef extern from "ttcr_amd.h" nogil:
    ctypedef struct ttcr_fsm_grid:
        pass
    cdef enum:
        TTCR_F32
        TTCR_F64
    cdef enum:
        TTCR_OK
        TTCR_ERR_VALUE
        TTCR_ERR_RUNTIME
        TTCR_ERR_DEVICE
        TTCR_ERR_UNSUPPORTED
    ctypedef struct ttcr_fsm_timing:
        double sweep_ms
        double total_ms
        long long kernel_launches
        long long node_updates
        long long evaluated_updates
        int iterations
        int n_sources
    int ttcr_fsm_device_count()
    const char* ttcr_fsm_last_error()
    int ttcr_fsm3d_create(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncy, uint32_t ncz,
                          double dx, double xmin, double ymin, double zmin, double eps, int maxit, int weno, int n_slots,
                          int translate_origin, int device)
    int ttcr_fsm2d_create(ttcr_fsm_grid** out, int dtype, int cell_slowness, uint32_t ncx, uint32_t ncz, double dx, double dz,
                          double xmin, double zmin, double eps, int maxit, int weno, int rotated_template, int n_slots,
                          int device)
    void ttcr_fsm_destroy(ttcr_fsm_grid* g)
    int ttcr_fsm_set_slowness(ttcr_fsm_grid* g, const void* s, size_t n)
    int ttcr_fsm_set_slowness_device(ttcr_fsm_grid* g, const void* d_s, size_t n)
    int ttcr_fsm_set_slowness_c_order(ttcr_fsm_grid* g, const void* s, size_t n)
    int ttcr_fsm_get_slowness(ttcr_fsm_grid* g, void* out, size_t n)
    int ttcr_fsm_raytrace(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                          void* tt_out)
    int ttcr_fsm_raytrace_multi(ttcr_fsm_grid* g, int n_src, const int* tx_off, const void* tx, const void* t0,
                                const int* rx_off, const void* rx, void* tt_out)
    int ttcr_fsm_get_tt(ttcr_fsm_grid* g, int slot, void* out, size_t n)
    int ttcr_fsm_get_tt_device(ttcr_fsm_grid* g, int slot, void** d_ptr)
    int ttcr_fsm_get_tt_device_view(ttcr_fsm_grid* g, int slot, void** d_ptr, size_t* stride)
    int ttcr_fsm_interp(ttcr_fsm_grid* g, int slot, int n_pts, const void* pts, void* tt_out)
    int ttcr_fsm_compute_slowness(ttcr_fsm_grid* g, int n_pts, const void* pts, int translated, void* out)
    int ttcr_fsm_get_niter(ttcr_fsm_grid* g, int slot, int* niter, int* niterw)
    int ttcr_fsm_n_slots(const ttcr_fsm_grid* g)
    size_t ttcr_fsm_n_nodes(const ttcr_fsm_grid* g)
    size_t ttcr_fsm_n_cells(const ttcr_fsm_grid* g)
    int ttcr_fsm_set_option(ttcr_fsm_grid* g, const char* key, double value)
    int ttcr_fsm_rays_size(const ttcr_fsm_grid* g, size_t* n_rays, size_t* n_points)
    int ttcr_fsm_get_rays(const ttcr_fsm_grid* g, long long* offsets, void* pts)
    # per-slot rays and the matrices M / L (the r_data, m_data and l_data overloads of Grid3D / Grid2D::raytrace)
    int ttcr_fsm_raytrace_rays(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                               void* tt_out)
    int ttcr_fsm_slot_rays_size(const ttcr_fsm_grid* g, int slot, size_t* n_rays, size_t* n_points)
    int ttcr_fsm_get_slot_rays(const ttcr_fsm_grid* g, int slot, long long* offsets, void* pts)
    int ttcr_fsm_raytrace_m(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                            void* tt_out)
    int ttcr_fsm_raytrace_rm(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                             void* tt_out)
    int ttcr_fsm_raytrace_multi_m(ttcr_fsm_grid* g, int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off,
                                  const void* rx, void* tt_out, int with_rays)
    int ttcr_fsm_multi_m_size(const ttcr_fsm_grid* g, size_t* n_rows, size_t* nnz)
    int ttcr_fsm_get_multi_m(const ttcr_fsm_grid* g, long long* row_off, long long* j, void* v)
    int ttcr_fsm_slot_m_size(const ttcr_fsm_grid* g, int slot, size_t* n_rows, size_t* nnz)
    int ttcr_fsm_get_slot_m(const ttcr_fsm_grid* g, int slot, long long* row_off, long long* j, void* v)
    int ttcr_fsm_raytrace_l(ttcr_fsm_grid* g, int slot, int n_tx, const void* tx, const void* t0, int n_rx, const void* rx,
                            void* tt_out, int with_rays)
    int ttcr_fsm_raytrace_multi_l(ttcr_fsm_grid* g, int n_src, const int* tx_off, const void* tx, const void* t0, const int* rx_off,
                                  const void* rx, void* tt_out, int with_rays)
    int ttcr_fsm_multi_l_size(const ttcr_fsm_grid* g, size_t* n_rows, size_t* nnz)
    int ttcr_fsm_get_multi_l(const ttcr_fsm_grid* g, long long* row_off, long long* cell, void* v)
    int ttcr_fsm_slot_l_size(const ttcr_fsm_grid* g, int slot, size_t* n_rows, size_t* nnz)
    int ttcr_fsm_get_slot_l(const ttcr_fsm_grid* g, int slot, long long* row_off, long long* cell, void* v)
    int ttcr_fsm_last_timing(const ttcr_fsm_grid* g, ttcr_fsm_timing* out)
    const char* ttcr_fsm_build_id()
