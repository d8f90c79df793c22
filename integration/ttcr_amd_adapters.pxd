# integration/ttcr_amd_adapters.pxd -- the adapter classes as rgrid.pxd would declare them.  In rgrid.pxd they sit next to
# Grid3Drnfs / Grid2Drnfs and derive from the Grid3D / Grid2D declarations there (rgrid.pxd:30-106, :158-283), so a
# `Grid3D[double,uint32_t]*` member can hold them (rgrid.pyx:153); here the members ttcrpy uses are repeated so that the
# file stands on its own.  Needs the reference headers on the include path (build container).
from libc.stdint cimport uint32_t
from libcpp cimport bool
from libcpp.vector cimport vector

cdef extern from "ttcr_t.h" namespace "ttcr" nogil:
    cdef cppclass sxyz[T]:
        sxyz()
        sxyz(T, T, T)
        T x
        T y
        T z
    cdef cppclass sxz[T]:
        sxz()
        sxz(T, T)
        T x
        T z

cdef extern from "Grid3Drnfs_amd.h" namespace "ttcr" nogil:
    cdef cppclass Grid3Drnfs_amd[T1, T2]:
        Grid3Drnfs_amd(bool, T2, T2, T2, T1, T1, T1, T1, T1, int, bool, bool, bool, size_t, bool) except +
        size_t getNthreads()
        size_t getNumberOfNodes()
        void setTraveltimeFromRaypath(bool)
        void setSlowness(vector[T1]&) except +
        void getSlowness(vector[T1]&) except +
        T1 computeSlowness(sxyz[T1]) except +
        void getTT(vector[T1]& tt, size_t threadNo) except +
        void raytrace(vector[sxyz[T1]]& Tx, vector[T1]& t0, vector[sxyz[T1]]& Rx, vector[T1]& tt, size_t thread_no) except +
        void raytrace(vector[sxyz[T1]]& Tx, vector[T1]& t0, vector[sxyz[T1]]& Rx, vector[T1]& tt,
                      vector[vector[sxyz[T1]]]& r_data, size_t thread_no) except +
        void raytrace_batch(vector[vector[sxyz[T1]]]& Tx, vector[vector[T1]]& t0, vector[vector[sxyz[T1]]]& Rx,
                            vector[vector[T1]]& traveltimes) except +

cdef extern from "Grid2Drnfs_amd.h" namespace "ttcr" nogil:
    cdef cppclass Grid2Drnfs_amd[T1, T2, S]:
        Grid2Drnfs_amd(bool, T2, T2, T1, T1, T1, T1, T1, int, bool, bool, bool, size_t) except +
        size_t getNthreads()
        void setSlowness(vector[T1]&) except +
        void getTT(vector[T1]& tt, size_t threadNo) except +
        void raytrace(vector[S]& Tx, vector[T1]& t0, vector[S]& Rx, vector[T1]& tt, size_t thread_no) except +
        void raytrace_batch(vector[vector[S]]& Tx, vector[vector[T1]]& t0, vector[vector[S]]& Rx,
                            vector[vector[T1]]& traveltimes) except +
