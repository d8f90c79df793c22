// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// Thin extern "C" driver around the UNMODIFIED reference headers, which are
// #include'd by absolute path from /root/reference at build time (nothing of
// the reference is copied into this repo).  It is compiled by oracle/Makefile
// into oracle/_ref/libttcr_ref.so (git-ignored) and is used in the build
// container to (a) generate tests/golden/*.npz and (b) validate the C
// restatement in oracle/fsm_oracle.c value-for-value.
//
// Reference entry points exercised:
//   Grid3Drnfs<T,uint32_t>  ttcr/Grid3Drnfs.h:39-50, :84-155
//   Grid3Drcfs<T,uint32_t>  ttcr/Grid3Drcfs.h:41-52, :88-171
//   Grid2Drnfs<T,uint32_t,sxz<T>>  ttcr/Grid2Drnfs.h:84-95, :198-299
//   Grid2Drcfs<T,uint32_t,sxz<T>>  ttcr/Grid2Drcfs.h:98-138
// called through the public base-class overloads Grid3D::raytrace
// (ttcr/Grid3D.h:470-502) / Grid2D::raytrace; plus the file formats either side of the path:
//   Grid3Drn::saveTT (ttcr/Grid3Drn.h:2679-2762), Grid2Drn::saveTT (ttcr/Grid2Drn.h:419-500),
//   Src<T>::init (ttcr/Src.h:62-131), Rcv<T>::init / save_tt / save_rcvfile (ttcr/Rcv.h:78-216).
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "Grid2Drcfs.h"
#include "Grid2Drnfs.h"
#include "Grid3Drcfs.h"
#include "Grid3Drnfs.h"
#include "Rcv.h"
#include "Src.h"

namespace ttcr {
int verbose = 0;
int gpu_profile = 0;
}  // namespace ttcr

static thread_local std::string g_err;
// when set, the next solve also calls saveTT(base, all, 0, format) on the reference grid
static thread_local std::string g_save_base;
static thread_local int g_save_all = 0, g_save_format = 1;
// when set, the next 3-D solve uses the raypath overload Grid3D::raytrace(Tx,t0,Rx,tt,r_data,threadNo)
// (ttcr/Grid3D.h:546-586): ray n occupies points [off[n], off[n+1]) of buf (xyz triples, as double)
static thread_local double* g_ray_buf = nullptr;
static thread_local long g_ray_cap = 0;
static thread_local long* g_ray_off = nullptr;
extern "C" void ref_set_rays(double* buf, long cap_pts, long* off) {
    g_ray_buf = buf;
    g_ray_cap = cap_pts;
    g_ray_off = off;
}
// when set, the next 3-D solve uses the overload with m_data, Grid3D::raytrace(Tx,t0,Rx,tt,m_data,threadNo) (ttcr/Grid3D.h:743-772
// -> Grid3Drn::getRaypath(Tx,t0,Rx,m_data,RxNo,tt,threadNo), ttcr/Grid3Drn.h:1503-1800): the (j, v) entries of receiver n, in
// the order the reference pushed them, occupy [off[n], off[n+1])
static thread_local long long* g_m_j = nullptr;
static thread_local double* g_m_v = nullptr;
static thread_local long g_m_cap = 0;
static thread_local long* g_m_off = nullptr;
extern "C" void ref_set_m(long long* j, double* v, long cap, long* off) {
    g_m_j = j;
    g_m_v = v;
    g_m_cap = cap;
    g_m_off = off;
}
// when set, the next 2-D solve uses the overload with l_data, Grid2D::raytrace(Tx,t0,Rx,tt,[r_data,]l_data,threadNo)
// (ttcr/Grid2D.h:583-640; with r_data when ref_set_rays is on as well): the (cell, length) entries of receiver n, sorted as the
// reference sorts them, occupy [off[n], off[n+1])
static thread_local long long* g_l_i = nullptr;
static thread_local double* g_l_v = nullptr;
static thread_local long g_l_cap = 0;
static thread_local long* g_l_off = nullptr;
extern "C" void ref_set_l(long long* cell, double* v, long cap, long* off) {
    g_l_i = cell;
    g_l_v = v;
    g_l_cap = cap;
    g_l_off = off;
}
extern "C" void ref_set_save(const char* base, int all, int format) {
    g_save_base = base ? base : "";
    g_save_all = all;
    g_save_format = format;
}

template <typename T, typename GRID>
static int run3d(GRID& g, const T* slowness, size_t n_slowness, int n_src, const T* src_xyz,
                 const T* t0, int n_rcv, const T* rcv_xyz, T* tt_rcv, T* tt_grid, int* niter) {
    using namespace ttcr;
    try {
        std::vector<T> s(slowness, slowness + n_slowness);
        g.setSlowness(s);
        std::vector<sxyz<T>> Tx(n_src), Rx(n_rcv);
        std::vector<T> vt0(t0, t0 + n_src), tt;
        for (int n = 0; n < n_src; ++n) Tx[n] = {src_xyz[3 * n], src_xyz[3 * n + 1], src_xyz[3 * n + 2]};
        for (int n = 0; n < n_rcv; ++n) Rx[n] = {rcv_xyz[3 * n], rcv_xyz[3 * n + 1], rcv_xyz[3 * n + 2]};
        if (g_ray_buf && g_m_j) {
            // both: Grid3D::raytrace(Tx,t0,Rx,tt,r_data,m_data,threadNo) (ttcr/Grid3D.h:646-680 -> Grid3Drn::getRaypath(Tx,t0,Rx,
            // r_data,m_data,RxNo,tt,threadNo), ttcr/Grid3Drn.h:2144-2470) -- what ttcrpy calls for compute_M with return_rays
            std::vector<std::vector<sxyz<T>>> r_data;
            std::vector<std::vector<sijv<T>>> m_data;
            static_cast<Grid3D<T, uint32_t>&>(g).raytrace(Tx, vt0, Rx, tt, r_data, m_data, 0);
            long k = 0;
            for (int n = 0; n < n_rcv; ++n) {
                g_ray_off[n] = k;
                for (const auto& p : r_data[n]) {
                    if (k < g_ray_cap) { g_ray_buf[3 * k] = p.x; g_ray_buf[3 * k + 1] = p.y; g_ray_buf[3 * k + 2] = p.z; }
                    ++k;
                }
            }
            g_ray_off[n_rcv] = k;
            k = 0;
            for (int n = 0; n < n_rcv; ++n) {
                g_m_off[n] = k;
                for (const auto& e : m_data[n]) {
                    if (k < g_m_cap) { g_m_j[k] = (long long)e.j; g_m_v[k] = (double)e.v; }
                    ++k;
                }
            }
            g_m_off[n_rcv] = k;
        } else if (g_ray_buf) {
            std::vector<std::vector<sxyz<T>>> r_data;
            static_cast<Grid3D<T, uint32_t>&>(g).raytrace(Tx, vt0, Rx, tt, r_data, 0);
            long k = 0;
            for (int n = 0; n < n_rcv; ++n) {
                g_ray_off[n] = k;
                for (const auto& p : r_data[n]) {
                    if (k < g_ray_cap) { g_ray_buf[3 * k] = p.x; g_ray_buf[3 * k + 1] = p.y; g_ray_buf[3 * k + 2] = p.z; }
                    ++k;
                }
            }
            g_ray_off[n_rcv] = k;
        } else if (g_m_j) {
            std::vector<std::vector<sijv<T>>> m_data;
            static_cast<Grid3D<T, uint32_t>&>(g).raytrace(Tx, vt0, Rx, tt, m_data, 0);
            long k = 0;
            for (int n = 0; n < n_rcv; ++n) {
                g_m_off[n] = k;
                for (const auto& e : m_data[n]) {
                    if (k < g_m_cap) { g_m_j[k] = (long long)e.j; g_m_v[k] = (double)e.v; }
                    ++k;
                }
            }
            g_m_off[n_rcv] = k;
        } else {
            static_cast<Grid3D<T, uint32_t>&>(g).raytrace(Tx, vt0, Rx, tt, 0);
        }
        for (int n = 0; n < n_rcv; ++n) tt_rcv[n] = tt[n];
        std::vector<T> grid_tt;
        g.getTT(grid_tt, 0);
        std::memcpy(tt_grid, grid_tt.data(), grid_tt.size() * sizeof(T));
        niter[0] = g.get_niter();
        niter[1] = g.get_niterw();
        if (!g_save_base.empty()) g.saveTT(g_save_base, g_save_all, 0, g_save_format);
        return 0;
    } catch (std::exception& e) {
        g_err = e.what();
        return 1;
    }
}

template <typename T, typename GRID>
static int run2d(GRID& g, const T* slowness, size_t n_slowness, int n_src, const T* src_xz,
                 const T* t0, int n_rcv, const T* rcv_xz, T* tt_rcv, T* tt_grid, int* niter) {
    using namespace ttcr;
    try {
        std::vector<T> s(slowness, slowness + n_slowness);
        g.setSlowness(s);
        std::vector<sxz<T>> Tx(n_src), Rx(n_rcv);
        std::vector<T> vt0(t0, t0 + n_src), tt;
        for (int n = 0; n < n_src; ++n) Tx[n] = {src_xz[2 * n], src_xz[2 * n + 1]};
        for (int n = 0; n < n_rcv; ++n) Rx[n] = {rcv_xz[2 * n], rcv_xz[2 * n + 1]};
        if (g_l_i) {   // the overloads with l_data
            std::vector<std::vector<sxz<T>>> r_data;
            std::vector<std::vector<siv<T>>> l_data;
            if (g_ray_buf) static_cast<Grid2D<T, uint32_t, sxz<T>>&>(g).raytrace(Tx, vt0, Rx, tt, r_data, l_data, 0);
            else static_cast<Grid2D<T, uint32_t, sxz<T>>&>(g).raytrace(Tx, vt0, Rx, tt, l_data, 0);
            long k = 0;
            for (int n = 0; n < n_rcv; ++n) {
                g_l_off[n] = k;
                for (const auto& e : l_data[n]) {
                    if (k < g_l_cap) { g_l_i[k] = (long long)e.i; g_l_v[k] = e.v; }
                    ++k;
                }
            }
            g_l_off[n_rcv] = k;
            if (g_ray_buf) {
                k = 0;
                for (int n = 0; n < n_rcv; ++n) {
                    g_ray_off[n] = k;
                    for (const auto& p : r_data[n]) {
                        if (k < g_ray_cap) { g_ray_buf[2 * k] = p.x; g_ray_buf[2 * k + 1] = p.z; }
                        ++k;
                    }
                }
                g_ray_off[n_rcv] = k;
            }
        } else if (g_ray_buf) {   // Grid2D::raytrace(Tx,t0,Rx,tt,r_data,threadNo): xz pairs
            std::vector<std::vector<sxz<T>>> r_data;
            static_cast<Grid2D<T, uint32_t, sxz<T>>&>(g).raytrace(Tx, vt0, Rx, tt, r_data, 0);
            long k = 0;
            for (int n = 0; n < n_rcv; ++n) {
                g_ray_off[n] = k;
                for (const auto& p : r_data[n]) {
                    if (k < g_ray_cap) { g_ray_buf[2 * k] = p.x; g_ray_buf[2 * k + 1] = p.z; }
                    ++k;
                }
            }
            g_ray_off[n_rcv] = k;
        } else {
            static_cast<Grid2D<T, uint32_t, sxz<T>>&>(g).raytrace(Tx, vt0, Rx, tt, 0);
        }
        for (int n = 0; n < n_rcv; ++n) tt_rcv[n] = tt[n];
        std::vector<T> grid_tt;
        g.getTT(grid_tt, 0);
        std::memcpy(tt_grid, grid_tt.data(), grid_tt.size() * sizeof(T));
        niter[0] = g.get_niter();
        niter[1] = g.get_niterw();
        if (!g_save_base.empty()) g.saveTT(g_save_base, g_save_all, 0, g_save_format);
        return 0;
    } catch (std::exception& e) {
        g_err = e.what();
        return 1;
    }
}

// ncx/ncy/ncz are CELL counts, as in the reference constructors.
#define REF3D(NAME, T)                                                                              \
    extern "C" int NAME(int cell_slowness, uint32_t ncx, uint32_t ncy, uint32_t ncz, T dx, T xmin,  \
                        T ymin, T zmin, T eps, int maxit, int weno, int translate, int ttrp,        \
                        int intvel, const T* slowness, int n_src, const T* src_xyz, const T* t0,    \
                        int n_rcv, const T* rcv_xyz, T* tt_rcv, T* tt_grid, int* niter) {           \
        try {                                                                                       \
            if (cell_slowness) {                                                                    \
                ttcr::Grid3Drcfs<T, uint32_t> g(ncx, ncy, ncz, dx, xmin, ymin, zmin, eps, maxit,    \
                                                weno != 0, ttrp != 0, intvel != 0, 1,               \
                                                translate != 0);                                    \
                return run3d<T>(g, slowness, (size_t)ncx * ncy * ncz, n_src, src_xyz, t0, n_rcv,    \
                                rcv_xyz, tt_rcv, tt_grid, niter);                                   \
            }                                                                                       \
            ttcr::Grid3Drnfs<T, uint32_t> g(ncx, ncy, ncz, dx, xmin, ymin, zmin, eps, maxit,        \
                                            weno != 0, ttrp != 0, intvel != 0, 1, translate != 0);  \
            return run3d<T>(g, slowness, (size_t)(ncx + 1) * (ncy + 1) * (ncz + 1), n_src, src_xyz, \
                            t0, n_rcv, rcv_xyz, tt_rcv, tt_grid, niter);                            \
        } catch (std::exception & e) {                                                              \
            g_err = e.what();                                                                       \
            return 1;                                                                               \
        }                                                                                           \
    }

REF3D(ref_fsm3d_f32, float)
REF3D(ref_fsm3d_f64, double)

#define REF2D(NAME, T)                                                                             \
    extern "C" int NAME(int cell_slowness, uint32_t ncx, uint32_t ncz, T dx, T dz, T xmin, T zmin, \
                        T eps, int maxit, int weno, int rotated, const T* slowness, int n_src,     \
                        const T* src_xz, const T* t0, int n_rcv, const T* rcv_xz, T* tt_rcv,       \
                        T* tt_grid, int* niter, int ttrp) {                                        \
        try {                                                                                      \
            if (cell_slowness) {                                                                   \
                ttcr::Grid2Drcfs<T, uint32_t, ttcr::sxz<T>> g(ncx, ncz, dx, dz, xmin, zmin, eps,   \
                                                              maxit, weno != 0, rotated != 0,      \
                                                              ttrp != 0, 1);                       \
                return run2d<T>(g, slowness, (size_t)ncx * ncz, n_src, src_xz, t0, n_rcv, rcv_xz,  \
                                tt_rcv, tt_grid, niter);                                           \
            }                                                                                      \
            ttcr::Grid2Drnfs<T, uint32_t, ttcr::sxz<T>> g(ncx, ncz, dx, dz, xmin, zmin, eps,       \
                                                          maxit, weno != 0, rotated != 0,          \
                                                          ttrp != 0, 1);                           \
            return run2d<T>(g, slowness, (size_t)(ncx + 1) * (ncz + 1), n_src, src_xz, t0, n_rcv,  \
                            rcv_xz, tt_rcv, tt_grid, niter);                                       \
        } catch (std::exception & e) {                                                             \
            g_err = e.what();                                                                      \
            return 1;                                                                              \
        }                                                                                          \
    }

REF2D(ref_fsm2d_f32, float)
REF2D(ref_fsm2d_f64, double)

// Grid3D::computeSlowness(pt) / Grid2D::computeSlowness(pt) as the Cython layer calls them (get_s0,
// src/ttcrpy/rgrid.pyx:824, :3799), through the base-class pointer
#define REFCS3D(NAME, T)                                                                            \
    extern "C" int NAME(int cell_slowness, uint32_t ncx, uint32_t ncy, uint32_t ncz, T dx, T xmin,  \
                        T ymin, T zmin, int translate, int intvel, const T* slowness, int n,        \
                        const T* pts, T* out) {                                                     \
        try {                                                                                       \
            std::unique_ptr<ttcr::Grid3D<T, uint32_t>> g;                                           \
            size_t ns;                                                                              \
            if (cell_slowness) {                                                                    \
                g.reset(new ttcr::Grid3Drcfs<T, uint32_t>(ncx, ncy, ncz, dx, xmin, ymin, zmin, 1e-5, 5, false, \
                                                          false, intvel != 0, 1, translate != 0));  \
                ns = (size_t)ncx * ncy * ncz;                                                       \
            } else {                                                                                \
                g.reset(new ttcr::Grid3Drnfs<T, uint32_t>(ncx, ncy, ncz, dx, xmin, ymin, zmin, 1e-5, 5, false, \
                                                          false, intvel != 0, 1, translate != 0));  \
                ns = (size_t)(ncx + 1) * (ncy + 1) * (ncz + 1);                                     \
            }                                                                                       \
            std::vector<T> s(slowness, slowness + ns);                                              \
            g->setSlowness(s);                                                                      \
            for (int i = 0; i < n; ++i)                                                             \
                out[i] = g->computeSlowness(ttcr::sxyz<T>(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2])); \
            return 0;                                                                               \
        } catch (std::exception & e) {                                                              \
            g_err = e.what();                                                                       \
            return 1;                                                                               \
        }                                                                                           \
    }
REFCS3D(ref_compute_slowness3d_f32, float)
REFCS3D(ref_compute_slowness3d_f64, double)

#define REFCS2D(NAME, T)                                                                            \
    extern "C" int NAME(int cell_slowness, uint32_t ncx, uint32_t ncz, T dx, T dz, T xmin, T zmin,  \
                        const T* slowness, int n, const T* pts, T* out) {                           \
        try {                                                                                       \
            std::unique_ptr<ttcr::Grid2D<T, uint32_t, ttcr::sxz<T>>> g;                             \
            size_t ns;                                                                              \
            if (cell_slowness) {                                                                    \
                g.reset(new ttcr::Grid2Drcfs<T, uint32_t, ttcr::sxz<T>>(ncx, ncz, dx, dz, xmin, zmin, 1e-5, 5, \
                                                                        false, false, false, 1));   \
                ns = (size_t)ncx * ncz;                                                             \
            } else {                                                                                \
                g.reset(new ttcr::Grid2Drnfs<T, uint32_t, ttcr::sxz<T>>(ncx, ncz, dx, dz, xmin, zmin, 1e-5, 5, \
                                                                        false, false, false, 1));   \
                ns = (size_t)(ncx + 1) * (ncz + 1);                                                 \
            }                                                                                       \
            std::vector<T> s(slowness, slowness + ns);                                              \
            g->setSlowness(s);                                                                      \
            for (int i = 0; i < n; ++i) out[i] = g->computeSlowness(ttcr::sxz<T>(pts[2 * i], pts[2 * i + 1])); \
            return 0;                                                                               \
        } catch (std::exception & e) {                                                              \
            g_err = e.what();                                                                       \
            return 1;                                                                               \
        }                                                                                           \
    }
REFCS2D(ref_compute_slowness2d_f32, float)
REFCS2D(ref_compute_slowness2d_f64, double)

// Src / Rcv text files (double instantiation): coordinates (+ t0) into caller buffers, count returned
extern "C" int ref_read_src(const char* fname, double* xyz, double* t0, int max_n) {
    ttcr::Src<double> s(fname);
    s.init();
    const int n = (int)s.get_coord().size();
    for (int i = 0; i < n && i < max_n; ++i) {
        xyz[3 * i] = s.get_coord()[i].x; xyz[3 * i + 1] = s.get_coord()[i].y; xyz[3 * i + 2] = s.get_coord()[i].z;
        t0[i] = s.get_t0()[i];
    }
    return n;
}
extern "C" int ref_read_rcv(const char* fname, double* xyz, int max_n) {
    ttcr::Rcv<double> r(fname);
    r.init(1);
    const int n = (int)r.get_coord().size();
    for (int i = 0; i < n && i < max_n; ++i) {
        xyz[3 * i] = r.get_coord()[i].x; xyz[3 * i + 1] = r.get_coord()[i].y; xyz[3 * i + 2] = r.get_coord()[i].z;
    }
    return n;
}
// Rcv::save_rcvfile to `rcvfile` and Rcv::save_tt (one arrival per receiver) to `ttfile`
extern "C" void ref_write_rcv(const char* rcvfile, const char* ttfile, int n, const double* xyz, const double* tt) {
    ttcr::Rcv<double> r(rcvfile);
    for (int i = 0; i < n; ++i) r.add_coord({xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
    r.init_tt(1);
    r.get_tt(0).assign(tt, tt + n);
    r.save_rcvfile();
    r.save_tt(ttfile, 0);
}

extern "C" const char* ref_last_error() { return g_err.c_str(); }
