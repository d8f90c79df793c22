// integration/Grid3Drnfs_amd.h -- reference-side binding of the MI355X FSM backend (3-D).
//
// This file belongs in the reference tree (ttcr/Grid3Drnfs_amd.h).  It makes libttcr_amd.so (C ABI:
// include/ttcr_amd.h) look like one more leaf of ttcr::Grid3D<T1,T2>, so ttcrpy's Cython classes, which hold a
// Grid3D<T,uint32_t>* (src/ttcrpy/rgrid.pyx:153, :1862), keep every `self.grid.*` call they make
// (src/ttcrpy/rgrid.pxd:30-106).  It takes the seat of Grid3Drnfs / Grid3Drcfs (ttcr/Grid3Drnfs.h:39-50,
// ttcr/Grid3Drcfs.h:41-52) and of their OpenCL twins (ttcr/Grid3Drnfs_OpenCL.h), selected in __cinit__
// (rgrid.pyx:208-282).  Nothing of the reference is modified; this repository only compiles it against the
// reference headers where they lie (integration/Makefile) to prove that it builds and runs.
//
// Overridden virtuals of Grid3D (ttcr/Grid3D.h): setSlowness x2, getSlowness, getNumberOfNodes/Cells, getTT,
// the geometry getters, get_niter/get_niterw, computeSlowness, checkPts, and the PUBLIC single-source raytrace
// overloads (tt / tt + r_data / list-of-receiver-lists forms); the l_data and m_data overloads throw like
// rgrid.pyx:916-917 says for FSM ("L not implemented") resp. with the reason given in DESIGN.md (compute_M).
// Grid3D's multi-source overloads (ttcr/Grid3D.h:810-853) are NOT virtual: through a Grid3D* they spawn nt host
// threads that call the single-source virtual with threadNo -- that works here (calls on one handle are
// serialised by the library) but solves one source at a time.  raytrace_batch() below hands all sources to the
// device in ONE call; rgrid.pyx reaches it through a dynamic_cast.
#ifndef TTCR_GRID3DRNFS_AMD_H
#define TTCR_GRID3DRNFS_AMD_H

#include <atomic>
#include <mutex>
#include <iostream>   // (Grid3D.h / Grid2D.h use std::cout without including it)
#include <sstream>
#include <stdexcept>
#include <vector>

#include "Grid3D.h"
#include "ttcr_amd.h"

namespace ttcr {

template <typename T1, typename T2>
class Grid3Drnfs_amd : public Grid3D<T1, T2> {
    static_assert(sizeof(sxyz<T1>) == 3 * sizeof(T1), "sxyz<T1> must be three packed T1 (passed as const void*)");
    static_assert(sizeof(T1) == 4 || sizeof(T1) == 8, "float or double");

   public:
    // cellSlowness: false <-> Grid3Drnfs, true <-> Grid3Drcfs; the other arguments are those constructors'.
    Grid3Drnfs_amd(const bool cellSlowness, const T2 nx, const T2 ny, const T2 nz, const T1 ddx, const T1 minx,
                   const T1 miny, const T1 minz, const T1 eps, const int maxit, const bool w, const bool ttrp = true,
                   const bool intVel = false, const size_t nt = 1, const bool translateOrigin = false, const int device = -1)
        // no cell->node neighbour lists (ncells = 0) and no host thread pool: the sources run side by side on the GPU
        : Grid3D<T1, T2>(ttrp, 0, nt, translateOrigin, false),
          ncx(nx), ncy(ny), ncz(nz), dx(ddx), xmin(minx), ymin(miny), zmin(minz),
          xmax(minx + nx * ddx), ymax(miny + ny * ddx), zmax(minz + nz * ddx), cells(cellSlowness), last_slot(0) {
        chk(ttcr_fsm3d_create(&h, sizeof(T1) == 4 ? TTCR_F32 : TTCR_F64, cellSlowness ? 1 : 0, nx, ny, nz, ddx, minx, miny,
                              minz, eps, maxit, w ? 1 : 0, (int)nt, translateOrigin ? 1 : 0, device));
        chk(ttcr_fsm_set_option(h, "interp_vel", intVel ? 1.0 : 0.0));
        if (translateOrigin) this->origin = {minx, miny, minz};   // as buildGridNodes does (ttcr/Grid3Drn.h:362-372)
    }
    ~Grid3Drnfs_amd() override { ttcr_fsm_destroy(h); }
    Grid3Drnfs_amd(const Grid3Drnfs_amd&) = delete;
    Grid3Drnfs_amd& operator=(const Grid3Drnfs_amd&) = delete;

    void setSlowness(const std::vector<T1>& s) override { chk(ttcr_fsm_set_slowness(h, s.data(), s.size())); }
    void setSlowness(const T1* s, const size_t ns) override { chk(ttcr_fsm_set_slowness(h, s, ns)); }
    void getSlowness(std::vector<T1>& s) const override {
        s.resize(ttcr_fsm_n_nodes(h));
        chk(ttcr_fsm_get_slowness(h, s.data(), s.size()));
    }
    size_t getNumberOfNodes() const override { return ttcr_fsm_n_nodes(h); }
    size_t getNumberOfCells() const override { return ttcr_fsm_n_cells(h); }
    void getTT(std::vector<T1>& tt, const size_t threadNo = 0) const override {
        tt.resize(ttcr_fsm_n_nodes(h));
        chk(ttcr_fsm_get_tt(h, (int)threadNo, tt.data(), tt.size()));
    }
    const T1 getXmin() const override { return xmin; }
    const T1 getXmax() const override { return xmax; }
    const T1 getYmin() const override { return ymin; }
    const T1 getYmax() const override { return ymax; }
    const T1 getZmin() const override { return zmin; }
    const T1 getZmax() const override { return zmax; }
    const T1 getDx() const override { return dx; }
    const T1 getDy() const override { return dx; }
    const T1 getDz() const override { return dx; }
    const T2 getNcx() const override { return ncx; }
    const T2 getNcy() const override { return ncy; }
    const T2 getNcz() const override { return ncz; }
    const T2 getNsnx() const override { return 0; }
    const T2 getNsny() const override { return 0; }
    const T2 getNsnz() const override { return 0; }
    // the reference keeps ONE (racy) pair per grid; here: the slot of the last single-source call
    const int get_niter() const override { int a = 0, b = 0; chk(ttcr_fsm_get_niter(h, last_slot.load(), &a, &b)); return a; }
    const int get_niterw() const override { int a = 0, b = 0; chk(ttcr_fsm_get_niter(h, last_slot.load(), &a, &b)); return b; }

    T1 computeSlowness(sxyz<T1> pt, const bool isTranslated = false) const override {
        T1 out = 0;
        chk(ttcr_fsm_compute_slowness(h, 1, &pt, isTranslated ? 1 : 0, &out));
        return out;
    }
    void checkPts(std::vector<sxyz<T1>> pts, const bool translated = false) const override {
        for (size_t n = 0; n < pts.size(); ++n) {
            sxyz<T1> p = pts[n];
            if (this->translateOrigin && !translated) p -= this->origin;
            const T1 x0 = this->translateOrigin ? T1(0) : xmin, y0 = this->translateOrigin ? T1(0) : ymin,
                     z0 = this->translateOrigin ? T1(0) : zmin;
            const T1 x1 = x0 + (xmax - xmin), y1 = y0 + (ymax - ymin), z1 = z0 + (zmax - zmin);
            if (p.x < x0 || p.x > x1 || p.y < y0 || p.y > y1 || p.z < z0 || p.z > z1) {
                std::ostringstream msg;
                msg << "Error: Point (" << p << ") outside grid.";
                throw std::runtime_error(msg.str());
            }
        }
    }

    // ---- single source (ttcr/Grid3D.h:470-502): origin translation, solve and receiver traveltimes all behind the ABI
    void raytrace(const std::vector<sxyz<T1>>& Tx, const std::vector<T1>& t0, const std::vector<sxyz<T1>>& Rx,
                  std::vector<T1>& traveltimes, const size_t threadNo = 0) const override {
        if (t0.size() != Tx.size()) throw std::runtime_error("Error: Tx and t0 of different sizes.");
        traveltimes.resize(Rx.size());
        chk(ttcr_fsm_set_option(h, "tt_from_rp", this->tt_from_rp ? 1.0 : 0.0));   // setTraveltimeFromRaypath() may have changed it
        chk(ttcr_fsm_raytrace(h, (int)threadNo, (int)Tx.size(), Tx.data(), t0.data(), (int)Rx.size(), Rx.data(), traveltimes.data()));
        last_slot.store((int)threadNo);
    }
    // one source, several receiver lists (ttcr/Grid3D.h:505-543): one solve, the lists strung together
    void raytrace(const std::vector<sxyz<T1>>& Tx, const std::vector<T1>& t0, const std::vector<std::vector<sxyz<T1>>>& Rx,
                  std::vector<std::vector<T1>*>& traveltimes, const size_t threadNo = 0) const override {
        std::vector<sxyz<T1>> all;
        for (const auto& r : Rx) all.insert(all.end(), r.begin(), r.end());
        std::vector<T1> tt;
        raytrace(Tx, t0, all, tt, threadNo);
        size_t k = 0;
        for (size_t n = 0; n < Rx.size(); ++n) {
            traveltimes[n]->assign(tt.begin() + k, tt.begin() + k + Rx[n].size());
            k += Rx[n].size();
        }
    }
    // with raypaths (ttcr/Grid3D.h:546-586)
    void raytrace(const std::vector<sxyz<T1>>& Tx, const std::vector<T1>& t0, const std::vector<sxyz<T1>>& Rx,
                  std::vector<T1>& traveltimes, std::vector<std::vector<sxyz<T1>>>& r_data, const size_t threadNo = 0) const override {
        // one call solves and keeps the rays of THIS slot: Grid3D's multi-source r_data overload (ttcr/Grid3D.h:855-905)
        // calls this from nt host threads at once, and every thread must find its own rays
        if (t0.size() != Tx.size()) throw std::runtime_error("Error: Tx and t0 of different sizes.");
        traveltimes.resize(Rx.size());
        chk(ttcr_fsm_set_option(h, "tt_from_rp", this->tt_from_rp ? 1.0 : 0.0));
        chk(ttcr_fsm_raytrace_rays(h, (int)threadNo, (int)Tx.size(), Tx.data(), t0.data(), (int)Rx.size(), Rx.data(), traveltimes.data()));
        last_slot.store((int)threadNo);
        fetch_slot_rays(r_data, threadNo);
    }
    void raytrace(const std::vector<sxyz<T1>>& Tx, const std::vector<T1>& t0, const std::vector<std::vector<sxyz<T1>>>& Rx,
                  std::vector<std::vector<T1>*>& traveltimes, std::vector<std::vector<std::vector<sxyz<T1>>>*>& r_data,
                  const size_t threadNo = 0) const override {
        std::vector<sxyz<T1>> all;
        for (const auto& r : Rx) all.insert(all.end(), r.begin(), r.end());
        std::vector<T1> tt;
        std::vector<std::vector<sxyz<T1>>> rays;
        raytrace(Tx, t0, all, tt, rays, threadNo);
        size_t k = 0;
        for (size_t n = 0; n < Rx.size(); ++n) {
            traveltimes[n]->assign(tt.begin() + k, tt.begin() + k + Rx[n].size());
            r_data[n]->assign(rays.begin() + k, rays.begin() + k + Rx[n].size());
            k += Rx[n].size();
        }
    }
    // L: "compute_L not implemented for FSM" in ttcrpy itself (rgrid.pyx:916-917): refused, never silently wrong
    void raytrace(const std::vector<sxyz<T1>>&, const std::vector<T1>&, const std::vector<sxyz<T1>>&, std::vector<T1>&,
                  std::vector<std::vector<siv<T1>>>&, const size_t = 0) const override { no_LM("l_data"); }
    void raytrace(const std::vector<sxyz<T1>>&, const std::vector<T1>&, const std::vector<sxyz<T1>>&, std::vector<T1>&,
                  std::vector<std::vector<sxyz<T1>>>&, std::vector<std::vector<siv<T1>>>&, const size_t = 0) const override { no_LM("l_data"); }
    // M (ttcr/Grid3D.h:743-772 and, with the rays, :646-680): one call behind the ABI, entries per receiver in the
    // reference's push order (its degenerate interior segments included -- include/ttcr_amd.h, ttcr_fsm_raytrace_m)
    void raytrace(const std::vector<sxyz<T1>>& Tx, const std::vector<T1>& t0, const std::vector<sxyz<T1>>& Rx, std::vector<T1>& traveltimes,
                  std::vector<std::vector<sijv<T1>>>& m_data, const size_t threadNo = 0) const override {
        solve_m(Tx, t0, Rx, traveltimes, m_data, threadNo);
    }
    void raytrace(const std::vector<sxyz<T1>>& Tx, const std::vector<T1>& t0, const std::vector<sxyz<T1>>& Rx, std::vector<T1>& traveltimes,
                  std::vector<std::vector<sxyz<T1>>>& r_data, std::vector<std::vector<sijv<T1>>>& m_data, const size_t threadNo = 0) const override {
        solve_m(Tx, t0, Rx, traveltimes, m_data, threadNo, true);   // (this overload's own terms: ttcr_fsm_raytrace_rm)
        fetch_slot_rays(r_data, threadNo);
    }

    // ---- all sources in one device call: what Grid3D's multi-source overload (ttcr/Grid3D.h:810-853) does with host
    // threads.  Same arguments, same results; r_data (optional) as in the overload of :855-905.
    void raytrace_batch(const std::vector<std::vector<sxyz<T1>>>& Tx, const std::vector<std::vector<T1>>& t0,
                        const std::vector<std::vector<sxyz<T1>>>& Rx, std::vector<std::vector<T1>>& traveltimes,
                        std::vector<std::vector<std::vector<sxyz<T1>>>>* r_data = nullptr) const {
        const size_t ns = Tx.size();
        if (t0.size() != ns || Rx.size() != ns) throw std::runtime_error("Error: Tx, t0 and Rx of different sizes.");
        std::vector<int> tx_off(ns + 1, 0), rx_off(ns + 1, 0);
        std::vector<sxyz<T1>> tx, rx;
        std::vector<T1> vt0;
        for (size_t n = 0; n < ns; ++n) {
            if (t0[n].size() != Tx[n].size()) throw std::runtime_error("Error: Tx and t0 of different sizes.");
            tx.insert(tx.end(), Tx[n].begin(), Tx[n].end());
            vt0.insert(vt0.end(), t0[n].begin(), t0[n].end());
            rx.insert(rx.end(), Rx[n].begin(), Rx[n].end());
            tx_off[n + 1] = (int)tx.size();
            rx_off[n + 1] = (int)rx.size();
        }
        std::vector<T1> tt(rx.size());
        chk(ttcr_fsm_set_option(h, "tt_from_rp", this->tt_from_rp ? 1.0 : 0.0));
        {
            std::lock_guard<std::mutex> rays_lock(rays_mu);   // option, solve and fetch are one unit
            RaysOn on(h, r_data != nullptr);
            chk(ttcr_fsm_raytrace_multi(h, (int)ns, tx_off.data(), tx.data(), vt0.data(), rx_off.data(), rx.data(), tt.data()));
            if (r_data) {
                std::vector<std::vector<sxyz<T1>>> rays;
                fetch_rays(rays);
                r_data->resize(ns);
                for (size_t n = 0; n < ns; ++n) (*r_data)[n].assign(rays.begin() + rx_off[n], rays.begin() + rx_off[n + 1]);
            }
        }
        traveltimes.resize(ns);
        for (size_t n = 0; n < ns; ++n) traveltimes[n].assign(tt.begin() + rx_off[n], tt.begin() + rx_off[n + 1]);
    }

    // ---- the same for the overloads with m_data (ttcr/Grid3D.h:896-1000: Grid3D runs the single-source overload per source on host
    // threads): ONE call, batched solves, then the walks of each batch.  m_data[n][r] as the single-source overloads fill them;
    // r_data (optional) selects the overload that keeps the rays -- its terms are another matrix (include/ttcr_amd.h).
    void raytrace_batch_m(const std::vector<std::vector<sxyz<T1>>>& Tx, const std::vector<std::vector<T1>>& t0,
                          const std::vector<std::vector<sxyz<T1>>>& Rx, std::vector<std::vector<T1>>& traveltimes,
                          std::vector<std::vector<std::vector<sijv<T1>>>>& m_data,
                          std::vector<std::vector<std::vector<sxyz<T1>>>>* r_data = nullptr) const {
        const size_t ns = Tx.size();
        if (t0.size() != ns || Rx.size() != ns) throw std::runtime_error("Error: Tx, t0 and Rx of different sizes.");
        std::vector<int> tx_off(ns + 1, 0), rx_off(ns + 1, 0);
        std::vector<sxyz<T1>> tx, rx;
        std::vector<T1> vt0;
        for (size_t n = 0; n < ns; ++n) {
            if (t0[n].size() != Tx[n].size()) throw std::runtime_error("Error: Tx and t0 of different sizes.");
            tx.insert(tx.end(), Tx[n].begin(), Tx[n].end());
            vt0.insert(vt0.end(), t0[n].begin(), t0[n].end());
            rx.insert(rx.end(), Rx[n].begin(), Rx[n].end());
            tx_off[n + 1] = (int)tx.size();
            rx_off[n + 1] = (int)rx.size();
        }
        std::vector<T1> tt(rx.size());
        std::lock_guard<std::mutex> rays_lock(rays_mu);   // call and fetches are one unit
        chk(ttcr_fsm_raytrace_multi_m(h, (int)ns, tx_off.data(), tx.data(), vt0.data(), rx_off.data(), rx.data(), tt.data(), r_data ? 1 : 0));
        size_t nrow = 0, nnz = 0;
        chk(ttcr_fsm_multi_m_size(h, &nrow, &nnz));
        std::vector<long long> off(nrow + 1), j(nnz ? nnz : 1);
        std::vector<T1> v(nnz ? nnz : 1);
        chk(ttcr_fsm_get_multi_m(h, off.data(), j.data(), v.data()));
        traveltimes.resize(ns);
        m_data.assign(ns, std::vector<std::vector<sijv<T1>>>());
        for (size_t n = 0; n < ns; ++n) {
            traveltimes[n].assign(tt.begin() + rx_off[n], tt.begin() + rx_off[n + 1]);
            m_data[n].resize(Rx[n].size());
            for (size_t r = 0; r < Rx[n].size(); ++r)
                for (long long e = off[rx_off[n] + r]; e < off[rx_off[n] + r + 1]; ++e) m_data[n][r].push_back(sijv<T1>(r, (size_t)j[e], v[e]));
        }
        if (r_data) {
            std::vector<std::vector<sxyz<T1>>> rays;
            fetch_rays(rays);
            r_data->resize(ns);
            for (size_t n = 0; n < ns; ++n) (*r_data)[n].assign(rays.begin() + rx_off[n], rays.begin() + rx_off[n + 1]);
        }
    }

    ttcr_fsm_grid* handle() const { return h; }

   private:
    ttcr_fsm_grid* h = nullptr;
    mutable std::mutex rays_mu;
    T2 ncx, ncy, ncz;
    T1 dx, xmin, ymin, zmin, xmax, ymax, zmax;
    bool cells;
    mutable std::atomic<int> last_slot;

    // the reference's convention: every failure is a C++ exception; Cython's `except +` turns it into RuntimeError
    static void chk(int st) {
        if (st != TTCR_OK) throw std::runtime_error(ttcr_fsm_last_error());
    }
    [[noreturn]] static void no_LM(const char* what) {
        throw std::runtime_error(std::string("Error: raytrace overload with ") + what + " is not available for the FSM backend on MI355X");
    }
    void fetch_slot_rays(std::vector<std::vector<sxyz<T1>>>& r_data, const size_t threadNo) const {
        size_t nr = 0, np = 0;
        chk(ttcr_fsm_slot_rays_size(h, (int)threadNo, &nr, &np));
        std::vector<long long> off(nr + 1);
        std::vector<sxyz<T1>> pts(np ? np : 1);
        chk(ttcr_fsm_get_slot_rays(h, (int)threadNo, off.data(), pts.data()));
        r_data.resize(nr);
        for (size_t n = 0; n < nr; ++n) r_data[n].assign(pts.begin() + off[n], pts.begin() + off[n + 1]);
    }
    void solve_m(const std::vector<sxyz<T1>>& Tx, const std::vector<T1>& t0, const std::vector<sxyz<T1>>& Rx, std::vector<T1>& traveltimes,
                 std::vector<std::vector<sijv<T1>>>& m_data, const size_t threadNo, bool with_rays = false) const {
        if (t0.size() != Tx.size()) throw std::runtime_error("Error: Tx and t0 of different sizes.");
        traveltimes.resize(Rx.size());
        chk((with_rays ? ttcr_fsm_raytrace_rm : ttcr_fsm_raytrace_m)(h, (int)threadNo, (int)Tx.size(), Tx.data(), t0.data(), (int)Rx.size(),
                                                                     Rx.data(), traveltimes.data()));
        last_slot.store((int)threadNo);
        size_t nrow = 0, nnz = 0;
        chk(ttcr_fsm_slot_m_size(h, (int)threadNo, &nrow, &nnz));
        std::vector<long long> off(nrow + 1), j(nnz ? nnz : 1);
        std::vector<T1> v(nnz ? nnz : 1);
        chk(ttcr_fsm_get_slot_m(h, (int)threadNo, off.data(), j.data(), v.data()));
        m_data.assign(Rx.size(), std::vector<sijv<T1>>());
        for (size_t n = 0; n < nrow && n < Rx.size(); ++n)
            for (long long e = off[n]; e < off[n + 1]; ++e) m_data[n].push_back(sijv<T1>(n, (size_t)j[e], v[e]));
    }
    struct RaysOn {   // option "return_rays" for the duration of one call
        ttcr_fsm_grid* g;
        bool on;
        explicit RaysOn(ttcr_fsm_grid* g_, bool on_ = true) : g(g_), on(on_) { if (on) chk(ttcr_fsm_set_option(g, "return_rays", 1.0)); }
        ~RaysOn() { if (on) (void)ttcr_fsm_set_option(g, "return_rays", 0.0); }
    };
    void fetch_rays(std::vector<std::vector<sxyz<T1>>>& r_data) const {
        size_t nr = 0, np = 0;
        chk(ttcr_fsm_rays_size(h, &nr, &np));
        std::vector<long long> off(nr + 1);
        std::vector<sxyz<T1>> pts(np ? np : 1);
        chk(ttcr_fsm_get_rays(h, off.data(), pts.data()));
        r_data.resize(nr);
        for (size_t n = 0; n < nr; ++n) r_data[n].assign(pts.begin() + off[n], pts.begin() + off[n + 1]);
    }
};

}  // namespace ttcr
#endif
