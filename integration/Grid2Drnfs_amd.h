// integration/Grid2Drnfs_amd.h -- reference-side binding of the MI355X FSM backend (2-D).
//
// Belongs in the reference tree as ttcr/Grid2Drnfs_amd.h: libttcr_amd.so (include/ttcr_amd.h) as one more leaf of
// ttcr::Grid2D<T1,T2,S>, in the seat of Grid2Drnfs / Grid2Drcfs (ttcr/Grid2Drnfs.h:84-95, ttcr/Grid2Drcfs.h:39-58) and
// their OpenCL twins, selected in Grid2d.__cinit__ (src/ttcrpy/rgrid.pyx:2929-2966).  See Grid3Drnfs_amd.h for the
// conventions (errors as exceptions, non-virtual multi-source overloads, raytrace_batch).
#ifndef TTCR_GRID2DRNFS_AMD_H
#define TTCR_GRID2DRNFS_AMD_H

#include <atomic>
#include <mutex>
#include <iostream>   // (Grid3D.h / Grid2D.h use std::cout without including it)
#include <stdexcept>
#include <vector>

#include "Grid2D.h"
#include "ttcr_amd.h"

namespace ttcr {

template <typename T1, typename T2, typename S>
class Grid2Drnfs_amd : public Grid2D<T1, T2, S> {
    static_assert(sizeof(S) == 2 * sizeof(T1), "S must be two packed T1 (sxz<T1>; passed as const void*)");

   public:
    Grid2Drnfs_amd(const bool cellSlowness, const T2 nx, const T2 nz, const T1 ddx, const T1 ddz, const T1 minx, const T1 minz,
                   const T1 eps, const int maxit, const bool w, const bool rt, const bool ttrp, const size_t nt = 1,
                   const int device = -1)
        : Grid2D<T1, T2, S>(0, ttrp, nt, false), ncx(nx), ncz(nz), dx(ddx), dz(ddz), xmin(minx), zmin(minz),
          xmax(minx + nx * ddx), zmax(minz + nz * ddz), last_slot(0) {
        chk(ttcr_fsm2d_create(&h, sizeof(T1) == 4 ? TTCR_F32 : TTCR_F64, cellSlowness ? 1 : 0, nx, nz, ddx, ddz, minx, minz, eps,
                              maxit, w ? 1 : 0, rt ? 1 : 0, (int)nt, device));
    }
    ~Grid2Drnfs_amd() override { ttcr_fsm_destroy(h); }
    Grid2Drnfs_amd(const Grid2Drnfs_amd&) = delete;
    Grid2Drnfs_amd& operator=(const Grid2Drnfs_amd&) = delete;

    void setSlowness(const std::vector<T1>& s) override { chk(ttcr_fsm_set_slowness(h, s.data(), s.size())); }
    void setSlowness(const T1* s, const size_t ns) override { chk(ttcr_fsm_set_slowness(h, s, ns)); }
    void getSlowness(std::vector<T1>& s) const override {
        s.resize(ttcr_fsm_n_nodes(h));
        chk(ttcr_fsm_get_slowness(h, s.data(), s.size()));
    }
    size_t getNumberOfNodes(const bool = false) const override { return ttcr_fsm_n_nodes(h); }
    size_t getNumberOfCells() const override { return ttcr_fsm_n_cells(h); }
    void getTT(std::vector<T1>& tt, const size_t threadNo = 0) const override {
        tt.resize(ttcr_fsm_n_nodes(h));
        chk(ttcr_fsm_get_tt(h, (int)threadNo, tt.data(), tt.size()));
    }
    const T1 getXmin() const override { return xmin; }
    const T1 getXmax() const override { return xmax; }
    const T1 getZmin() const override { return zmin; }
    const T1 getZmax() const override { return zmax; }
    const T1 getDx() const override { return dx; }
    const T1 getDz() const override { return dz; }
    const T2 getNcx() const override { return ncx; }
    const T2 getNcz() const override { return ncz; }
    const T2 getNsnx() const override { return 0; }
    const T2 getNsnz() const override { return 0; }
    const int get_niter() const override { int a = 0, b = 0; chk(ttcr_fsm_get_niter(h, last_slot.load(), &a, &b)); return a; }
    const int get_niterw() const override { int a = 0, b = 0; chk(ttcr_fsm_get_niter(h, last_slot.load(), &a, &b)); return b; }
    T1 computeSlowness(const S& pt) const override {
        T1 out = 0;
        chk(ttcr_fsm_compute_slowness(h, 1, &pt, 0, &out));
        return out;
    }

    void raytrace(const std::vector<S>& Tx, const std::vector<T1>& t0, const std::vector<S>& Rx, std::vector<T1>& traveltimes,
                  const size_t threadNo = 0) const override {
        if (t0.size() != Tx.size()) throw std::runtime_error("Error: Tx and t0 of different sizes.");
        traveltimes.resize(Rx.size());
        chk(ttcr_fsm_set_option(h, "tt_from_rp", this->tt_from_rp ? 1.0 : 0.0));
        chk(ttcr_fsm_raytrace(h, (int)threadNo, (int)Tx.size(), Tx.data(), t0.data(), (int)Rx.size(), Rx.data(), traveltimes.data()));
        last_slot.store((int)threadNo);
    }
    void raytrace(const std::vector<S>& Tx, const std::vector<T1>& t0, const std::vector<const std::vector<S>*>& Rx,
                  std::vector<std::vector<T1>*>& traveltimes, const size_t threadNo = 0) const override {
        std::vector<S> all;
        for (const auto* r : Rx) all.insert(all.end(), r->begin(), r->end());
        std::vector<T1> tt;
        raytrace(Tx, t0, all, tt, threadNo);
        size_t k = 0;
        for (size_t n = 0; n < Rx.size(); ++n) {
            traveltimes[n]->assign(tt.begin() + k, tt.begin() + k + Rx[n]->size());
            k += Rx[n]->size();
        }
    }
    void raytrace(const std::vector<S>& Tx, const std::vector<T1>& t0, const std::vector<S>& Rx, std::vector<T1>& traveltimes,
                  std::vector<std::vector<S>>& r_data, const size_t threadNo = 0) const override {
        // one call solves and keeps the rays of THIS slot: Grid3D's multi-source r_data overload (ttcr/Grid3D.h:855-905)
        // calls this from nt host threads at once, and every thread must find its own rays
        if (t0.size() != Tx.size()) throw std::runtime_error("Error: Tx and t0 of different sizes.");
        traveltimes.resize(Rx.size());
        chk(ttcr_fsm_set_option(h, "tt_from_rp", this->tt_from_rp ? 1.0 : 0.0));
        chk(ttcr_fsm_raytrace_rays(h, (int)threadNo, (int)Tx.size(), Tx.data(), t0.data(), (int)Rx.size(), Rx.data(), traveltimes.data()));
        last_slot.store((int)threadNo);
        size_t nr = 0, np = 0;
        chk(ttcr_fsm_slot_rays_size(h, (int)threadNo, &nr, &np));
        std::vector<long long> off(nr + 1);
        std::vector<S> pts(np ? np : 1);
        chk(ttcr_fsm_get_slot_rays(h, (int)threadNo, off.data(), pts.data()));
        r_data.resize(nr);
        for (size_t n = 0; n < nr; ++n) r_data[n].assign(pts.begin() + off[n], pts.begin() + off[n + 1]);
    }
    void raytrace(const std::vector<S>& Tx, const std::vector<T1>& t0, const std::vector<const std::vector<S>*>& Rx,
                  std::vector<std::vector<T1>*>& traveltimes, std::vector<std::vector<std::vector<S>>*>& r_data,
                  const size_t threadNo = 0) const override {
        std::vector<S> all;
        for (const auto* r : Rx) all.insert(all.end(), r->begin(), r->end());
        std::vector<T1> tt;
        std::vector<std::vector<S>> rays;
        raytrace(Tx, t0, all, tt, rays, threadNo);
        size_t k = 0;
        for (size_t n = 0; n < Rx.size(); ++n) {
            traveltimes[n]->assign(tt.begin() + k, tt.begin() + k + Rx[n]->size());
            r_data[n]->assign(rays.begin() + k, rays.begin() + k + Rx[n]->size());
            k += Rx[n]->size();
        }
    }
    // L (ray-projection matrix of a cell grid): Grid2D::raytrace(Tx, t0, Rx, tt, [r_data,] l_data, threadNo), ttcr/Grid2D.h:583-640
    void raytrace(const std::vector<S>& Tx, const std::vector<T1>& t0, const std::vector<S>& Rx, std::vector<T1>& traveltimes,
                  std::vector<std::vector<S>>& r_data, std::vector<std::vector<siv<T1>>>& l_data, const size_t threadNo = 0) const override {
        run_l(Tx, t0, Rx, traveltimes, &r_data, l_data, threadNo);
    }
    void raytrace(const std::vector<S>& Tx, const std::vector<T1>& t0, const std::vector<S>& Rx, std::vector<T1>& traveltimes,
                  std::vector<std::vector<siv<T1>>>& l_data, const size_t threadNo = 0) const override {
        run_l(Tx, t0, Rx, traveltimes, nullptr, l_data, threadNo);
    }
    void run_l(const std::vector<S>& Tx, const std::vector<T1>& t0, const std::vector<S>& Rx, std::vector<T1>& traveltimes,
               std::vector<std::vector<S>>* r_data, std::vector<std::vector<siv<T1>>>& l_data, const size_t threadNo) const {
        if (t0.size() != Tx.size()) throw std::runtime_error("Error: Tx and t0 of different sizes.");
        traveltimes.resize(Rx.size());
        std::vector<T1> tt(Rx.size() ? Rx.size() : 1);
        chk(ttcr_fsm_raytrace_l(h, (int)threadNo, (int)Tx.size(), Tx.data(), t0.data(), (int)Rx.size(), Rx.data(), tt.data(), r_data ? 1 : 0));
        last_slot.store((int)threadNo);
        std::copy(tt.begin(), tt.begin() + Rx.size(), traveltimes.begin());
        size_t nrow = 0, nnz = 0;
        chk(ttcr_fsm_slot_l_size(h, (int)threadNo, &nrow, &nnz));
        std::vector<long long> off(nrow + 1), cellno(nnz ? nnz : 1);
        std::vector<T1> v(nnz ? nnz : 1);
        chk(ttcr_fsm_get_slot_l(h, (int)threadNo, off.data(), cellno.data(), v.data()));
        l_data.assign(Rx.size(), std::vector<siv<T1>>());
        for (size_t n = 0; n < nrow; ++n)
            for (long long e = off[n]; e < off[n + 1]; ++e) l_data[n].push_back(siv<T1>((size_t)cellno[e], v[e]));
        if (r_data) {
            size_t nr = 0, np = 0;
            chk(ttcr_fsm_slot_rays_size(h, (int)threadNo, &nr, &np));
            std::vector<long long> roff(nr + 1);
            std::vector<S> pts(np ? np : 1);
            chk(ttcr_fsm_get_slot_rays(h, (int)threadNo, roff.data(), pts.data()));
            r_data->assign(Rx.size(), std::vector<S>());
            for (size_t n = 0; n < nr; ++n) (*r_data)[n].assign(pts.begin() + roff[n], pts.begin() + roff[n + 1]);
        }
    }

    void raytrace_batch(const std::vector<std::vector<S>>& Tx, const std::vector<std::vector<T1>>& t0,
                        const std::vector<std::vector<S>>& Rx, std::vector<std::vector<T1>>& traveltimes,
                        std::vector<std::vector<std::vector<S>>>* r_data = nullptr) const {
        const size_t ns = Tx.size();
        if (t0.size() != ns || Rx.size() != ns) throw std::runtime_error("Error: Tx, t0 and Rx of different sizes.");
        std::vector<int> tx_off(ns + 1, 0), rx_off(ns + 1, 0);
        std::vector<S> tx, rx;
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
                std::vector<std::vector<S>> rays;
                fetch_rays(rays);
                r_data->resize(ns);
                for (size_t n = 0; n < ns; ++n) (*r_data)[n].assign(rays.begin() + rx_off[n], rays.begin() + rx_off[n + 1]);
            }
        }
        traveltimes.resize(ns);
        for (size_t n = 0; n < ns; ++n) traveltimes[n].assign(tt.begin() + rx_off[n], tt.begin() + rx_off[n + 1]);
    }

    ttcr_fsm_grid* handle() const { return h; }

   private:
    ttcr_fsm_grid* h = nullptr;
    mutable std::mutex rays_mu;
    T2 ncx, ncz;
    T1 dx, dz, xmin, zmin, xmax, zmax;
    mutable std::atomic<int> last_slot;

    static void chk(int st) {
        if (st != TTCR_OK) throw std::runtime_error(ttcr_fsm_last_error());
    }
    [[noreturn]] static void no_LM(const char* what) {
        throw std::runtime_error(std::string("Error: raytrace overload with ") + what + " is not available for the FSM backend on MI355X");
    }
    struct RaysOn {
        ttcr_fsm_grid* g;
        bool on;
        explicit RaysOn(ttcr_fsm_grid* g_, bool on_ = true) : g(g_), on(on_) { if (on) chk(ttcr_fsm_set_option(g, "return_rays", 1.0)); }
        ~RaysOn() { if (on) (void)ttcr_fsm_set_option(g, "return_rays", 0.0); }
    };
    void fetch_rays(std::vector<std::vector<S>>& r_data) const {
        size_t nr = 0, np = 0;
        chk(ttcr_fsm_rays_size(h, &nr, &np));
        std::vector<long long> off(nr + 1);
        std::vector<S> pts(np ? np : 1);
        chk(ttcr_fsm_get_rays(h, off.data(), pts.data()));
        r_data.resize(nr);
        for (size_t n = 0; n < nr; ++n) r_data[n].assign(pts.begin() + off[n], pts.begin() + off[n + 1]);
    }
};

}  // namespace ttcr
#endif
