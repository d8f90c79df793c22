// integration/adapter_check.cpp -- compiles the reference-side adapters (Grid3Drnfs_amd.h, Grid2Drnfs_amd.h) against the
// UNMODIFIED reference headers where they lie (-I/root/reference/ttcr, nothing copied) and against include/ttcr_amd.h,
// checks at compile time that every virtual of Grid3D / Grid2D the Cython layer calls (src/ttcrpy/rgrid.pxd:30-106,
// :158-283) is overridden BY the adapter, and -- run on a GPU box -- drives the backend exactly as ttcrpy would:
// through Grid3D<T,uint32_t>* / Grid2D<T,uint32_t,sxz<T>>*.  Output: hex floats that tests/test_integration.py compares
// with the CPU oracle.  Test infrastructure (built by integration/Makefile into integration/_build/, git-ignored).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "Grid2Drnfs_amd.h"
#include "Grid3Drnfs_amd.h"

namespace ttcr {
int verbose = 0;
int gpu_profile = 0;
}  // namespace ttcr

using namespace ttcr;

// ---- compile-time: which class DECLARES the member a call through the adapter resolves to?
template <typename T> struct Owner3 {
    using A = Grid3Drnfs_amd<T, uint32_t>;
    using P = std::vector<sxyz<T>>;
    using V = std::vector<T>;
    using R = std::vector<std::vector<sxyz<T>>>;
    template <class D> static D set_slowness(void (D::*)(const V&));
    template <class D> static D get_slowness(void (D::*)(V&) const);
    template <class D> static D compute_slowness(T (D::*)(sxyz<T>, const bool) const);
    template <class D> static D get_tt(void (D::*)(V&, const size_t) const);
    template <class D> static D rt(void (D::*)(const P&, const V&, const P&, V&, const size_t) const);
    template <class D> static D rt_r(void (D::*)(const P&, const V&, const P&, V&, R&, const size_t) const);
    template <class D> static D rt_l(void (D::*)(const P&, const V&, const P&, V&, std::vector<std::vector<siv<T>>>&, const size_t) const);
    template <class D> static D rt_rl(void (D::*)(const P&, const V&, const P&, V&, R&, std::vector<std::vector<siv<T>>>&, const size_t) const);
    template <class D> static D rt_m(void (D::*)(const P&, const V&, const P&, V&, std::vector<std::vector<sijv<T>>>&, const size_t) const);
    template <class D> static D rt_rm(void (D::*)(const P&, const V&, const P&, V&, R&, std::vector<std::vector<sijv<T>>>&, const size_t) const);
    static_assert(std::is_same<decltype(set_slowness(&A::setSlowness)), A>::value, "setSlowness(vector&)");
    static_assert(std::is_same<decltype(get_slowness(&A::getSlowness)), A>::value, "getSlowness");
    static_assert(std::is_same<decltype(compute_slowness(&A::computeSlowness)), A>::value, "computeSlowness");
    static_assert(std::is_same<decltype(get_tt(&A::getTT)), A>::value, "getTT");
    static_assert(std::is_same<decltype(rt(&A::raytrace)), A>::value, "raytrace(tt)");
    static_assert(std::is_same<decltype(rt_r(&A::raytrace)), A>::value, "raytrace(tt, r_data)");
    static_assert(std::is_same<decltype(rt_l(&A::raytrace)), A>::value, "raytrace(tt, l_data)");
    static_assert(std::is_same<decltype(rt_rl(&A::raytrace)), A>::value, "raytrace(tt, r_data, l_data)");
    static_assert(std::is_same<decltype(rt_m(&A::raytrace)), A>::value, "raytrace(tt, m_data)");
    static_assert(std::is_same<decltype(rt_rm(&A::raytrace)), A>::value, "raytrace(tt, r_data, m_data)");
    static_assert(!std::is_abstract<A>::value && std::is_base_of<Grid3D<T, uint32_t>, A>::value, "a Grid3D leaf");
};
template struct Owner3<float>;
template struct Owner3<double>;

template <typename T> struct Owner2 {
    using S = sxz<T>;
    using A = Grid2Drnfs_amd<T, uint32_t, S>;
    using P = std::vector<S>;
    using V = std::vector<T>;
    using R = std::vector<std::vector<S>>;
    template <class D> static D set_slowness(void (D::*)(const V&));
    template <class D> static D get_slowness(void (D::*)(V&) const);
    template <class D> static D compute_slowness(T (D::*)(const S&) const);
    template <class D> static D get_tt(void (D::*)(V&, const size_t) const);
    template <class D> static D rt(void (D::*)(const P&, const V&, const P&, V&, const size_t) const);
    template <class D> static D rt_r(void (D::*)(const P&, const V&, const P&, V&, R&, const size_t) const);
    static_assert(std::is_same<decltype(set_slowness(&A::setSlowness)), A>::value, "setSlowness(vector&)");
    static_assert(std::is_same<decltype(get_slowness(&A::getSlowness)), A>::value, "getSlowness");
    static_assert(std::is_same<decltype(compute_slowness(&A::computeSlowness)), A>::value, "computeSlowness");
    static_assert(std::is_same<decltype(get_tt(&A::getTT)), A>::value, "getTT");
    static_assert(std::is_same<decltype(rt(&A::raytrace)), A>::value, "raytrace(tt)");
    static_assert(std::is_same<decltype(rt_r(&A::raytrace)), A>::value, "raytrace(tt, r_data)");
    static_assert(!std::is_abstract<A>::value && std::is_base_of<Grid2D<T, uint32_t, S>, A>::value, "a Grid2D leaf");
};
template struct Owner2<float>;
template struct Owner2<double>;

// ---- run time ------------------------------------------------------------------------------------------------
static int failures = 0;
#define CHECK(cond, what) do { if (cond) std::printf("ok %s\n", what); else { std::printf("FAIL %s\n", what); ++failures; } } while (0)

static float slow(unsigned n) {   // the formula of tests/capi_smoke.c
    unsigned h = n * 2654435761u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    return 0.3f + 0.7f * (float)(h & 0xffffu) / 65535.0f;
}

/* smooth 3-D model for the grid whose raypaths are traced (on a rough medium the reference's steepest-descent walk
 * leaves the grid and throws): every product and sum rounded to float, left to right */
static float smooth3(unsigned n, unsigned nnx, unsigned nny) {
    const unsigned i = n % nnx, j = (n / nnx) % nny, k = n / (nnx * nny);
    float v = 0.4f + 0.02f * (float)k;
    v = v + 0.01f * (float)j;
    v = v + 0.005f * (float)i;
    return v;
}

template <typename F> static std::string what_of(F&& f) {
    try { f(); } catch (const std::exception& e) { return e.what(); }
    return "";
}

int main() {
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    if (ttcr_fsm_device_count() < 1) { std::printf("FAIL no HIP device\n"); return 100; }
    {   // ------------------------------------------------ 3-D node grid, fp32, 3 slots, through Grid3D<float,uint32_t>*
        const uint32_t ncx = 18, ncy = 14, ncz = 11;
        const size_t nn = (size_t)(ncx + 1) * (ncy + 1) * (ncz + 1);
        std::unique_ptr<Grid3D<float, uint32_t>> g(new Grid3Drnfs_amd<float, uint32_t>(false, ncx, ncy, ncz, 0.5f, 1.0f, -2.0f, 0.0f,
                                                                                         1e-5f, 50, false, false, false, 3, false));
        std::vector<float> s(nn);
        for (size_t n = 0; n < nn; ++n) s[n] = smooth3((unsigned)n, ncx + 1, ncy + 1);
        std::vector<float> bad(nn - 1, 1.0f);
        CHECK(what_of([&] { g->setSlowness(bad); }) == "Error: slowness vectors of incompatible size.", "setSlowness wrong size throws the reference's message");
        g->setSlowness(s);
        std::vector<float> back;
        g->getSlowness(back);
        CHECK(back == s, "getSlowness");
        CHECK(g->getNthreads() == 3 && g->getNumberOfNodes() == nn, "getNthreads / getNumberOfNodes");
        std::vector<sxyz<float>> Tx = {{3.3f, 1.1f, 2.7f}}, Rx = {{1.0f, -2.0f, 0.0f}, {10.0f, 5.0f, 5.5f}, {4.4f, 0.3f, 1.9f}};
        std::vector<float> t0 = {0.25f}, tt;
        g->raytrace(Tx, t0, Rx, tt, 2);
        std::printf("a3_tt %a %a %a\n", tt[0], tt[1], tt[2]);
        std::printf("a3_niter %d\n", g->get_niter());
        std::vector<float> field;
        g->getTT(field, 2);
        double sum = 0;
        for (float v : field) sum += v;
        std::printf("a3_field_sum %a\n", sum);
        std::printf("a3_s0 %a\n", g->computeSlowness(Tx[0]));
        // raypaths
        std::vector<std::vector<sxyz<float>>> r_data;
        std::vector<float> tt_r;
        g->raytrace(Tx, t0, Rx, tt_r, r_data, 1);
        bool ends = r_data.size() == 3;
        for (size_t n = 0; ends && n < 3; ++n)
            ends = r_data[n].size() >= 2 && r_data[n].front().x == Rx[n].x && r_data[n].front().z == Rx[n].z && r_data[n].back().x == Tx[0].x;
        CHECK(ends, "raytrace(tt, r_data): one ray per receiver, receiver -> source");
        std::printf("a3_tt_rays %a %a %a\n", tt_r[0], tt_r[1], tt_r[2]);
        // setTraveltimeFromRaypath() of the base class is honoured by the next call
        g->setTraveltimeFromRaypath(true);
        std::vector<float> tt_rp;
        g->raytrace(Tx, t0, Rx, tt_rp, 0);
        CHECK(tt_rp == tt_r, "tt_from_rp traveltimes == traveltimes of the raypath overload");
        g->setTraveltimeFromRaypath(false);
        // Grid3D's own (non-virtual) multi-source overload: nt host threads -> the adapter's single-source virtual
        std::vector<std::vector<sxyz<float>>> mTx = {{{3.3f, 1.1f, 2.7f}}, {{8.0f, 2.0f, 4.0f}}, {{1.0f, -2.0f, 0.0f}}, {{5.5f, 3.3f, 1.1f}}};
        std::vector<std::vector<float>> mt0 = {{0.25f}, {0.0f}, {1.0f}, {0.0f}}, mtt, btt;
        std::vector<std::vector<sxyz<float>>> mRx = {Rx, {{2.0f, 2.0f, 2.0f}}, {{9.5f, 4.5f, 5.0f}, {3.0f, 0.0f, 1.0f}}, Rx};
        mtt.resize(mTx.size());   // (the reference indexes traveltimes[n] unchecked, like rgrid.pyx pre-sizes vtt)
        g->raytrace(mTx, mt0, mRx, mtt);
        dynamic_cast<Grid3Drnfs_amd<float, uint32_t>&>(*g).raytrace_batch(mTx, mt0, mRx, btt);
        CHECK(mtt == btt, "Grid3D multi-source overload (host threads) == raytrace_batch (one device call)");
        std::printf("a3_multi");
        for (const auto& v : btt) for (float x : v) std::printf(" %a", x);
        std::printf("\n");
        CHECK(mtt[0] == tt, "source 0 of the batch == the single solve");
        // the same with raypaths: Grid3D's multi-source r_data overload (ttcr/Grid3D.h:855-905) calls the adapter's
        // single-source r_data virtual from its host threads at once -- every thread has to get the rays of its own call
        {
            // (interior receivers: a walk that starts on the corner of the grid can leave it, which the reference and
            //  this backend both answer with an exception -- thrown inside a host thread of Grid3D's pool)
            const std::vector<sxyz<float>> Ri = {{2.0f, 2.0f, 2.0f}, {9.0f, 4.0f, 4.5f}, {4.4f, 0.3f, 1.9f}};
            const std::vector<std::vector<sxyz<float>>> mRx = {Ri, {{2.0f, 2.0f, 2.0f}}, {{9.0f, 4.0f, 5.0f}, {3.0f, 0.0f, 1.0f}}, Ri};
            const std::vector<std::vector<sxyz<float>>> mTx = {{{3.3f, 1.1f, 2.7f}}, {{8.0f, 2.0f, 4.0f}}, {{2.2f, -1.0f, 0.9f}}, {{5.5f, 3.3f, 1.1f}}};
            std::vector<std::vector<float>> rtt(mTx.size()), qtt;
            std::vector<std::vector<std::vector<sxyz<float>>>> rrays(mTx.size()), qrays;
            bool same = true;
            for (int rep = 0; rep < 5 && same; ++rep) {
                g->raytrace(mTx, mt0, mRx, rtt, rrays);
                dynamic_cast<Grid3Drnfs_amd<float, uint32_t>&>(*g).raytrace_batch(mTx, mt0, mRx, qtt, &qrays);
                same = rtt == qtt && rrays.size() == qrays.size();
                for (size_t n = 0; same && n < rrays.size(); ++n) {
                    same = rrays[n].size() == qrays[n].size();
                    for (size_t r = 0; same && r < rrays[n].size(); ++r) {
                        same = rrays[n][r].size() == qrays[n][r].size();
                        for (size_t q = 0; same && q < rrays[n][r].size(); ++q)
                            same = rrays[n][r][q].x == qrays[n][r][q].x && rrays[n][r][q].y == qrays[n][r][q].y && rrays[n][r][q].z == qrays[n][r][q].z;
                    }
                }
            }
            CHECK(same, "Grid3D multi-source r_data overload (host threads): every thread gets its own rays == raytrace_batch");
        }
        // error paths, as exceptions with the reference's texts
        std::vector<sxyz<float>> out = {{1.0f, -2.0f, 5.6f}};
        CHECK(what_of([&] { g->raytrace(Tx, t0, out, tt, 0); }) == "Error: Point (1 -2 5.6) outside grid.", "receiver outside grid throws the reference's message");
        CHECK(what_of([&] { g->checkPts(out); }) == "Error: Point (1 -2 5.6) outside grid.", "checkPts");
        std::vector<std::vector<siv<float>>> l_data;
        CHECK(!what_of([&] { g->raytrace(Tx, t0, Rx, tt, l_data, 0); }).empty(), "l_data overload refuses");
        // M through the base class: one entry list per receiver; every value is finite, only the last hops carry weight
        {
            const std::vector<sxyz<float>> Rm = {{2.0f, 2.0f, 2.0f}, {9.0f, 4.0f, 4.5f}, {3.3f, 1.1f, 2.7f}};
            std::vector<std::vector<sijv<float>>> m_data;
            std::vector<float> ttm;
            g->raytrace(Tx, t0, Rm, ttm, m_data, 0);
            bool okm = m_data.size() == 3 && m_data[2].empty() && ttm[2] == 0.0f && !m_data[0].empty() && !m_data[1].empty();
            size_t nz = 0;
            for (const auto& row : m_data) for (const auto& e : row) { okm = okm && e.v == e.v; nz += e.v != 0.0f; }
            CHECK(okm && nz >= 8 && nz <= 48, "m_data overload: entries per receiver, none for a receiver on the source");
            std::printf("m3_entries %zu %zu nonzero %zu\n", m_data[0].size(), m_data[1].size(), nz);
            // the overload that keeps the rays: its own terms (every segment carries its length), the rays of the r_data overload
            std::vector<std::vector<sijv<float>>> m2;
            std::vector<std::vector<sxyz<float>>> r2, r1;
            std::vector<float> tt2, tt1;
            g->raytrace(Tx, t0, Rm, tt2, r2, m2, 0);
            g->raytrace(Tx, t0, Rm, tt1, r1, 0);
            size_t nz2 = 0;
            for (const auto& row : m2) for (const auto& e : row) nz2 += e.v != 0.0f;
            bool okrm = m2.size() == 3 && r2.size() == 3 && m2[2].empty() && r2[2].size() == 1 && tt2[2] == 0.0f && nz2 > nz;
            for (size_t n = 0; n < 3 && okrm; ++n) {
                okrm = okrm && r2[n].size() == r1[n].size() && (n == 2 || tt2[n] == tt1[n]);
                for (size_t k = 0; k < r2[n].size() && okrm; ++k) okrm = r2[n][k] == r1[n][k];
            }
            CHECK(okrm, "r_data + m_data overload: rays of the r_data overload, more weighted entries than the m_data-only overload");
            std::printf("rm3_entries %zu %zu nonzero %zu\n", m2[0].size(), m2[1].size(), nz2);
            // several sources with m_data in ONE call (raytrace_batch_m) against the single-source overload, entry for entry
            {
                auto* ga = dynamic_cast<Grid3Drnfs_amd<float, uint32_t>*>(g.get());
                const std::vector<std::vector<sxyz<float>>> bTx = {Tx, {{8.0f, 2.0f, 4.0f}}}, bRx = {Rm, {{2.0f, 2.0f, 2.0f}, {9.0f, 4.0f, 4.5f}}};
                const std::vector<std::vector<float>> bt0 = {t0, {0.5f}};
                std::vector<std::vector<float>> btt;
                std::vector<std::vector<std::vector<sijv<float>>>> bm;
                ga->raytrace_batch_m(bTx, bt0, bRx, btt, bm);
                bool okb = bm.size() == 2 && btt.size() == 2;
                for (size_t n = 0; n < 2 && okb; ++n) {
                    std::vector<std::vector<sijv<float>>> m1;
                    std::vector<float> tt1;
                    g->raytrace(bTx[n], bt0[n], bRx[n], tt1, m1, 0);
                    okb = okb && tt1 == btt[n] && m1.size() == bm[n].size();
                    for (size_t r = 0; r < m1.size() && okb; ++r) {
                        okb = m1[r].size() == bm[n][r].size();
                        for (size_t e = 0; e < m1[r].size() && okb; ++e)
                            okb = m1[r][e].i == bm[n][r][e].i && m1[r][e].j == bm[n][r][e].j && m1[r][e].v == bm[n][r][e].v;
                    }
                }
                CHECK(okb, "raytrace_batch_m: two sources in one call == the m_data overload source by source");
            }
        }
    }
    {   // ------------------------------------------------ 3-D cell grid, fp64, translated origin (Grid3Drcfs seat)
        std::unique_ptr<Grid3D<double, uint32_t>> g(new Grid3Drnfs_amd<double, uint32_t>(true, 6, 5, 4, 1.0, 500000.0, 4000000.0, -1000.0,
                                                                                           1e-5, 50, true, true, false, 1, true));
        std::vector<double> sc(6 * 5 * 4);
        for (size_t n = 0; n < sc.size(); ++n) sc[n] = slow(1000u + (unsigned)n);
        g->setSlowness(sc);
        std::vector<sxyz<double>> Tx = {{500002.5, 4000002.5, -999.0}}, Rx = {{500001.0, 4000001.5, -998.5}, {500005.0, 4000004.0, -997.0}};
        std::vector<double> t0 = {0.0}, tt;
        g->raytrace(Tx, t0, Rx, tt, 0);   // the ttcrpy default: cells, WENO, traveltimes from raypaths
        std::printf("a3c_tt %a %a\n", tt[0], tt[1]);
        std::printf("a3c_niter %d %d\n", g->get_niter(), g->get_niterw());
        std::printf("a3c_s0 %a\n", g->computeSlowness(Tx[0]));
    }
    {   // ------------------------------------------------ 2-D, dx != dz, through Grid2D<float,uint32_t,sxz<float>>*
        using S = sxz<float>;
        std::unique_ptr<Grid2D<float, uint32_t, S>> g(new Grid2Drnfs_amd<float, uint32_t, S>(false, 20, 12, 0.5f, 0.25f, 0.0f, 0.0f, 1e-5f, 50,
                                                                                              false, false, false, 2));
        std::vector<float> s(21 * 13);
        for (size_t n = 0; n < s.size(); ++n) s[n] = slow(5000u + (unsigned)n);
        g->setSlowness(s);
        std::vector<S> Tx = {{3.3f, 1.1f}}, Rx = {{0.0f, 0.0f}, {10.0f, 3.0f}};
        std::vector<float> t0 = {0.0f}, tt;
        g->raytrace(Tx, t0, Rx, tt, 1);
        std::printf("a2_tt %a %a\n", tt[0], tt[1]);
        std::printf("a2_s0 %a\n", g->computeSlowness(Tx[0]));
        std::vector<std::vector<S>> mTx = {{{3.3f, 1.1f}}, {{7.0f, 2.0f}}, {{0.0f, 0.0f}}}, mRx = {Rx, {{5.0f, 1.0f}}, Rx};
        std::vector<std::vector<float>> mt0 = {{0.0f}, {0.5f}, {0.0f}}, mtt, btt;
        mtt.resize(mTx.size());
        g->raytrace(mTx, mt0, mRx, mtt);
        dynamic_cast<Grid2Drnfs_amd<float, uint32_t, S>&>(*g).raytrace_batch(mTx, mt0, mRx, btt);
        CHECK(mtt == btt && mtt[0] == tt, "2-D: Grid2D multi-source overload == raytrace_batch");
        std::printf("a2_multi");
        for (const auto& v : btt) for (float x : v) std::printf(" %a", x);
        std::printf("\n");
        std::vector<S> out = {{10.5f, 1.0f}};
        CHECK(what_of([&] { g->raytrace(Tx, t0, out, tt, 0); }) == "Error: Point (10.5, 1) outside grid.", "2-D point outside throws the reference's message");
    }
    {   // ------------------------------------------------ 2-D cell grid: the overloads with l_data (compute_L) through Grid2D*
        using S = sxz<float>;
        std::unique_ptr<Grid2D<float, uint32_t, S>> g(new Grid2Drnfs_amd<float, uint32_t, S>(true, 20, 12, 0.5f, 0.5f, 0.0f, 0.0f, 1e-5f, 50,
                                                                                              false, false, false, 2));
        std::vector<float> s(20 * 12);
        for (size_t n = 0; n < s.size(); ++n) s[n] = slow(7000u + (unsigned)n);
        g->setSlowness(s);
        std::vector<S> Tx = {{3.3f, 1.1f}}, Rx = {{0.4f, 0.3f}, {9.0f, 5.0f}, {3.3f, 1.1f}};
        std::vector<float> t0 = {0.0f}, tt, tt2;
        std::vector<std::vector<siv<float>>> l_data, l2;
        std::vector<std::vector<S>> r_data;
        g->raytrace(Tx, t0, Rx, tt, l_data, 1);
        g->raytrace(Tx, t0, Rx, tt2, r_data, l2, 1);
        bool ok = l_data.size() == 3 && l_data[2].empty() && tt[2] == 0.0f && !l_data[0].empty() && !l_data[1].empty() && r_data.size() == 3;
        for (size_t n = 0; n < 2 && ok; ++n) {
            double len = 0, straight = std::sqrt((double)(Rx[n].x - Tx[0].x) * (Rx[n].x - Tx[0].x) + (double)(Rx[n].z - Tx[0].z) * (Rx[n].z - Tx[0].z));
            for (size_t e = 0; e < l_data[n].size(); ++e) {
                ok = ok && l_data[n][e].i < 240 && (e == 0 || l_data[n][e - 1].i <= l_data[n][e].i) && l2[n][e].i == l_data[n][e].i && l2[n][e].v == l_data[n][e].v;
                len += l_data[n][e].v;
            }
            ok = ok && len >= straight * 0.999 && r_data[n].size() >= 2;
        }
        CHECK(ok, "l_data overloads: entries sorted by cell, none for a receiver on the source, lengths cover the straight distance");
        std::printf("l2_entries %zu %zu tt %a %a\n", l_data[0].size(), l_data[1].size(), tt[0], tt[1]);
    }
    std::printf("failures %d\n", failures);
    return failures;
}
