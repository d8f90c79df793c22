/* tests/capi_smoke.c -- the C ABI of include/ttcr_amd.h driven from plain C, no Python in between:
 * create -> set_slowness -> raytrace -> get_tt / interp / niter -> error paths -> destroy, 3-D and 2-D.
 * Built with `gcc tests/capi_smoke.c -Iinclude -Lttcr_amd -lttcr_amd` by tests/test_capi_smoke_gpu.py, which
 * compares the printed traveltimes (hex floats, lossless) with the CPU oracle.  Exit code = failed checks. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ttcr_amd.h"

static int failures = 0;

#define CHECK(cond, what)                                            \
    do {                                                             \
        if (cond) {                                                  \
            printf("ok %s\n", what);                                 \
        } else {                                                     \
            printf("FAIL %s (last error: %s)\n", what, ttcr_fsm_last_error()); \
            ++failures;                                              \
        }                                                            \
    } while (0)

/* deterministic heterogeneous slowness in [0.3, 1.0] (the test regenerates it from the same formula) */
static float slow(unsigned n) {
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

int main(void) {
    if (ttcr_fsm_device_count() < 1) {
        printf("FAIL no HIP device\n");
        return 100;
    }
    /* ------------------------------------------------------------------ 3-D, node slowness, fp32 */
    const unsigned ncx = 18, ncy = 14, ncz = 11;
    const size_t nn = (size_t)(ncx + 1) * (ncy + 1) * (ncz + 1);
    ttcr_fsm_grid* g = NULL;
    int st = ttcr_fsm3d_create(&g, TTCR_F32, 0, ncx, ncy, ncz, 0.5, 1.0, -2.0, 0.0, 1e-5, 50, 0, 2, 0, -1);
    CHECK(st == TTCR_OK && g != NULL, "create3d");
    if (!g) return 101;
    CHECK(ttcr_fsm_n_nodes(g) == nn && ttcr_fsm_n_cells(g) == (size_t)ncx * ncy * ncz && ttcr_fsm_n_slots(g) == 2, "sizes");

    const float tx[3] = {3.3f, 1.1f, 2.7f}, t0[1] = {0.25f};
    const float rx[9] = {1.0f, -2.0f, 0.0f, 10.0f, 5.0f, 5.5f, 4.4f, 0.3f, 1.9f};
    float tt[3] = {0, 0, 0};
    /* raytrace before a model was assigned: the reference's grids hold zeros; here it is an error */
    st = ttcr_fsm_raytrace(g, 0, 1, tx, t0, 3, rx, tt);
    CHECK(st == TTCR_ERR_RUNTIME && strstr(ttcr_fsm_last_error(), "slowness"), "raytrace_without_slowness -> TTCR_ERR_RUNTIME");

    float* s = (float*)malloc(nn * sizeof(float));
    for (size_t n = 0; n < nn; ++n) s[n] = smooth3((unsigned)n, ncx + 1, ncy + 1);
    /* Grid3Drn::setSlowness: std::length_error("Error: slowness vectors of incompatible size.") */
    st = ttcr_fsm_set_slowness(g, s, nn - 1);
    CHECK(st == TTCR_ERR_RUNTIME && strcmp(ttcr_fsm_last_error(), "Error: slowness vectors of incompatible size.") == 0,
          "set_slowness_wrong_size -> TTCR_ERR_RUNTIME + reference message");
    CHECK(ttcr_fsm_set_slowness(g, s, nn) == TTCR_OK, "set_slowness");
    float* back = (float*)malloc(nn * sizeof(float));
    CHECK(ttcr_fsm_get_slowness(g, back, nn) == TTCR_OK && memcmp(back, s, nn * sizeof(float)) == 0, "get_slowness round trip");

    CHECK(ttcr_fsm_raytrace(g, 1, 1, tx, t0, 3, rx, tt) == TTCR_OK, "raytrace slot 1");
    int niter = -1, niterw = -1;
    CHECK(ttcr_fsm_get_niter(g, 1, &niter, &niterw) == TTCR_OK && niter >= 1 && niterw == 0, "get_niter");
    printf("niter3d %d\n", niter);
    printf("tt3d %a %a %a\n", tt[0], tt[1], tt[2]);
    float* field = (float*)malloc(nn * sizeof(float));
    CHECK(ttcr_fsm_get_tt(g, 1, field, nn) == TTCR_OK, "get_tt");
    double sum = 0;
    for (size_t n = 0; n < nn; ++n) sum += field[n];
    printf("field3d_sum %a\n", sum);
    printf("field3d_probe %a %a %a\n", field[0], field[nn / 2], field[nn - 1]);
    float ti[3];
    CHECK(ttcr_fsm_interp(g, 1, 3, rx, ti) == TTCR_OK && memcmp(ti, tt, sizeof(ti)) == 0, "interp == raytrace receivers");
    ttcr_fsm_timing tm;
    CHECK(ttcr_fsm_last_timing(g, &tm) == TTCR_OK && tm.n_sources == 1 && tm.iterations == niter, "last_timing");
    void* dptr = NULL;
    size_t stride = 0;
    CHECK(ttcr_fsm_get_tt_device_view(g, 1, &dptr, &stride) == TTCR_OK && dptr && stride == 1,
          "get_tt_device_view (2 slots of a small grid: contiguous fields, stride 1)");
    CHECK(ttcr_fsm_get_tt_device(g, 1, &dptr) == TTCR_OK && dptr, "get_tt_device");

    /* Grid3Drn::checkPts: runtime_error("Error: Point (x y z) outside grid.") -- receiver and source */
    const float rx_out[3] = {1.0f, -2.0f, 5.6f};
    st = ttcr_fsm_raytrace(g, 0, 1, tx, t0, 1, rx_out, tt);
    CHECK(st == TTCR_ERR_RUNTIME && strcmp(ttcr_fsm_last_error(), "Error: Point (1 -2 5.6) outside grid.") == 0,
          "receiver outside -> TTCR_ERR_RUNTIME + reference message");
    const float tx_out[3] = {0.5f, 0.0f, 1.0f};
    st = ttcr_fsm_raytrace(g, 0, 1, tx_out, t0, 3, rx, tt);
    CHECK(st == TTCR_ERR_RUNTIME && strcmp(ttcr_fsm_last_error(), "Error: Point (0.5 0 1) outside grid.") == 0,
          "source outside -> TTCR_ERR_RUNTIME + reference message");
    CHECK(ttcr_fsm_raytrace(g, 2, 1, tx, t0, 3, rx, tt) == TTCR_ERR_VALUE, "slot out of range -> TTCR_ERR_VALUE");
    CHECK(ttcr_fsm_get_tt(g, 0, field, nn - 3) == TTCR_ERR_VALUE, "get_tt wrong size -> TTCR_ERR_VALUE");
    CHECK(ttcr_fsm_set_option(g, "no_such_option", 1.0) == TTCR_ERR_VALUE, "unknown option -> TTCR_ERR_VALUE");

    /* multi-source overload: 3 sources on 2 slots, ragged receiver lists, raypaths returned */
    const float mtx[9] = {3.3f, 1.1f, 2.7f, 8.0f, 2.0f, 4.0f, 2.0f, 0.0f, 1.5f};
    const float mt0[3] = {0.25f, 0.0f, 1.0f};
    const int tx_off[4] = {0, 1, 2, 3}, rx_off[4] = {0, 3, 4, 6};
    const float mrx[18] = {1.0f, -2.0f, 0.0f, 10.0f, 5.0f, 5.5f, 4.4f, 0.3f, 1.9f, 2.0f, 2.0f, 2.0f, 9.5f, 4.5f, 5.0f, 3.0f, 0.0f, 1.0f};
    float mtt[6];
    CHECK(ttcr_fsm_raytrace_multi(g, 3, tx_off, mtx, mt0, rx_off, mrx, mtt) == TTCR_OK, "raytrace_multi");
    printf("ttmulti %a %a %a %a %a %a\n", mtt[0], mtt[1], mtt[2], mtt[3], mtt[4], mtt[5]);
    CHECK(mtt[0] == ti[0] && mtt[1] == ti[1] && mtt[2] == ti[2], "source 0 of the batch == the single solve");
    CHECK(ttcr_fsm_set_option(g, "return_rays", 1.0) == TTCR_OK, "return_rays on");
    float rtt[6];
    CHECK(ttcr_fsm_raytrace_multi(g, 3, tx_off, mtx, mt0, rx_off, mrx, rtt) == TTCR_OK, "raytrace_multi with rays");
    size_t n_rays = 0, n_pts = 0;
    CHECK(ttcr_fsm_rays_size(g, &n_rays, &n_pts) == TTCR_OK && n_rays == 6 && n_pts >= 12, "rays_size");
    long long* off = (long long*)malloc((n_rays + 1) * sizeof(long long));
    float* pts = (float*)malloc(3 * n_pts * sizeof(float));
    CHECK(ttcr_fsm_get_rays(g, off, pts) == TTCR_OK && off[0] == 0 && (size_t)off[n_rays] == n_pts, "get_rays");
    int ends_ok = 1;
    for (int n = 0; n < 3; ++n)
        for (int r = rx_off[n]; r < rx_off[n + 1]; ++r) {   /* every ray runs from its receiver to its source, in row order */
            const float* a = pts + 3 * off[r];
            const float* b = pts + 3 * (off[r + 1] - 1);
            if (memcmp(a, mrx + 3 * r, 3 * sizeof(float)) != 0 || memcmp(b, mtx + 3 * n, 3 * sizeof(float)) != 0) ends_ok = 0;
        }
    CHECK(ends_ok, "rays in receiver-row order, receiver -> source");
    CHECK(ttcr_fsm_set_option(g, "return_rays", 0.0) == TTCR_OK, "return_rays off");

    /* the r_data overload a worker thread calls: traveltimes along the ray + the rays of THIS slot */
    const float ry[6] = {10.0f, 5.0f, 5.5f, 4.4f, 0.3f, 1.9f};
    float rtt2[2];
    CHECK(ttcr_fsm_raytrace_rays(g, 0, 1, tx, t0, 2, ry, rtt2) == TTCR_OK, "raytrace_rays slot 0");
    printf("ttrays %a %a\n", rtt2[0], rtt2[1]);
    size_t sr = 0, sp = 0;
    CHECK(ttcr_fsm_slot_rays_size(g, 0, &sr, &sp) == TTCR_OK && sr == 2 && sp >= 4, "slot_rays_size");
    long long soff[3];
    float* spts = (float*)malloc(3 * sp * sizeof(float));
    CHECK(ttcr_fsm_get_slot_rays(g, 0, soff, spts) == TTCR_OK && soff[0] == 0 && (size_t)soff[2] == sp, "get_slot_rays");
    CHECK(memcmp(spts, ry, 3 * sizeof(float)) == 0 && memcmp(spts + 3 * (soff[1] - 1), tx, 3 * sizeof(float)) == 0,
          "slot ray 0 runs receiver -> source");
    printf("rays_npts %lld %lld\n", soff[1], soff[2] - soff[1]);
    /* the m_data overload (compute_M) on the other slot; the rays of slot 0 stay what they were */
    float mtt2[2];
    CHECK(ttcr_fsm_raytrace_m(g, 1, 1, tx, t0, 2, ry, mtt2) == TTCR_OK, "raytrace_m slot 1");
    printf("ttm %a %a\n", mtt2[0], mtt2[1]);
    size_t mrows = 0, mnnz = 0;
    CHECK(ttcr_fsm_slot_m_size(g, 1, &mrows, &mnnz) == TTCR_OK && mrows == 2 && mnnz >= 16, "slot_m_size");
    long long mro[3];
    long long* mj = (long long*)malloc(mnnz * sizeof(long long));
    float* mv = (float*)malloc(mnnz * sizeof(float));
    CHECK(ttcr_fsm_get_slot_m(g, 1, mro, mj, mv) == TTCR_OK && mro[0] == 0 && (size_t)mro[2] == mnnz, "get_slot_m");
    double msum = 0;
    long long jsum = 0;
    for (size_t n = 0; n < mnnz; ++n) { msum += mv[n]; jsum += mj[n]; }
    printf("m_shape %lld %lld\n", mro[1], mro[2] - mro[1]);
    printf("m_sums %a %lld\n", msum, jsum);
    size_t sr2 = 0, sp2 = 0;
    CHECK(ttcr_fsm_slot_rays_size(g, 0, &sr2, &sp2) == TTCR_OK && sr2 == sr && sp2 == sp, "rays of slot 0 untouched by slot 1");
    CHECK(ttcr_fsm_raytrace_m(g, 2, 1, tx, t0, 2, ry, mtt2) == TTCR_ERR_VALUE, "raytrace_m slot out of range -> TTCR_ERR_VALUE");
    /* the overload with r_data and m_data (compute_M with return_rays): more entries, the rays of slot 1 alongside */
    size_t mrows2 = 0, mnnz2 = 0, sr3 = 0, sp3 = 0;
    CHECK(ttcr_fsm_raytrace_rm(g, 1, 1, tx, t0, 2, ry, mtt2) == TTCR_OK, "raytrace_rm slot 1");
    CHECK(ttcr_fsm_slot_m_size(g, 1, &mrows2, &mnnz2) == TTCR_OK && mrows2 == 2 && mnnz2 >= mnnz, "slot_m_size after raytrace_rm");
    CHECK(ttcr_fsm_slot_rays_size(g, 1, &sr3, &sp3) == TTCR_OK && sr3 == 2 && sp3 == sp, "rays of raytrace_rm = rays of raytrace_rays");
    printf("ttrm %a %a\n", mtt2[0], mtt2[1]);
    double ch[50], chw[50];
    CHECK(ttcr_fsm_get_changes(g, 1, ch, 50, chw, 50) == TTCR_OK && ch[0] > 0 && ch[niter - 1] >= 0 && chw[0] == 0, "get_changes");
    printf("change3d %a %a\n", ch[0], ch[niter - 1]);
    CHECK(ttcr_fsm_n_devices(g) == 1, "n_devices of a plain grid");
    free(spts); free(mj); free(mv);
    ttcr_fsm_destroy(g);

    /* ------------------------------------------------------------------ one handle over several replicas */
    g = NULL;
    const int devs[2] = {0, 0};   /* the same ordinal twice: two replicas on the one GPU of the test box */
    CHECK(ttcr_fsm3d_create_multi(&g, TTCR_F32, 0, ncx, ncy, ncz, 0.5, 1.0, -2.0, 0.0, 1e-5, 50, 0, 4, 0, devs, 2) == TTCR_OK,
          "create3d_multi");
    CHECK(ttcr_fsm_n_devices(g) == 2 && ttcr_fsm_n_slots(g) == 4, "n_devices / n_slots of the multi handle");
    CHECK(ttcr_fsm_set_slowness(g, s, nn) == TTCR_OK, "set_slowness reaches every replica");
    float dmtt[6];
    CHECK(ttcr_fsm_raytrace_multi(g, 3, tx_off, mtx, mt0, rx_off, mrx, dmtt) == TTCR_OK, "raytrace_multi over two replicas");
    CHECK(memcmp(dmtt, mtt, sizeof(mtt)) == 0, "two replicas == one device, bit for bit");
    CHECK(ttcr_fsm_raytrace(g, 3, 1, tx, t0, 3, rx, tt) == TTCR_OK && memcmp(tt, ti, sizeof(ti)) == 0, "slot 3 lives on replica 1");
    ttcr_fsm_grid* g2 = NULL;
    CHECK(ttcr_fsm3d_create_multi(&g2, TTCR_F32, 0, ncx, ncy, ncz, 0.5, 1.0, -2.0, 0.0, 1e-5, 50, 0, 4, 0, devs, 0) == TTCR_ERR_VALUE && !g2,
          "empty device list -> TTCR_ERR_VALUE");
    ttcr_fsm_destroy(g);

    /* ------------------------------------------------------------------ 3-D cell slowness + bad arguments */
    g = NULL;
    CHECK(ttcr_fsm3d_create(&g, 7, 0, 4, 4, 4, 1.0, 0, 0, 0, 1e-5, 50, 0, 1, 0, -1) == TTCR_ERR_VALUE && g == NULL, "bad dtype -> TTCR_ERR_VALUE");
    CHECK(ttcr_fsm3d_create(&g, TTCR_F64, 1, 0, 4, 4, 1.0, 0, 0, 0, 1e-5, 50, 0, 1, 0, -1) == TTCR_ERR_VALUE, "zero cells -> TTCR_ERR_VALUE");
    CHECK(ttcr_fsm3d_create(&g, TTCR_F64, 1, 6, 5, 4, 1.0, 0, 0, 0, 1e-5, 50, 0, 1, 0, -1) == TTCR_OK, "create3d cells fp64");
    double sc[6 * 5 * 4];
    for (int n = 0; n < 6 * 5 * 4; ++n) sc[n] = slow(1000u + (unsigned)n);
    CHECK(ttcr_fsm_set_slowness(g, sc, 7 * 6 * 5) == TTCR_ERR_RUNTIME, "cell grid takes the CELL count");
    CHECK(ttcr_fsm_set_slowness(g, sc, 6 * 5 * 4) == TTCR_OK, "set_slowness cells");
    const double dtx[3] = {2.5, 2.5, 1.0}, dt0[1] = {0.0}, drx[6] = {0, 0, 0, 6, 5, 4};
    double dtt[2];
    CHECK(ttcr_fsm_raytrace(g, 0, 1, dtx, dt0, 2, drx, dtt) == TTCR_OK, "raytrace cells fp64");
    printf("ttcells %a %a\n", dtt[0], dtt[1]);
    ttcr_fsm_destroy(g);

    /* ------------------------------------------------------------------ 2-D, dx != dz */
    g = NULL;
    CHECK(ttcr_fsm2d_create(&g, TTCR_F32, 0, 20, 12, 0.5, 0.25, 0.0, 0.0, 1e-5, 50, 0, 0, 1, -1) == TTCR_OK, "create2d");
    const size_t nn2 = 21 * 13;
    float s2[21 * 13];
    for (size_t n = 0; n < nn2; ++n) s2[n] = slow(5000u + (unsigned)n);
    CHECK(ttcr_fsm_set_slowness(g, s2, nn2) == TTCR_OK, "set_slowness 2d");
    const float tx2[2] = {3.3f, 1.1f}, t02[1] = {0.0f}, rx2[4] = {0.0f, 0.0f, 10.0f, 3.0f};
    float tt2[2];
    CHECK(ttcr_fsm_raytrace(g, 0, 1, tx2, t02, 2, rx2, tt2) == TTCR_OK, "raytrace 2d");
    printf("tt2d %a %a\n", tt2[0], tt2[1]);
    const float rx2_out[2] = {10.5f, 1.0f};
    st = ttcr_fsm_raytrace(g, 0, 1, tx2, t02, 1, rx2_out, tt2);
    CHECK(st == TTCR_ERR_RUNTIME && strcmp(ttcr_fsm_last_error(), "Error: Point (10.5, 1) outside grid.") == 0,
          "2-D point outside -> TTCR_ERR_RUNTIME + reference message");
    ttcr_fsm_destroy(g);

    free(s); free(back); free(field); free(off); free(pts);
    printf("failures %d\n", failures);
    return failures;
}
