/* oracle/fsm_oracle_impl.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, serial, structure-of-arrays) of the reference's
 * rectilinear fast-sweeping eikonal solver.  This file is a "template": it is
 * included twice by fsm_oracle.c with
 *     REAL = float,  SFX(name) = name##_f32
 *     REAL = double, SFX(name) = name##_f64
 * so that both instantiations of the reference (T1 = float / double) are
 * restated with the reference's exact arithmetic, including the promotion to
 * double that the reference's double literals (2., 0.5, 1./3., small ...) force
 * when T1 = float.
 *
 * It is NOT the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call it, and only as the checker / the timed CPU
 * baseline.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py compares this
 * restatement bit-for-bit with the compiled, unmodified reference
 * (oracle/_ref/libttcr_ref.so, built by oracle/Makefile from the sources under
 * /root/reference) in the build container, and tests/test_oracle_golden.py
 * compares it with the committed vectors in tests/golden/ (generated from that
 * same compiled reference by tests/golden/make_golden.py).
 */

#ifndef REAL
#error "include from fsm_oracle.c"
#endif

/* ttcr/ttcr_t.h:42-43 */
#define FSM_SMALL 1.e-4
#define FSM_SMALL2 (FSM_SMALL * FSM_SMALL)
/* Index of a coordinate.  The reference converts a double to its unsigned index type; for a point OUTSIDE the grid (the end game
 * of a ray with several source points within a cell diagonal moves curr_pt without a bounds check, e.g. ttcr/Grid2Drn.h:1596-1655)
 * the conversion of a negative value is undefined in C++ -- the compiled reference then reads far outside its arrays (garbage
 * traveltimes, or a crash).  Here and in the HIP kernels: negative -> 0, followed by the callers' upper clamps, i.e. the nearest
 * cell / node.  Identical to the plain conversion for every point inside the grid. */
#undef FSM_U32
#define FSM_U32(v) ((v) < 0 ? (uint32_t)0 : ((v) >= 4294967295.0 ? (uint32_t)4294967295u : (uint32_t)(v)))

/* ------------------------------------------------------------------ 3D -- */

/* Node coordinate as built by Grid3Drn::buildGridNodes (ttcr/Grid3Drn.h:388-399):
 * T1 x = xmin + ni*dx with ni an unsigned 32-bit integer. */
static REAL SFX(coord)(REAL cmin, uint32_t n, REAL d) { return cmin + n * d; }

/* Grid3Drn::update_node, ttcr/Grid3Drn.h:2902-2959 */
static void SFX(update_node3d)(REAL* T, const REAL* s, REAL dx, size_t nnx, size_t nny, size_t nnz,
                               size_t i, size_t j, size_t k) {
    const size_t ncx = nnx - 1, ncy = nny - 1, ncz = nnz - 1;
    REAL a1, a2, a3, t;

    if (k == 0)
        a1 = T[((k + 1) * nny + j) * nnx + i];
    else if (k == ncz)
        a1 = T[((k - 1) * nny + j) * nnx + i];
    else {
        a1 = T[((k - 1) * nny + j) * nnx + i];
        t = T[((k + 1) * nny + j) * nnx + i];
        a1 = a1 < t ? a1 : t;
    }

    if (j == 0)
        a2 = T[(k * nny + j + 1) * nnx + i];
    else if (j == ncy)
        a2 = T[(k * nny + j - 1) * nnx + i];
    else {
        a2 = T[(k * nny + j - 1) * nnx + i];
        t = T[(k * nny + j + 1) * nnx + i];
        a2 = a2 < t ? a2 : t;
    }

    if (i == 0)
        a3 = T[(k * nny + j) * nnx + i + 1];
    else if (i == ncx)
        a3 = T[(k * nny + j) * nnx + i - 1];
    else {
        a3 = T[(k * nny + j) * nnx + i - 1];
        t = T[(k * nny + j) * nnx + i + 1];
        a3 = a3 < t ? a3 : t;
    }

    if (a1 > a2) { REAL w = a1; a1 = a2; a2 = w; }
    if (a1 > a3) { REAL w = a1; a1 = a3; a3 = w; }
    if (a2 > a3) { REAL w = a2; a2 = a3; a3 = w; }

    const size_t n = (k * nny + j) * nnx + i;
    REAL fh = s[n] * dx;

    t = a1 + fh;
    if (t > a2) {
        /* double literals: evaluated in double also when REAL is float */
        t = 0.5 * (a1 + a2 + sqrt(2. * fh * fh - (a1 - a2) * (a1 - a2)));
        if (t > a3) {
            t = 1. / 3. * ((a1 + a2 + a3) + sqrt(-2. * a1 * a1 + 2. * a1 * a2 - 2. * a2 * a2 +
                                                 2. * a1 * a3 + 2. * a2 * a3 -
                                                 2. * a3 * a3 + 3. * fh * fh));
        }
    }
    if (t < T[n]) T[n] = t;
}

/* Grid3Drn::sweep, ttcr/Grid3Drn.h:2816-2899: 8 lexicographic Gauss-Seidel
 * sweeps, k outer / j / i inner, sign order (i,j,k) = +++ -++ +-+ --+ ++- -+- +-- --- */
static void SFX(sweep3d)(REAL* T, const REAL* s, const unsigned char* frozen, REAL dx, size_t nnx,
                         size_t nny, size_t nnz) {
    for (int dir = 0; dir < 8; ++dir) {
        const int ri = dir & 1, rj = (dir >> 1) & 1, rk = (dir >> 2) & 1;
        for (size_t kk = 0; kk < nnz; ++kk) {
            const size_t k = rk ? nnz - 1 - kk : kk;
            for (size_t jj = 0; jj < nny; ++jj) {
                const size_t j = rj ? nny - 1 - jj : jj;
                for (size_t ii = 0; ii < nnx; ++ii) {
                    const size_t i = ri ? nnx - 1 - ii : ii;
                    if (!frozen[(k * nny + j) * nnx + i]) SFX(update_node3d)(T, s, dx, nnx, nny, nnz, i, j, k);
                }
            }
        }
    }
}

/* Node3Dn::getDistance, ttcr/Node3Dn.h:142-144 */
static REAL SFX(dist3d)(REAL x, REAL y, REAL z, REAL px, REAL py, REAL pz) {
    return (REAL)sqrt((x - px) * (x - px) + (y - py) * (y - py) + (z - pz) * (z - pz));
}

/* Grid3Drn::getCellNo, ttcr/Grid3Drn.h:207-215, followed by the decomposition of
 * the flat cell number done in initFSM (:3530-3534); both in the reference's
 * unsigned 32-bit arithmetic. */
static void SFX(cell3d)(const SFX(fsm_grid3d) * g, REAL px, REAL py, REAL pz, ptrdiff_t* ci,
                        ptrdiff_t* cj, ptrdiff_t* ck) {
    const uint32_t ncx = (uint32_t)(g->nnx - 1), ncy = (uint32_t)(g->nny - 1);
    REAL x = g->xmax - px < FSM_SMALL2 ? (REAL)(g->xmax - .5 * g->dx) : px;
    REAL y = g->ymax - py < FSM_SMALL2 ? (REAL)(g->ymax - .5 * g->dx) : py;
    REAL z = g->zmax - pz < FSM_SMALL2 ? (REAL)(g->zmax - .5 * g->dx) : pz;
    uint32_t nx = FSM_U32(FSM_SMALL2 + (x - g->xmin) / g->dx);
    uint32_t ny = FSM_U32(FSM_SMALL2 + (y - g->ymin) / g->dx);
    uint32_t nz = FSM_U32(FSM_SMALL2 + (z - g->zmin) / g->dx);
    const ptrdiff_t cellNo = (ptrdiff_t)(uint32_t)(ny * ncx + nz * (ncx * ncy) + nx);
    const ptrdiff_t k = cellNo / ((ptrdiff_t)ncy * ncx);
    const ptrdiff_t j = (cellNo - k * (ptrdiff_t)ncy * ncx) / ncx;
    *ci = cellNo - (k * (ptrdiff_t)ncy + j) * ncx;
    *cj = j;
    *ck = k;
}

/* Grid3Drn::initFSM, ttcr/Grid3Drn.h:3487-3556 */
static void SFX(init3d)(const SFX(fsm_grid3d) * g, const REAL* s, REAL* T, unsigned char* frozen,
                        int n_src, const REAL* src, const REAL* t0, int npts) {
    const ptrdiff_t nnx = g->nnx, nny = g->nny, nnz = g->nnz;
    const ptrdiff_t ncx = nnx - 1, ncy = nny - 1, ncz = nnz - 1;
    for (int n = 0; n < n_src; ++n) {
        const REAL px = src[3 * n], py = src[3 * n + 1], pz = src[3 * n + 2];
        /* first node (in linear order) with |coord - Tx| < small on every axis
         * (Node3Dn::operator==, ttcr/Node3Dn.h:147-149).  The test is separable,
         * so the first match in x-fastest order is (first i, first j, first k). */
        ptrdiff_t fi = -1, fj = -1, fk = -1;
        for (ptrdiff_t i = 0; i < nnx && fi < 0; ++i)
            if (FABS(SFX(coord)(g->xmin, (uint32_t)i, g->dx) - px) < FSM_SMALL) fi = i;
        for (ptrdiff_t j = 0; j < nny && fj < 0; ++j)
            if (FABS(SFX(coord)(g->ymin, (uint32_t)j, g->dx) - py) < FSM_SMALL) fj = j;
        for (ptrdiff_t k = 0; k < nnz && fk < 0; ++k)
            if (FABS(SFX(coord)(g->zmin, (uint32_t)k, g->dx) - pz) < FSM_SMALL) fk = k;
        if (fi >= 0 && fj >= 0 && fk >= 0) {
            const ptrdiff_t i = fi, j = fj, k = fk;
            const size_t nn = (size_t)((k * nny + j) * nnx + i);
            T[nn] = t0[n];
            frozen[nn] = 1;
            for (ptrdiff_t kk = k - npts; kk <= k + npts; ++kk) {
                if (kk < 0 || kk > ncz) continue;
                for (ptrdiff_t jj = j - npts; jj <= j + npts; ++jj) {
                    if (jj < 0 || jj > ncy) continue;
                    for (ptrdiff_t ii = i - npts; ii <= i + npts; ++ii) {
                        if (ii >= 0 && ii <= ncx && !(ii == i && jj == j && kk == k)) {
                            const size_t m = (size_t)((kk * nny + jj) * nnx + ii);
                            REAL d = SFX(dist3d)(SFX(coord)(g->xmin, (uint32_t)ii, g->dx),
                                                 SFX(coord)(g->ymin, (uint32_t)jj, g->dx),
                                                 SFX(coord)(g->zmin, (uint32_t)kk, g->dx), px, py, pz);
                            REAL tt = t0[n] + d * s[m];
                            T[m] = tt;
                            frozen[m] = 1;
                        }
                    }
                }
            }
        } else {
            ptrdiff_t i, j, k;
            SFX(cell3d)(g, px, py, pz, &i, &j, &k);
            for (ptrdiff_t kk = k - (npts - 1); kk <= k + npts; ++kk) {
                if (kk < 0 || kk > ncz) continue;
                for (ptrdiff_t jj = j - (npts - 1); jj <= j + npts; ++jj) {
                    if (jj < 0 || jj > ncy) continue;
                    for (ptrdiff_t ii = i - (npts - 1); ii <= i + npts; ++ii) {
                        /* node (i,j,k) itself is skipped, as in the reference (:3541) */
                        if (ii >= 0 && ii <= ncx && !(ii == i && jj == j && kk == k)) {
                            const size_t m = (size_t)((kk * nny + jj) * nnx + ii);
                            REAL d = SFX(dist3d)(SFX(coord)(g->xmin, (uint32_t)ii, g->dx),
                                                 SFX(coord)(g->ymin, (uint32_t)jj, g->dx),
                                                 SFX(coord)(g->zmin, (uint32_t)kk, g->dx), px, py, pz);
                            REAL tt = t0[n] + d * s[m];
                            T[m] = tt;
                            frozen[m] = 1;
                        }
                    }
                }
            }
        }
    }
}


/* --------------------------------------------------------------- WENO3 ---- */
/* Grid3Drn::weno3_upwind, ttcr/Grid3Drn.h:3047-3075 (forward / backward one-sided third-order
 * WENO value); the 2-D code (ttcr/Grid2Drn.h:1078-1125) inlines the same expressions.  Every
 * named intermediate is a T1; the double literals make the right-hand sides double. */
static REAL SFX(weno_fwd)(REAL v1, REAL v2, REAL v3, REAL v4, REAL h) {
    const REAL eps = REAL_EPS;
    const REAL num = (v4 - 2.0 * v3 + v2);
    const REAL den = (v3 - 2.0 * v2 + v1);
    const REAL r = (eps + num * num) / (eps + den * den);
    const REAL w = 1.0 / (1.0 + 2.0 * r * r);
    const REAL ap = (1.0 - w) * (v3 - v1) / (2.0 * h) + w * (-v4 + 4.0 * v3 - 3.0 * v2) / (2.0 * h);
    return v2 + h * ap;
}
static REAL SFX(weno_bwd)(REAL v0, REAL v1, REAL v2, REAL v3, REAL h) {
    const REAL eps = REAL_EPS;
    const REAL num = (v2 - 2.0 * v1 + v0);
    const REAL den = (v3 - 2.0 * v2 + v1);
    const REAL r = (eps + num * num) / (eps + den * den);
    const REAL w = 1.0 / (1.0 + 2.0 * r * r);
    const REAL am = (1.0 - w) * (v3 - v1) / (2.0 * h) + w * (3.0 * v2 - 4.0 * v1 + v0) / (2.0 * h);
    return v2 - h * am;
}

/* One axis of update_node_weno3 (ttcr/Grid3Drn.h:3084-3196 for k, :3199-3313 j, :3316-3430 i;
 * 2-D ttcr/Grid2Drn.h:1076-1128): T points at the node, st is the stride along the axis, idx the
 * node's index on that axis, n the last index (cells), h the spacing.
 *   idx == 0    : first-order, T[+1]          idx == n   : first-order, T[-1]
 *   idx == 1    : min(weno_fwd, T[-1])        idx == n-1 : min(weno_bwd, T[+1])
 *   otherwise   : min(weno_fwd, weno_bwd)     (a<t ? a : t  selects, NaN-transparent like the reference) */
static REAL SFX(weno_axis)(const REAL* T, ptrdiff_t st, size_t idx, size_t n, REAL h) {
    REAL a, t;
    if (idx == 0) {
        a = T[st];
    } else if (idx == 1) {
        a = SFX(weno_fwd)(T[-st], T[0], T[st], T[2 * st], h);
        t = T[-st];
        a = a < t ? a : t;
    } else if (idx == n) {
        a = T[-st];
    } else if (idx == n - 1) {
        a = SFX(weno_bwd)(T[-2 * st], T[-st], T[0], T[st], h);
        t = T[st];
        a = a < t ? a : t;
    } else {
        a = SFX(weno_fwd)(T[-st], T[0], T[st], T[2 * st], h);
        t = SFX(weno_bwd)(T[-2 * st], T[-st], T[0], T[st], h);
        a = a < t ? a : t;
    }
    return a;
}

/* Grid3Drn::update_node_weno3, ttcr/Grid3Drn.h:3078-3484 */
static void SFX(update_node3d_weno)(REAL* T, const REAL* s, REAL dx, size_t nnx, size_t nny, size_t nnz,
                                    size_t i, size_t j, size_t k) {
    const size_t n = (k * nny + j) * nnx + i;
    REAL a1 = SFX(weno_axis)(T + n, (ptrdiff_t)(nnx * nny), k, nnz - 1, dx);
    REAL a2 = SFX(weno_axis)(T + n, (ptrdiff_t)nnx, j, nny - 1, dx);
    REAL a3 = SFX(weno_axis)(T + n, 1, i, nnx - 1, dx);
    REAL t;
    if (a1 > a2) { REAL w = a1; a1 = a2; a2 = w; }
    if (a1 > a3) { REAL w = a1; a1 = a3; a3 = w; }
    if (a2 > a3) { REAL w = a2; a2 = a3; a3 = w; }
    REAL fh = s[n] * dx;
    t = a1 + fh;
    if (t > a2) {
        t = 0.5 * (a1 + a2 + sqrt(2. * fh * fh - (a1 - a2) * (a1 - a2)));
        if (t > a3) {
            t = 1. / 3. * ((a1 + a2 + a3) + sqrt(-2. * a1 * a1 + 2. * a1 * a2 - 2. * a2 * a2 + 2. * a1 * a3 +
                                                 2. * a2 * a3 - 2. * a3 * a3 + 3. * fh * fh));
        }
    }
    if (t < T[n]) T[n] = t;
}

/* Grid3Drn::sweep_weno3, ttcr/Grid3Drn.h:2962-3044: same 8 directions as sweep */
static void SFX(sweep3d_weno)(REAL* T, const REAL* s, const unsigned char* frozen, REAL dx, size_t nnx,
                              size_t nny, size_t nnz) {
    for (int dir = 0; dir < 8; ++dir) {
        const int ri = dir & 1, rj = (dir >> 1) & 1, rk = (dir >> 2) & 1;
        for (size_t kk = 0; kk < nnz; ++kk) {
            const size_t k = rk ? nnz - 1 - kk : kk;
            for (size_t jj = 0; jj < nny; ++jj) {
                const size_t j = rj ? nny - 1 - jj : jj;
                for (size_t ii = 0; ii < nnx; ++ii) {
                    const size_t i = ri ? nnx - 1 - ii : ii;
                    if (!frozen[(k * nny + j) * nnx + i]) SFX(update_node3d_weno)(T, s, dx, nnx, nny, nnz, i, j, k);
                }
            }
        }
    }
}

/* Grid3Drn ctor (ttcr/Grid3Drn.h:67-78) + Grid3Drnfs ctor (ttcr/Grid3Drnfs.h:39-50)
 * + translateOrigin handling of buildGridNodes (ttcr/Grid3Drn.h:362-372). */
void SFX(fsm_grid3d_init)(SFX(fsm_grid3d) * g, uint32_t ncx, uint32_t ncy, uint32_t ncz, REAL dx,
                          REAL xmin, REAL ymin, REAL zmin, int translate) {
    g->nnx = (size_t)ncx + 1;
    g->nny = (size_t)ncy + 1;
    g->nnz = (size_t)ncz + 1;
    g->dx = dx;
    g->xmin = xmin; g->ymin = ymin; g->zmin = zmin;
    g->xmax = xmin + ncx * dx;
    g->ymax = ymin + ncy * dx;
    g->zmax = zmin + ncz * dx;
    g->ox = g->oy = g->oz = 0;
    if (translate) {
        g->ox = xmin; g->oy = ymin; g->oz = zmin;
        g->xmax -= g->xmin; g->ymax -= g->ymin; g->zmax -= g->zmin;
        g->xmin = 0; g->ymin = 0; g->zmin = 0;
    }
}

/* Grid3Drn::checkPts, ttcr/Grid3Drn.h:771-790.  Returns 1 if a point is outside. */
int SFX(fsm_outside3d)(const SFX(fsm_grid3d) * g, int n, const REAL* p) {
    for (int m = 0; m < n; ++m) {
        if (p[3 * m] < g->xmin || p[3 * m] > g->xmax || p[3 * m + 1] < g->ymin ||
            p[3 * m + 1] > g->ymax || p[3 * m + 2] < g->zmin || p[3 * m + 2] > g->zmax)
            return 1;
    }
    return 0;
}

/* Grid3Drcfs::setSlowness, ttcr/Grid3Drcfs.h:88-171: cell -> node slowness.
 * Written as one generic loop: a node averages the cells that touch it; the
 * reference's summation order inside each branch is k outer / j / i inner with the
 * "+0" neighbour first (i, then i-1; ...), except on the x-min/x-max faces where k
 * is innermost; this loop reproduces both (the factors .5/.25/.125 are exact). */
void SFX(fsm_cells_to_nodes3d)(size_t ncx, size_t ncy, size_t ncz, const REAL* sc, REAL* sn) {
    const size_t nnx = ncx + 1, nny = ncy + 1, nnz = ncz + 1;
    for (size_t k = 0; k < nnz; ++k)
        for (size_t j = 0; j < nny; ++j)
            for (size_t i = 0; i < nnx; ++i) {
                /* candidate cell indices along each axis: c, then c-1 */
                size_t ci[2], cj[2], ck[2];
                int ni = 0, nj = 0, nk = 0;
                if (i < ncx) ci[ni++] = i;
                if (i > 0) ci[ni++] = i - 1;
                if (j < ncy) cj[nj++] = j;
                if (j > 0) cj[nj++] = j - 1;
                if (k < ncz) ck[nk++] = k;
                if (k > 0) ck[nk++] = k - 1;
                REAL sum = 0;
                int first = 1;
                if (ni == 1) {
                    /* x-min / x-max faces (:139-146): k varies fastest in the reference's sum */
                    for (int b = 0; b < nj; ++b)
                        for (int a = 0; a < nk; ++a) {
                            REAL v = sc[(ck[a] * ncy + cj[b]) * ncx + ci[0]];
                            if (first) { sum = v; first = 0; } else sum = sum + v;
                        }
                } else {
                    for (int a = 0; a < nk; ++a)
                        for (int b = 0; b < nj; ++b)
                            for (int c = 0; c < ni; ++c) {
                                REAL v = sc[(ck[a] * ncy + cj[b]) * ncx + ci[c]];
                                if (first) { sum = v; first = 0; } else sum = sum + v;
                            }
                }
                const int cnt = ni * nj * nk;
                REAL out;
                if (cnt == 1) out = sum;
                else if (cnt == 2) out = (REAL)(0.5 * sum);
                else if (cnt == 4) out = (REAL)(0.25 * sum);
                else out = (REAL)(0.125 * sum);
                sn[(k * nny + j) * nnx + i] = out;
            }
}

/* Driver loop of Grid3Drnfs::raytrace (ttcr/Grid3Drnfs.h:84-155): weno3 == false branch
 * (:137-153) or the two-stage weno3 branch (:104-136: first-order sweeps to convergence, then
 * sweep_weno3 to convergence, frozen box npts = 2), preceded by reinit (:92-94) and initFSM (:97-100).
 * eps is the per-node tolerance; the ctor scales it by the node count (:49).
 * src are in grid coordinates (origin already subtracted when translated).
 * change_hist (optional, length maxit) receives the L1 change of every iteration.
 * Returns niter. */
int SFX(fsm_solve3d)(const SFX(fsm_grid3d) * g, const REAL* s, int n_src, const REAL* src,
                     const REAL* t0, REAL eps, int maxit, int weno, REAL* T, REAL* change_hist, int* niterw_out) {
    const size_t N = g->nnx * g->nny * g->nnz;
    REAL epsilon = eps;
    epsilon *= (REAL)N;
    unsigned char* frozen = (unsigned char*)calloc(N, 1);
    REAL* times = (REAL*)malloc(N * sizeof(REAL));
    for (size_t n = 0; n < N; ++n) T[n] = REAL_MAX;
    SFX(init3d)(g, s, T, frozen, n_src, src, t0, weno ? 2 : 1);
    for (size_t n = 0; n < N; ++n) times[n] = T[n];
    REAL change = REAL_MAX;
    int niter = 0, niterw = 0;
    while (change >= epsilon && niter < maxit) {
        SFX(sweep3d)(T, s, frozen, g->dx, g->nnx, g->nny, g->nnz);
        change = 0.0;
        for (size_t n = 0; n < N; ++n) {
            REAL dt = FABS(times[n] - T[n]);
            change += dt;
            times[n] = T[n];
        }
        if (change_hist) change_hist[niter] = change;
        niter++;
    }
    if (weno) { /* second stage, ttcr/Grid3Drnfs.h:125-136 */
        change = REAL_MAX;
        while (change >= epsilon && niterw < maxit) {
            SFX(sweep3d_weno)(T, s, frozen, g->dx, g->nnx, g->nny, g->nnz);
            change = 0.0;
            for (size_t n = 0; n < N; ++n) {
                REAL dt = FABS(times[n] - T[n]);
                change += dt;
                times[n] = T[n];
            }
            if (change_hist) change_hist[maxit + niterw] = change;
            niterw++;
        }
    }
    if (niterw_out) *niterw_out = niterw;
    free(times);
    free(frozen);
    return niter;
}

/* Grid3Drn::getIJK (ttcr/Grid3Drn.h:233-237) + getTraveltime (:794-930):
 * node value / linear / bilinear / trilinear interpolation at a receiver. */
REAL SFX(fsm_interp3d)(const SFX(fsm_grid3d) * g, const REAL* T, REAL px, REAL py, REAL pz) {
    const size_t nnx = g->nnx, nny = g->nny, nnz = g->nnz;
    const REAL xmin = g->xmin, ymin = g->ymin, zmin = g->zmin, dx = g->dx, dy = g->dx, dz = g->dx;
    const uint32_t i = FSM_U32(FSM_SMALL2 + (px - xmin) / dx);
    const uint32_t j = FSM_U32(FSM_SMALL2 + (py - ymin) / dy);
    const uint32_t k = FSM_U32(FSM_SMALL2 + (pz - zmin) / dz);
    const int onx = FABS(px - (xmin + i * dx)) < FSM_SMALL2;
    const int ony = FABS(py - (ymin + j * dy)) < FSM_SMALL2;
    const int onz = FABS(pz - (zmin + k * dz)) < FSM_SMALL2;
/* The index is a quotient rounded in REAL plus 1e-8, "on the plane" an ABSOLUTE distance below 1e-8: a point a
 * rounding error below the last plane of an axis (or further away than 1e-8 when dx > 1) gets the last node as
 * its lower index without counting as on the plane, and the reference then reads node index+1 -- past the row,
 * or past the array in the last plane.  Such an index is clamped to the last node here (and in the kernel); its
 * weight is the distance to the lower plane, ~0.  Indices in range are untouched. */
#define TT_CL(v, n) ((size_t)(v) < (n) ? (size_t)(v) : (n) - 1)
#define TT(ii, jj, kk) T[(TT_CL(kk, nnz) * nny + TT_CL(jj, nny)) * nnx + TT_CL(ii, nnx)]
    REAL tt;
    if (onx && ony && onz) {
        return TT(i, j, k);
    } else if (onx && ony) {
        REAL t1 = TT(i, j, k), t2 = TT(i, j, k + 1);
        REAL w1 = (zmin + (k + 1) * dz - pz) / dz, w2 = (pz - (zmin + k * dz)) / dz;
        tt = t1 * w1 + t2 * w2;
    } else if (onx && onz) {
        REAL t1 = TT(i, j, k), t2 = TT(i, j + 1, k);
        REAL w1 = (ymin + (j + 1) * dy - py) / dy, w2 = (py - (ymin + j * dy)) / dy;
        tt = t1 * w1 + t2 * w2;
    } else if (ony && onz) {
        REAL t1 = TT(i, j, k), t2 = TT(i + 1, j, k);
        REAL w1 = (xmin + (i + 1) * dx - px) / dx, w2 = (px - (xmin + i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    } else if (onx) {
        REAL t1 = TT(i, j, k), t2 = TT(i, j, k + 1), t3 = TT(i, j + 1, k), t4 = TT(i, j + 1, k + 1);
        REAL w1 = (zmin + (k + 1) * dz - pz) / dz, w2 = (pz - (zmin + k * dz)) / dz;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (ymin + (j + 1) * dy - py) / dy;
        w2 = (py - (ymin + j * dy)) / dy;
        tt = t1 * w1 + t2 * w2;
    } else if (ony) {
        REAL t1 = TT(i, j, k), t2 = TT(i, j, k + 1), t3 = TT(i + 1, j, k), t4 = TT(i + 1, j, k + 1);
        REAL w1 = (zmin + (k + 1) * dz - pz) / dz, w2 = (pz - (zmin + k * dz)) / dz;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (xmin + (i + 1) * dx - px) / dx;
        w2 = (px - (xmin + i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    } else if (onz) {
        REAL t1 = TT(i, j, k), t2 = TT(i, j + 1, k), t3 = TT(i + 1, j, k), t4 = TT(i + 1, j + 1, k);
        REAL w1 = (ymin + (j + 1) * dy - py) / dy, w2 = (py - (ymin + j * dy)) / dy;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (xmin + (i + 1) * dx - px) / dx;
        w2 = (px - (xmin + i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    } else {
        REAL t1 = TT(i, j, k), t2 = TT(i, j, k + 1), t3 = TT(i, j + 1, k), t4 = TT(i, j + 1, k + 1);
        REAL t5 = TT(i + 1, j, k), t6 = TT(i + 1, j, k + 1), t7 = TT(i + 1, j + 1, k),
             t8 = TT(i + 1, j + 1, k + 1);
        REAL w1 = (zmin + (k + 1) * dz - pz) / dz, w2 = (pz - (zmin + k * dz)) / dz;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        t3 = t5 * w1 + t6 * w2;
        t4 = t7 * w1 + t8 * w2;
        w1 = (ymin + (j + 1) * dy - py) / dy;
        w2 = (py - (ymin + j * dy)) / dy;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (xmin + (i + 1) * dx - px) / dx;
        w2 = (px - (xmin + i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    }
#undef TT
#undef TT_CL
    return tt;
}

/* --------------------------------------------- traveltime from raypath (3-D) -- */
/* Interpolator<T>::linear / bilinear / trilinear, ttcr/Interpolator.h:37-85 */
static REAL SFX(lin1)(const REAL x[3], const REAL s[2]) {
    return (s[0] * (x[2] - x[0]) + s[1] * (x[0] - x[1])) / (x[2] - x[1]);
}
static REAL SFX(lin2)(const REAL x[3], const REAL y[3], const REAL s[4]) {
    return (s[0] * (x[2] - x[0]) * (y[2] - y[0]) + s[1] * (x[2] - x[0]) * (y[0] - y[1]) +
            s[2] * (x[0] - x[1]) * (y[2] - y[0]) + s[3] * (x[0] - x[1]) * (y[0] - y[1])) /
           ((x[2] - x[1]) * (y[2] - y[1]));
}
static REAL SFX(lin3)(const REAL x[3], const REAL y[3], const REAL z[3], const REAL s[8]) {
    return (s[0] * (x[2] - x[0]) * (y[2] - y[0]) * (z[2] - z[0]) + s[1] * (x[2] - x[0]) * (y[2] - y[0]) * (z[0] - z[1]) +
            s[2] * (x[2] - x[0]) * (y[0] - y[1]) * (z[2] - z[0]) + s[3] * (x[2] - x[0]) * (y[0] - y[1]) * (z[0] - z[1]) +
            s[4] * (x[0] - x[1]) * (y[2] - y[0]) * (z[2] - z[0]) + s[5] * (x[0] - x[1]) * (y[2] - y[0]) * (z[0] - z[1]) +
            s[6] * (x[0] - x[1]) * (y[0] - y[1]) * (z[2] - z[0]) + s[7] * (x[0] - x[1]) * (y[0] - y[1]) * (z[0] - z[1])) /
           ((x[2] - x[1]) * (y[2] - y[1]) * (z[2] - z[1]));
}

/* Grid3Drn::computeSlowness(pt, isTranslated = true), ttcr/Grid3Drn.h:2451-2676.
 * iv = processVel (interpolate velocity instead of slowness). */
static REAL SFX(slowness_at3d)(const SFX(fsm_grid3d) * g, const REAL* sn, REAL px, REAL py, REAL pz, int iv) {
    const size_t nnx = g->nnx, nny = g->nny, nnz = g->nnz;
    const REAL xmin = g->xmin, ymin = g->ymin, zmin = g->zmin, dx = g->dx, dy = g->dx, dz = g->dx;
    ptrdiff_t onX = -1, onY = -1, onZ = -1;
    for (size_t n = 0; n < nnx; ++n)
        if (FABS(px - (xmin + n * dx)) < FSM_SMALL2) { onX = (ptrdiff_t)n; break; }
    for (size_t n = 0; n < nny; ++n)
        if (FABS(py - (ymin + n * dy)) < FSM_SMALL2) { onY = (ptrdiff_t)n; break; }
    for (size_t n = 0; n < nnz; ++n)
        if (FABS(pz - (zmin + n * dz)) < FSM_SMALL2) { onZ = (ptrdiff_t)n; break; }
/* The reference takes the cell index as (T2)(small + (p - min)/d) with small = 1e-4 but calls a point "on" a plane
 * only within small^2: a point closer than 1e-4 cell to the LAST plane of an axis, yet not on it, gets the cell
 * beyond the grid and the reference reads one node past its array there (heap garbage: its result is not
 * reproducible).  Restated in range: such an index is clamped to the last node.  In-range indices are untouched. */
#define SN_CL(v, n) ((size_t)(v) < (size_t)(n) ? (size_t)(v) : (size_t)(n) - 1)
#define SN_AT(ii, jj, kk) sn[(SN_CL(kk, nnz) * nny + SN_CL(jj, nny)) * nnx + SN_CL(ii, nnx)]
#define SN(ii, jj, kk) (iv ? (REAL)(1.0 / SN_AT(ii, jj, kk)) : SN_AT(ii, jj, kk))
#define RET(v) return iv ? (REAL)(1.0 / (v)) : (v)
    REAL s[8], x[3], y[3], z[3];
    if (onX != -1 && onY != -1 && onZ != -1) {
        return sn[((size_t)onZ * nny + onY) * nnx + onX];
    } else if (onX != -1 && onY != -1) {
        uint32_t k = FSM_U32(FSM_SMALL + (pz - zmin) / dz);
        s[0] = SN(onX, onY, k); s[1] = SN(onX, onY, k + 1);
        x[0] = pz; x[1] = zmin + k * dz; x[2] = zmin + (k + 1) * dz;
        RET(SFX(lin1)(x, s));
    } else if (onX != -1 && onZ != -1) {
        uint32_t j = FSM_U32(FSM_SMALL + (py - ymin) / dy);
        s[0] = SN(onX, j, onZ); s[1] = SN(onX, j + 1, onZ);
        x[0] = py; x[1] = ymin + j * dy; x[2] = ymin + (j + 1) * dy;
        RET(SFX(lin1)(x, s));
    } else if (onY != -1 && onZ != -1) {
        uint32_t i = FSM_U32(FSM_SMALL + (px - xmin) / dx);
        s[0] = SN(i, onY, onZ); s[1] = SN(i + 1, onY, onZ);
        x[0] = px; x[1] = xmin + i * dx; x[2] = xmin + (i + 1) * dx;
        RET(SFX(lin1)(x, s));
    } else if (onX != -1) {
        uint32_t j = FSM_U32(FSM_SMALL + (py - ymin) / dy);
        uint32_t k = FSM_U32(FSM_SMALL + (pz - zmin) / dz);
        s[0] = SN(onX, j, k); s[1] = SN(onX, j, k + 1); s[2] = SN(onX, j + 1, k); s[3] = SN(onX, j + 1, k + 1);
        x[0] = py; y[0] = pz; x[1] = ymin + j * dy; y[1] = zmin + k * dz; x[2] = ymin + (j + 1) * dy; y[2] = zmin + (k + 1) * dz;
        RET(SFX(lin2)(x, y, s));
    } else if (onY != -1) {
        uint32_t i = FSM_U32(FSM_SMALL + (px - xmin) / dx);
        uint32_t k = FSM_U32(FSM_SMALL + (pz - zmin) / dz);
        s[0] = SN(i, onY, k); s[1] = SN(i, onY, k + 1); s[2] = SN(i + 1, onY, k); s[3] = SN(i + 1, onY, k + 1);
        x[0] = px; y[0] = pz; x[1] = xmin + i * dx; y[1] = zmin + k * dz; x[2] = xmin + (i + 1) * dx; y[2] = zmin + (k + 1) * dz;
        RET(SFX(lin2)(x, y, s));
    } else if (onZ != -1) {
        uint32_t i = FSM_U32(FSM_SMALL + (px - xmin) / dx);
        uint32_t j = FSM_U32(FSM_SMALL + (py - ymin) / dy);
        s[0] = SN(i, j, onZ); s[1] = SN(i, j + 1, onZ); s[2] = SN(i + 1, j, onZ); s[3] = SN(i + 1, j + 1, onZ);
        x[0] = px; y[0] = py; x[1] = xmin + i * dx; y[1] = ymin + j * dy; x[2] = xmin + (i + 1) * dx; y[2] = ymin + (j + 1) * dy;
        RET(SFX(lin2)(x, y, s));
    } else {
        uint32_t i = FSM_U32(FSM_SMALL + (px - xmin) / dx);
        uint32_t j = FSM_U32(FSM_SMALL + (py - ymin) / dy);
        uint32_t k = FSM_U32(FSM_SMALL + (pz - zmin) / dz);
        s[0] = SN(i, j, k); s[1] = SN(i, j, k + 1); s[2] = SN(i, j + 1, k); s[3] = SN(i, j + 1, k + 1);
        s[4] = SN(i + 1, j, k); s[5] = SN(i + 1, j, k + 1); s[6] = SN(i + 1, j + 1, k); s[7] = SN(i + 1, j + 1, k + 1);
        x[0] = px; y[0] = py; z[0] = pz;
        x[1] = xmin + i * dx; y[1] = ymin + j * dy; z[1] = zmin + k * dz;
        x[2] = xmin + (i + 1) * dx; y[2] = ymin + (j + 1) * dy; z[2] = zmin + (k + 1) * dz;
        RET(SFX(lin3)(x, y, z, s));
    }
#undef SN
#undef RET
}

/* Grid3Drn::grad(g, pt, nt), ttcr/Grid3Drn.h:1033-1100: 4th-order centred operator on the
 * interpolated field.  (x uses pt.x - dx, y and z use pt - d/2.0 -- as in the reference.) */
/* Grid3D::computeSlowness(pt) as the Cython layer calls it (src/ttcrpy/rgrid.pyx:824, get_s0): the function above
 * on a point given in grid coordinates (the caller subtracts the origin of a translated grid, :2453-2455) */
REAL SFX(fsm_compute_slowness3d)(const SFX(fsm_grid3d) * g, const REAL* sn, REAL px, REAL py, REAL pz, int iv) {
    return SFX(slowness_at3d)(g, sn, px, py, pz, iv);
}

static void SFX(grad3d)(const SFX(fsm_grid3d) * g, const REAL* T, REAL ptx, REAL pty, REAL ptz, REAL* gx, REAL* gy, REAL* gz) {
    static const REAL k1 = 1. / 24.;
    static const REAL k2 = 9. / 8.;
    const REAL dx = g->dx, dy = g->dx, dz = g->dx;
    REAL p1 = ptx - dx;
    REAL p2 = p1 + 0.5 * dx, p3 = p1 + 1.5 * dx, p4 = p1 + 2.0 * dx;
    if (p1 <= g->xmin) {
        p1 = g->xmin; p2 = p1 + 0.5 * dx; p3 = p1 + 1.5 * dx; p4 = p1 + 2.0 * dx;
    } else if (p4 >= g->xmax) {
        p4 = g->xmax; p3 = p4 - 0.5 * dx; p2 = p4 - 1.5 * dx; p1 = p4 - 2.0 * dx;
    }
    *gx = (k1 * SFX(fsm_interp3d)(g, T, p1, pty, ptz) - k2 * SFX(fsm_interp3d)(g, T, p2, pty, ptz) +
           k2 * SFX(fsm_interp3d)(g, T, p3, pty, ptz) - k1 * SFX(fsm_interp3d)(g, T, p4, pty, ptz)) / dx;
    p1 = pty - dy / 2.0;
    p2 = p1 + 0.5 * dy; p3 = p1 + 1.5 * dy; p4 = p1 + 2.0 * dy;
    if (p1 <= g->ymin) {
        p1 = g->ymin; p2 = p1 + 0.5 * dy; p3 = p1 + 1.5 * dy; p4 = p1 + 2.0 * dy;
    } else if (p4 >= g->ymax) {
        p4 = g->ymax; p3 = p4 - 0.5 * dy; p2 = p4 - 1.5 * dy; p1 = p4 - 2.0 * dy;
    }
    *gy = (k1 * SFX(fsm_interp3d)(g, T, ptx, p1, ptz) - k2 * SFX(fsm_interp3d)(g, T, ptx, p2, ptz) +
           k2 * SFX(fsm_interp3d)(g, T, ptx, p3, ptz) - k1 * SFX(fsm_interp3d)(g, T, ptx, p4, ptz)) / dy;
    p1 = ptz - dz / 2.0;
    p2 = p1 + 0.5 * dz; p3 = p1 + 1.5 * dz; p4 = p1 + 2.0 * dz;
    if (p1 <= g->zmin) {
        p1 = g->zmin; p2 = p1 + 0.5 * dz; p3 = p1 + 1.5 * dz; p4 = p1 + 2.0 * dz;
    } else if (p4 >= g->zmax) {
        p4 = g->zmax; p3 = p4 - 0.5 * dz; p2 = p4 - 1.5 * dz; p1 = p4 - 2.0 * dz;
    }
    *gz = (k1 * SFX(fsm_interp3d)(g, T, ptx, pty, p1) - k2 * SFX(fsm_interp3d)(g, T, ptx, pty, p2) +
           k2 * SFX(fsm_interp3d)(g, T, ptx, pty, p3) - k1 * SFX(fsm_interp3d)(g, T, ptx, pty, p4)) / dz;
}

static int SFX(sgn)(REAL v) { return v == 0 ? 0 : (signbit(v) ? -1 : 1); } /* boost::math::sign */
static REAL SFX(dist3)(const REAL a[3], const REAL b[3]) { /* sxyz::getDistance, ttcr/ttcr_t.h:289-291 */
    return (REAL)sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
}

/* one "advance curr along g to the next grid plane" block of getTraveltimeFromRaypath
 * (ttcr/Grid3Drn.h:1131-1165 and :1193-1223) */
static void SFX(step_to_plane)(const SFX(fsm_grid3d) * g, REAL cur[3], const REAL gv[3]) {
    const REAL dx = g->dx, dy = g->dx, dz = g->dx;
    const ptrdiff_t i = (ptrdiff_t)(FSM_SMALL2 + (cur[0] - g->xmin) / dx);
    const ptrdiff_t j = (ptrdiff_t)(FSM_SMALL2 + (cur[1] - g->ymin) / dy);
    const ptrdiff_t k = (ptrdiff_t)(FSM_SMALL2 + (cur[2] - g->zmin) / dz);
    REAL xp = g->xmin + dx * (i + (SFX(sgn)(gv[0]) > 0.0 ? 1.0 : 0.0));
    REAL yp = g->ymin + dy * (j + (SFX(sgn)(gv[1]) > 0.0 ? 1.0 : 0.0));
    REAL zp = g->zmin + dz * (k + (SFX(sgn)(gv[2]) > 0.0 ? 1.0 : 0.0));
    if (FABS(xp - cur[0]) < FSM_SMALL2) xp += dx * SFX(sgn)(gv[0]);
    if (FABS(yp - cur[1]) < FSM_SMALL2) yp += dy * SFX(sgn)(gv[1]);
    if (FABS(zp - cur[2]) < FSM_SMALL2) zp += dz * SFX(sgn)(gv[2]);
    REAL tx = gv[0] != 0.0 ? (xp - cur[0]) / gv[0] : REAL_MAX;
    REAL ty = gv[1] != 0.0 ? (yp - cur[1]) / gv[1] : REAL_MAX;
    REAL tz = gv[2] != 0.0 ? (zp - cur[2]) / gv[2] : REAL_MAX;
    if (tx < ty && tx < tz) {
        cur[0] += tx * gv[0]; cur[1] += tx * gv[1]; cur[2] += tx * gv[2];
        cur[0] = xp;
    } else if (ty < tz) {
        cur[0] += ty * gv[0]; cur[1] += ty * gv[1]; cur[2] += ty * gv[2];
        cur[1] = yp;
    } else {
        cur[0] += tz * gv[0]; cur[1] += tz * gv[1]; cur[2] += tz * gv[2];
        cur[2] = zp;
    }
}

/* Grid3Drn::getTraveltimeFromRaypath, ttcr/Grid3Drn.h:1103-1243: steepest-descent walk from the
 * receiver to the source through the traveltime field, trapezoidal slowness integration.
 * Points are in grid coordinates.  Returns 0, or 1 when the ray leaves the grid (the reference
 * throws std::runtime_error), or 2 when max_steps is exhausted (the reference would loop). */
int SFX(fsm_tt_from_raypath3d)(const SFX(fsm_grid3d) * g, const REAL* sn, const REAL* T, int n_src, const REAL* src,
                               const REAL* t0, const REAL rx[3], int iv, long max_steps, REAL* tt_out) {
    REAL tt = 0.0, s1, s2;
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[3 * ns] && rx[1] == src[3 * ns + 1] && rx[2] == src[3 * ns + 2]) { *tt_out = t0[ns]; return 0; }
    REAL prev[3] = {rx[0], rx[1], rx[2]}, cur[3] = {rx[0], rx[1], rx[2]}, gv[3];
    s1 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
    const REAL dx = g->dx;
    const REAL maxDist = (REAL)sqrt(dx * dx + dx * dx + dx * dx);
    int reached = 0;
    long steps = 0;
    while (!reached) {
        if (++steps > max_steps) return 2;
        SFX(grad3d)(g, T, cur[0], cur[1], cur[2], &gv[0], &gv[1], &gv[2]);
        gv[0] *= (REAL)-1.0; gv[1] *= (REAL)-1.0; gv[2] *= (REAL)-1.0;
        SFX(step_to_plane)(g, cur, gv);
        if (cur[0] < g->xmin || cur[0] > g->xmax || cur[1] < g->ymin || cur[1] > g->ymax || cur[2] < g->zmin ||
            cur[2] > g->zmax)
            return 1;
        s2 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
        tt += 0.5 * (s1 + s2) * SFX(dist3)(prev, cur);
        s1 = s2;
        prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
        for (int ns = 0; ns < n_src; ++ns) {
            const REAL* tx = src + 3 * ns;
            REAL dist = SFX(dist3)(cur, tx);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1]; gv[2] = tx[2] - cur[2];
                SFX(step_to_plane)(g, cur, gv);
                if (SFX(dist3)(cur, prev) > dist || (cur[0] == tx[0] && cur[1] == tx[1] && cur[2] == tx[2])) {
                    s2 = SFX(slowness_at3d)(g, sn, tx[0], tx[1], tx[2], iv);
                    tt += t0[ns] + 0.5 * (s1 + s2) * SFX(dist3)(prev, tx);
                } else {
                    s2 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
                    tt += 0.5 * (s1 + s2) * SFX(dist3)(prev, cur);
                    s1 = s2;
                    s2 = SFX(slowness_at3d)(g, sn, tx[0], tx[1], tx[2], iv);
                    tt += t0[ns] + 0.5 * (s1 + s2) * SFX(dist3)(cur, tx);
                }
                reached = 1;
            }
        }
    }
    *tt_out = tt;
    return 0;
}

/* Grid3Drn::getRaypath(Tx, t0, Rx, r_data, tt, threadNo), ttcr/Grid3Drn.h:1339-1500: the same walk as
 * getTraveltimeFromRaypath, recording every point; `back` is r_data.back().  pts holds xyz triples
 * (capacity cap points).  Returns 0 ok, 1 ray left the grid (the reference throws), 2 step limit,
 * 3 capacity exceeded (npts still counts). */
int SFX(fsm_raypath3d)(const SFX(fsm_grid3d) * g, const REAL* sn, const REAL* T, int n_src, const REAL* src,
                       const REAL* t0, const REAL rx[3], int iv, long max_steps, REAL* tt_out, REAL* pts, long cap,
                       long* npts) {
    REAL tt = 0.0, s1, s2;
    long np = 0;
    int over = 0;
#define FSM_PUSH(P) do { if (np < cap) { pts[3 * np] = (P)[0]; pts[3 * np + 1] = (P)[1]; pts[3 * np + 2] = (P)[2]; } else over = 1; \
                         back[0] = (P)[0]; back[1] = (P)[1]; back[2] = (P)[2]; ++np; } while (0)
    REAL back[3], cur[3] = {rx[0], rx[1], rx[2]}, gv[3];
    FSM_PUSH(rx);
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[3 * ns] && rx[1] == src[3 * ns + 1] && rx[2] == src[3 * ns + 2]) { *tt_out = t0[ns]; *npts = np; return 0; }
    s1 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
    const REAL dx = g->dx;
    const REAL maxDist = (REAL)sqrt(dx * dx + dx * dx + dx * dx);
    int reached = 0;
    long steps = 0;
    while (!reached) {
        if (++steps > max_steps) { *npts = np; return 2; }
        SFX(grad3d)(g, T, cur[0], cur[1], cur[2], &gv[0], &gv[1], &gv[2]);
        gv[0] *= (REAL)-1.0; gv[1] *= (REAL)-1.0; gv[2] *= (REAL)-1.0;
        SFX(step_to_plane)(g, cur, gv);
        if (cur[0] < g->xmin || cur[0] > g->xmax || cur[1] < g->ymin || cur[1] > g->ymax || cur[2] < g->zmin ||
            cur[2] > g->zmax) { *npts = np; return 1; }
        s2 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
        tt += 0.5 * (s1 + s2) * SFX(dist3)(back, cur);
        s1 = s2;
        FSM_PUSH(cur);
        for (int ns = 0; ns < n_src; ++ns) {
            const REAL* tx = src + 3 * ns;
            REAL dist = SFX(dist3)(cur, tx);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1]; gv[2] = tx[2] - cur[2];
                SFX(step_to_plane)(g, cur, gv);
                if (SFX(dist3)(cur, back) > dist || (cur[0] == tx[0] && cur[1] == tx[1] && cur[2] == tx[2])) {
                    s2 = SFX(slowness_at3d)(g, sn, tx[0], tx[1], tx[2], iv);
                    tt += t0[ns] + 0.5 * (s1 + s2) * SFX(dist3)(back, tx);
                    FSM_PUSH(tx);
                } else {
                    s2 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
                    tt += 0.5 * (s1 + s2) * SFX(dist3)(back, cur);
                    FSM_PUSH(cur);
                    s1 = s2;
                    s2 = SFX(slowness_at3d)(g, sn, tx[0], tx[1], tx[2], iv);
                    tt += t0[ns] + 0.5 * (s1 + s2) * SFX(dist3)(cur, tx);
                    FSM_PUSH(tx);
                }
                reached = 1;
            }
        }
    }
#undef FSM_PUSH
    *tt_out = tt;
    *npts = np;
    return over ? 3 : 0;
}

/* Terms of the matrix M of one ray segment, ttcr/Grid3Drn.h:1586-1623 (and :1674-1707, :1715-1747, :1755-1787, the same
 * block four times): the eight nodes around the segment's mid-point get -s^2 ds w, entries with the same node index are
 * merged (linear search, push order kept).  As the reference has it: the weights use iv*dx without xmin, and ix+1 can
 * lie one node past the grid (the Python layer drops entries whose node index is not below the node count,
 * src/ttcrpy/rgrid.pyx:1180-1186).  mj / mv: capacity cap entries; *nm counts all of them. */
static void SFX(m_terms3d)(const SFX(fsm_grid3d) * g, const REAL* sn, const REAL mid[3], REAL ds, int iv, long long* mj, REAL* mv,
                           long cap, long* nm) {
    const REAL dx = g->dx, dy = g->dx, dz = g->dx;
    REAL s = SFX(slowness_at3d)(g, sn, mid[0], mid[1], mid[2], iv);
    s *= s;
    const size_t ix = (size_t)((mid[0] - g->xmin) / dx), iy = (size_t)((mid[1] - g->ymin) / dy), iz = (size_t)((mid[2] - g->zmin) / dz);
    for (size_t ii = 0; ii < 2; ++ii)
        for (size_t jj = 0; jj < 2; ++jj)
            for (size_t kk = 0; kk < 2; ++kk) {
                const size_t ivx = ix + ii, jv = iy + jj, kv = iz + kk;
                /* the 1. literals make the three factors and their product double; dvdv is a T1 */
                const REAL dvdv = (REAL)((1. - FABS(mid[0] - ivx * dx) / dx) * (1. - FABS(mid[1] - jv * dy) / dy) *
                                         (1. - FABS(mid[2] - kv * dz) / dz));
                const long long j = (long long)((kv * g->nny + jv) * g->nnx + ivx);
                const REAL v = -s * ds * dvdv;
                long q;
                const long have = *nm < cap ? *nm : cap;
                for (q = 0; q < have; ++q)
                    if (mj[q] == j) { mv[q] += v; break; }
                if (q == have) {
                    if (*nm < cap) { mj[*nm] = j; mv[*nm] = v; }
                    ++*nm;
                }
            }
}

/* Grid3Drn::getRaypath(Tx, t0, Rx, m_data, RxNo, tt, threadNo), ttcr/Grid3Drn.h:1503-1800: the walk of getRaypath with the
 * terms of M instead of the points.  Restated as it stands: prev_pt is overwritten with curr_pt BEFORE the segment's mid-
 * point and length are formed (:1590-1597), so every step of the walk contributes -s^2 * 0 * w = a signed zero at the eight
 * nodes around the step's END point, and only the last hop (or two) to the source carries weight; a receiver on a source
 * point returns tt = 0 (not t0) and no entries.  Returns like fsm_raypath3d (3: capacity exceeded, *nm still counts). */
int SFX(fsm_raypath3d_m)(const SFX(fsm_grid3d) * g, const REAL* sn, const REAL* T, int n_src, const REAL* src,
                         const REAL* t0, const REAL rx[3], int iv, long max_steps, REAL* tt_out, long long* mj, REAL* mv, long cap,
                         long* nm_out) {
    REAL tt = 0.0, s1, s2;
    long nm = 0;
    *tt_out = tt;
    *nm_out = 0;
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[3 * ns] && rx[1] == src[3 * ns + 1] && rx[2] == src[3 * ns + 2]) return 0;
    REAL prev[3] = {rx[0], rx[1], rx[2]}, cur[3] = {rx[0], rx[1], rx[2]}, gv[3], mid[3];
    s1 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
    const REAL dx = g->dx;
    const REAL maxDist = (REAL)sqrt(dx * dx + dx * dx + dx * dx);
    int reached = 0;
    long steps = 0;
#define FSM_MID(A, B) do { mid[0] = (REAL)0.5 * ((A)[0] + (B)[0]); mid[1] = (REAL)0.5 * ((A)[1] + (B)[1]); mid[2] = (REAL)0.5 * ((A)[2] + (B)[2]); } while (0)
    while (!reached) {
        if (++steps > max_steps) { *nm_out = nm; return 2; }
        SFX(grad3d)(g, T, cur[0], cur[1], cur[2], &gv[0], &gv[1], &gv[2]);
        gv[0] *= (REAL)-1.0; gv[1] *= (REAL)-1.0; gv[2] *= (REAL)-1.0;
        SFX(step_to_plane)(g, cur, gv);
        if (cur[0] < g->xmin || cur[0] > g->xmax || cur[1] < g->ymin || cur[1] > g->ymax || cur[2] < g->zmin ||
            cur[2] > g->zmax) { *nm_out = nm; return 1; }
        s2 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
        tt += 0.5 * (s1 + s2) * SFX(dist3)(prev, cur);
        s1 = s2;
        prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
        FSM_MID(cur, prev);                                  /* = cur: prev was just overwritten */
        SFX(m_terms3d)(g, sn, mid, SFX(dist3)(cur, prev), iv, mj, mv, cap, &nm);
        for (int ns = 0; ns < n_src; ++ns) {
            const REAL* tx = src + 3 * ns;
            REAL dist = SFX(dist3)(cur, tx);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1]; gv[2] = tx[2] - cur[2];
                SFX(step_to_plane)(g, cur, gv);
                if (SFX(dist3)(cur, prev) > dist || (cur[0] == tx[0] && cur[1] == tx[1] && cur[2] == tx[2])) {
                    s2 = SFX(slowness_at3d)(g, sn, tx[0], tx[1], tx[2], iv);
                    tt += t0[ns] + 0.5 * (s1 + s2) * SFX(dist3)(prev, tx);
                    FSM_MID(tx, prev);
                    SFX(m_terms3d)(g, sn, mid, SFX(dist3)(tx, prev), iv, mj, mv, cap, &nm);
                } else {
                    s2 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
                    tt += 0.5 * (s1 + s2) * SFX(dist3)(prev, cur);
                    s1 = s2;
                    FSM_MID(cur, prev);
                    SFX(m_terms3d)(g, sn, mid, SFX(dist3)(cur, prev), iv, mj, mv, cap, &nm);
                    s2 = SFX(slowness_at3d)(g, sn, tx[0], tx[1], tx[2], iv);
                    tt += t0[ns] + 0.5 * (s1 + s2) * SFX(dist3)(cur, tx);
                    FSM_MID(tx, cur);
                    SFX(m_terms3d)(g, sn, mid, SFX(dist3)(tx, cur), iv, mj, mv, cap, &nm);
                }
                reached = 1;
            }
        }
    }
#undef FSM_MID
    *tt_out = tt;
    *nm_out = nm;
    return nm > cap ? 3 : 0;
}

/* Grid3Drn::getRaypath(Tx, t0, Rx, r_data, m_data, RxNo, tt, threadNo), ttcr/Grid3Drn.h:2144-2470 -- the overload ttcrpy calls
 * for compute_M WITH return_rays (src/ttcrpy/rgrid.pyx:1050): the walk and the points of the r_data overload, and for every
 * pushed point the terms of M of the segment from the point pushed before it -- here prev_pt is taken BEFORE the push
 * (:2228-2236), so the segments carry their length (unlike the m_data-only overload above).  One exception, as the reference
 * has it: for the plane point pushed between the walk and Tx prev_pt is read AFTER the push (:2359-2366), a zero-length
 * segment at that point.  A receiver on a source point: the point, tt = 0 (not t0), no terms.
 * Returns like fsm_raypath3d (3: a capacity exceeded; *npts and *nm still count). */
int SFX(fsm_raypath3d_rm)(const SFX(fsm_grid3d) * g, const REAL* sn, const REAL* T, int n_src, const REAL* src,
                          const REAL* t0, const REAL rx[3], int iv, long max_steps, REAL* tt_out, REAL* pts, long cap_pts,
                          long* npts, long long* mj, REAL* mv, long cap, long* nm_out) {
    REAL tt = 0.0, s1, s2;
    long nm = 0, np = 0;
    REAL back[3], prev[3], mid[3];
#define FSM_PUSH(P) do { if (np < cap_pts) { pts[3 * np] = (P)[0]; pts[3 * np + 1] = (P)[1]; pts[3 * np + 2] = (P)[2]; } \
                         back[0] = (P)[0]; back[1] = (P)[1]; back[2] = (P)[2]; ++np; } while (0)
#define FSM_MID(A, B) do { mid[0] = (REAL)0.5 * ((A)[0] + (B)[0]); mid[1] = (REAL)0.5 * ((A)[1] + (B)[1]); mid[2] = (REAL)0.5 * ((A)[2] + (B)[2]); } while (0)
    *tt_out = tt;
    *nm_out = 0;
    FSM_PUSH(rx);
    *npts = np;
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[3 * ns] && rx[1] == src[3 * ns + 1] && rx[2] == src[3 * ns + 2]) return 0;
    REAL cur[3] = {rx[0], rx[1], rx[2]}, gv[3];
    s1 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
    const REAL dx = g->dx;
    const REAL maxDist = (REAL)sqrt(dx * dx + dx * dx + dx * dx);
    int reached = 0;
    long steps = 0;
    while (!reached) {
        if (++steps > max_steps) { *nm_out = nm; *npts = np; return 2; }
        SFX(grad3d)(g, T, cur[0], cur[1], cur[2], &gv[0], &gv[1], &gv[2]);
        gv[0] *= (REAL)-1.0; gv[1] *= (REAL)-1.0; gv[2] *= (REAL)-1.0;
        SFX(step_to_plane)(g, cur, gv);
        if (cur[0] < g->xmin || cur[0] > g->xmax || cur[1] < g->ymin || cur[1] > g->ymax || cur[2] < g->zmin ||
            cur[2] > g->zmax) { *nm_out = nm; *npts = np; return 1; }
        prev[0] = back[0]; prev[1] = back[1]; prev[2] = back[2];
        s2 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
        tt += 0.5 * (s1 + s2) * SFX(dist3)(back, cur);
        s1 = s2;
        FSM_PUSH(cur);
        FSM_MID(cur, prev);
        SFX(m_terms3d)(g, sn, mid, SFX(dist3)(cur, prev), iv, mj, mv, cap, &nm);
        for (int ns = 0; ns < n_src; ++ns) {
            const REAL* tx = src + 3 * ns;
            REAL dist = SFX(dist3)(cur, tx);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1]; gv[2] = tx[2] - cur[2];
                SFX(step_to_plane)(g, cur, gv);
                if (SFX(dist3)(cur, back) > dist || (cur[0] == tx[0] && cur[1] == tx[1] && cur[2] == tx[2])) {
                    prev[0] = back[0]; prev[1] = back[1]; prev[2] = back[2];
                    s2 = SFX(slowness_at3d)(g, sn, tx[0], tx[1], tx[2], iv);
                    tt += t0[ns] + 0.5 * (s1 + s2) * SFX(dist3)(back, tx);
                    FSM_PUSH(tx);
                    FSM_MID(tx, prev);
                    SFX(m_terms3d)(g, sn, mid, SFX(dist3)(tx, prev), iv, mj, mv, cap, &nm);
                } else {
                    s2 = SFX(slowness_at3d)(g, sn, cur[0], cur[1], cur[2], iv);
                    tt += 0.5 * (s1 + s2) * SFX(dist3)(back, cur);
                    FSM_PUSH(cur);
                    s1 = s2;
                    prev[0] = back[0]; prev[1] = back[1]; prev[2] = back[2];   /* = cur: read after the push */
                    FSM_MID(cur, prev);
                    SFX(m_terms3d)(g, sn, mid, SFX(dist3)(cur, prev), iv, mj, mv, cap, &nm);
                    prev[0] = back[0]; prev[1] = back[1]; prev[2] = back[2];
                    s2 = SFX(slowness_at3d)(g, sn, tx[0], tx[1], tx[2], iv);
                    tt += t0[ns] + 0.5 * (s1 + s2) * SFX(dist3)(cur, tx);
                    FSM_PUSH(tx);
                    FSM_MID(tx, prev);
                    SFX(m_terms3d)(g, sn, mid, SFX(dist3)(tx, prev), iv, mj, mv, cap, &nm);
                }
                reached = 1;
            }
        }
    }
#undef FSM_MID
#undef FSM_PUSH
    *tt_out = tt;
    *nm_out = nm;
    *npts = np;
    return (nm > cap || np > cap_pts) ? 3 : 0;
}

/* ------------------------------------------------------------------ 2D -- */
/* 2D node index is z-fastest: n = i*(ncz+1)+j (ttcr/Grid2Drn.h:720). */

void SFX(fsm_grid2d_init)(SFX(fsm_grid2d) * g, uint32_t ncx, uint32_t ncz, REAL dx, REAL dz,
                          REAL xmin, REAL zmin) {
    /* Grid2Drn ctor, ttcr/Grid2Drn.h:58-66 */
    g->nnx = (size_t)ncx + 1;
    g->nnz = (size_t)ncz + 1;
    g->dx = dx; g->dz = dz;
    g->xmin = xmin; g->zmin = zmin;
    g->xmax = xmin + ncx * dx;
    g->zmax = zmin + ncz * dz;
}

int SFX(fsm_outside2d)(const SFX(fsm_grid2d) * g, int n, const REAL* p) {
    for (int m = 0; m < n; ++m)
        if (p[2 * m] < g->xmin || p[2 * m] > g->xmax || p[2 * m + 1] < g->zmin || p[2 * m + 1] > g->zmax)
            return 1;
    return 0;
}

/* neighbour minima shared by the 2D updates (ttcr/Grid2Drn.h:923-943) */
static void SFX(ab2d)(const REAL* T, size_t nnx, size_t nnz, size_t i, size_t j, REAL* pa, REAL* pb) {
    const size_t ncx = nnx - 1, ncz = nnz - 1;
    REAL a, b, t;
    if (i == 0)
        a = T[(i + 1) * nnz + j];
    else if (i == ncx)
        a = T[(i - 1) * nnz + j];
    else {
        a = T[(i - 1) * nnz + j];
        t = T[(i + 1) * nnz + j];
        a = a < t ? a : t;
    }
    if (j == 0)
        b = T[i * nnz + j + 1];
    else if (j == ncz)
        b = T[i * nnz + j - 1];
    else {
        b = T[i * nnz + j - 1];
        t = T[i * nnz + j + 1];
        b = b < t ? b : t;
    }
    *pa = a;
    *pb = b;
}

/* Grid2Drn::update_node, ttcr/Grid2Drn.h:920-954 (square cells) */
static void SFX(update_node2d)(REAL* T, const REAL* s, REAL dx, size_t nnx, size_t nnz, size_t i, size_t j) {
    REAL a, b, t;
    SFX(ab2d)(T, nnx, nnz, i, j, &a, &b);
    REAL fh = s[i * nnz + j] * dx;
    if (FABS(a - b) >= fh)
        t = (a < b ? a : b) + fh;
    else
        t = 0.5 * (a + b + sqrt(2. * fh * fh - (a - b) * (a - b)));
    if (t < T[i * nnz + j]) T[i * nnz + j] = t;
}

/* Grid2Drn::update_node_xz, ttcr/Grid2Drn.h:1019-1058 (dx != dz) */
static void SFX(update_node2d_xz)(REAL* T, const REAL* s, REAL dx, REAL dz, size_t nnx, size_t nnz,
                                  size_t i, size_t j) {
    REAL a, b, t;
    SFX(ab2d)(T, nnx, nnz, i, j, &a, &b);
    const REAL sn = s[i * nnz + j];
    if (a < b && ((b - a) / dx) > sn) {
        t = a + sn * dx;
    } else if (a > b && ((a - b) / dz) > sn) {
        t = b + sn * dz;
    } else {
        REAL dx2 = dx * dx;
        REAL dz2 = dz * dz;
        REAL s2 = sn * sn;
        t = (b * dx2 + a * dz2) / (dx2 + dz2) +
            sqrt((2.0 * a * b * dx2 * dz2 - a * a * dx2 * dz2 - b * b * dx2 * dz2 + dx2 * dx2 * dz2 * s2 +
                  dx2 * dz2 * dz2 * s2) /
                 ((dx2 + dz2) * (dx2 + dz2)));
    }
    if (t < T[i * nnz + j]) T[i * nnz + j] = t;
}

/* Grid2Drn::sweep / sweep_xz, ttcr/Grid2Drn.h:713-752, :797-835:
 * directions (i+,j+), (i-,j+), (i-,j-), (i+,j-), j innermost. */
static void SFX(sweep2d)(REAL* T, const REAL* s, const unsigned char* frozen, REAL dx, REAL dz, int xz,
                         size_t nnx, size_t nnz) {
    static const int RI[4] = {0, 1, 1, 0};
    static const int RJ[4] = {0, 0, 1, 1};
    for (int dir = 0; dir < 4; ++dir) {
        for (size_t ii = 0; ii < nnx; ++ii) {
            const size_t i = RI[dir] ? nnx - 1 - ii : ii;
            for (size_t jj = 0; jj < nnz; ++jj) {
                const size_t j = RJ[dir] ? nnz - 1 - jj : jj;
                if (!frozen[i * nnz + j]) {
                    if (xz)
                        SFX(update_node2d_xz)(T, s, dx, dz, nnx, nnz, i, j);
                    else
                        SFX(update_node2d)(T, s, dx, nnx, nnz, i, j);
                }
            }
        }
    }
}


/* Grid2Drn::update_node_weno3 (ttcr/Grid2Drn.h:1061-1213, dx == dz) and update_node_weno3_xz
 * (:1217-1356): WENO axis values with the axis spacing, then the first-order local solver. */
static void SFX(update_node2d_weno)(REAL* T, const REAL* s, REAL dx, REAL dz, int xz, size_t nnx, size_t nnz,
                                    size_t i, size_t j) {
    const size_t n = i * nnz + j;
    REAL a = SFX(weno_axis)(T + n, (ptrdiff_t)nnz, i, nnx - 1, dx);
    REAL b = SFX(weno_axis)(T + n, 1, j, nnz - 1, xz ? dz : dx);
    REAL t;
    if (!xz) {
        REAL fh = s[n] * dx;
        if (FABS(a - b) >= fh)
            t = (a < b ? a : b) + fh;
        else
            t = 0.5 * (a + b + sqrt(2. * fh * fh - (a - b) * (a - b)));
    } else {
        const REAL sn = s[n];
        if (a < b && ((b - a) / dx) > sn) {
            t = a + sn * dx;
        } else if (a > b && ((a - b) / dz) > sn) {
            t = b + sn * dz;
        } else {
            REAL dx2 = dx * dx;
            REAL dz2 = dz * dz;
            REAL s2 = sn * sn;
            t = (b * dx2 + a * dz2) / (dx2 + dz2) +
                sqrt((2.0 * a * b * dx2 * dz2 - a * a * dx2 * dz2 - b * b * dx2 * dz2 + dx2 * dx2 * dz2 * s2 +
                      dx2 * dz2 * dz2 * s2) /
                     ((dx2 + dz2) * (dx2 + dz2)));
        }
    }
    if (t < T[n]) T[n] = t;
}

/* Grid2Drn::sweep_weno3 / sweep_weno3_xz, ttcr/Grid2Drn.h:838-917 */
static void SFX(sweep2d_weno)(REAL* T, const REAL* s, const unsigned char* frozen, REAL dx, REAL dz, int xz,
                              size_t nnx, size_t nnz) {
    static const int RI[4] = {0, 1, 1, 0};
    static const int RJ[4] = {0, 0, 1, 1};
    for (int dir = 0; dir < 4; ++dir) {
        for (size_t ii = 0; ii < nnx; ++ii) {
            const size_t i = RI[dir] ? nnx - 1 - ii : ii;
            for (size_t jj = 0; jj < nnz; ++jj) {
                const size_t j = RJ[dir] ? nnz - 1 - jj : jj;
                if (!frozen[i * nnz + j]) SFX(update_node2d_weno)(T, s, dx, dz, xz, nnx, nnz, i, j);
            }
        }
    }
}

/* Node2Dn::getDistance, ttcr/Node2Dn.h:134-136 */
static REAL SFX(dist2d)(REAL x, REAL z, REAL px, REAL pz) {
    return (REAL)sqrt((x - px) * (x - px) + (z - pz) * (z - pz));
}

/* Grid2Drn::initFSM, ttcr/Grid2Drn.h:1360-1418.  Differs from 3D: on-node
 * neighbours use the mean of the neighbour's and the source node's slowness
 * (:1384), the off-node branch skips nothing (:1403-1412), and getCellNo uses
 * `small`, not `small2` (:173-179). */
static void SFX(init2d)(const SFX(fsm_grid2d) * g, const REAL* s, REAL* T, unsigned char* frozen,
                        int n_src, const REAL* src, const REAL* t0, int npts) {
    const ptrdiff_t nnx = g->nnx, nnz = g->nnz, ncx = nnx - 1, ncz = nnz - 1;
    for (int n = 0; n < n_src; ++n) {
        const REAL px = src[2 * n], pz = src[2 * n + 1];
        ptrdiff_t fi = -1, fj = -1;
        for (ptrdiff_t i = 0; i < nnx && fi < 0; ++i)
            if (FABS(SFX(coord)(g->xmin, (uint32_t)i, g->dx) - px) < FSM_SMALL) fi = i;
        for (ptrdiff_t j = 0; j < nnz && fj < 0; ++j)
            if (FABS(SFX(coord)(g->zmin, (uint32_t)j, g->dz) - pz) < FSM_SMALL) fj = j;
        if (fi >= 0 && fj >= 0) {
            const ptrdiff_t i = fi, j = fj;
            const size_t nn = (size_t)(i * nnz + j);
            T[nn] = t0[n];
            frozen[nn] = 1;
            for (ptrdiff_t ii = i - npts; ii <= i + npts; ++ii) {
                if (ii < 0 || ii > ncx) continue;
                for (ptrdiff_t jj = j - npts; jj <= j + npts; ++jj) {
                    if (jj >= 0 && jj <= ncz && !(ii == i && jj == j)) {
                        const size_t m = (size_t)(ii * nnz + jj);
                        REAL d = SFX(dist2d)(SFX(coord)(g->xmin, (uint32_t)ii, g->dx),
                                             SFX(coord)(g->zmin, (uint32_t)jj, g->dz), px, pz);
                        REAL tt = t0[n] + d * 0.5 * (s[m] + s[nn]);
                        T[m] = tt;
                        frozen[m] = 1;
                    }
                }
            }
        } else {
            REAL x = g->xmax - px < FSM_SMALL ? (REAL)(g->xmax - .5 * g->dx) : px;
            REAL z = g->zmax - pz < FSM_SMALL ? (REAL)(g->zmax - .5 * g->dz) : pz;
            const uint32_t cnx = FSM_U32(FSM_SMALL + (x - g->xmin) / g->dx);
            const uint32_t cnz = FSM_U32(FSM_SMALL + (z - g->zmin) / g->dz);
            const ptrdiff_t cellNo = (ptrdiff_t)(uint32_t)(cnx * (uint32_t)ncz + cnz);
            const ptrdiff_t i = cellNo / ncz;
            const ptrdiff_t j = cellNo - i * ncz;
            for (ptrdiff_t ii = i - (npts - 1); ii <= i + npts; ++ii) {
                if (ii < 0 || ii > ncx) continue;
                for (ptrdiff_t jj = j - (npts - 1); jj <= j + npts; ++jj) {
                    if (jj >= 0 && jj <= ncz) {
                        const size_t m = (size_t)(ii * nnz + jj);
                        REAL d = SFX(dist2d)(SFX(coord)(g->xmin, (uint32_t)ii, g->dx),
                                             SFX(coord)(g->zmin, (uint32_t)jj, g->dz), px, pz);
                        REAL tt = t0[n] + d * s[m];
                        T[m] = tt;
                        frozen[m] = 1;
                    }
                }
            }
        }
    }
}

/* Stencil rotated by pi/4: Grid2Drn::update_node45 (ttcr/Grid2Drn.h:957-1015).  a = min over the
 * (+1,+1)/(-1,-1) diagonal, b = min over the (+1,-1)/(-1,+1) diagonal, neighbours outside the
 * grid count as max(); fh = sqrt(2) s dx (double literal: product formed in double, rounded once). */
static void SFX(update_node45)(REAL* T, const REAL* s, REAL dx, size_t nnx, size_t nnz, size_t i, size_t j) {
    const size_t ncx = nnx - 1, ncz = nnz - 1;
    REAL a, b, t;
    if (i == 0) {
        a = j != ncz ? T[(i + 1) * nnz + j + 1] : REAL_MAX;
    } else if (i == ncx) {
        a = j != 0 ? T[(i - 1) * nnz + j - 1] : REAL_MAX;
    } else {
        a = j != ncz ? T[(i + 1) * nnz + j + 1] : REAL_MAX;
        t = j != 0 ? T[(i - 1) * nnz + j - 1] : REAL_MAX;
        a = a < t ? a : t;
    }
    if (i == 0) {
        b = j != 0 ? T[(i + 1) * nnz + j - 1] : REAL_MAX;
    } else if (i == ncx) {
        b = j != ncz ? T[(i - 1) * nnz + j + 1] : REAL_MAX;
    } else {
        b = j != 0 ? T[(i + 1) * nnz + j - 1] : REAL_MAX;
        t = j != ncz ? T[(i - 1) * nnz + j + 1] : REAL_MAX;
        b = b < t ? b : t;
    }
    REAL fh = 1.414213562373095 * s[i * nnz + j] * dx;
    if (FABS(a - b) >= fh)
        t = (a < b ? a : b) + fh;
    else
        t = 0.5 * (a + b + sqrt(2. * fh * fh - (a - b) * (a - b)));
    if (t < T[i * nnz + j]) T[i * nnz + j] = t;
}

/* Grid2Drn::sweep45 (ttcr/Grid2Drn.h:756-794): the four orderings of sweep2d, rotated stencil */
static void SFX(sweep2d_45)(REAL* T, const REAL* s, const unsigned char* frozen, REAL dx, size_t nnx, size_t nnz) {
    static const int RI[4] = {0, 1, 1, 0};
    static const int RJ[4] = {0, 0, 1, 1};
    for (int dir = 0; dir < 4; ++dir) {
        for (size_t ii = 0; ii < nnx; ++ii) {
            const size_t i = RI[dir] ? nnx - 1 - ii : ii;
            for (size_t jj = 0; jj < nnz; ++jj) {
                const size_t j = RJ[dir] ? nnz - 1 - jj : jj;
                if (!frozen[i * nnz + j]) SFX(update_node45)(T, s, dx, nnx, nnz, i, j);
            }
        }
    }
}

/* Grid2Drcfs::setSlowness, ttcr/Grid2Drcfs.h:98-138 (cells z-fastest: c = i*ncz + j).
 * Summation order of the reference: (i,j) (i,j-1) (i-1,j) (i-1,j-1). */
void SFX(fsm_cells_to_nodes2d)(size_t ncx, size_t ncz, const REAL* sc, REAL* sn) {
    const size_t nnx = ncx + 1, nnz = ncz + 1;
    for (size_t i = 0; i < nnx; ++i)
        for (size_t j = 0; j < nnz; ++j) {
            size_t ci[2], cj[2];
            int ni = 0, nj = 0;
            if (i < ncx) ci[ni++] = i;
            if (i > 0) ci[ni++] = i - 1;
            if (j < ncz) cj[nj++] = j;
            if (j > 0) cj[nj++] = j - 1;
            REAL sum = 0;
            int first = 1;
            for (int a = 0; a < ni; ++a)
                for (int b = 0; b < nj; ++b) {
                    REAL v = sc[ci[a] * ncz + cj[b]];
                    if (first) { sum = v; first = 0; } else sum = sum + v;
                }
            const int cnt = ni * nj;
            sn[i * nnz + j] = cnt == 1 ? sum : (cnt == 2 ? (REAL)(0.5 * sum) : (REAL)(0.25 * sum));
        }
}

/* Grid2Drnfs::raytrace driver, ttcr/Grid2Drnfs.h:198-299.  `weno` carries two flags: bit 0 = weno3,
 * bit 1 = rotated_template (sweep45 after every sweep of the first-order solver when dx == dz and
 * weno3 is off, :277-286; ignored otherwise, like the reference does). */
int SFX(fsm_solve2d)(const SFX(fsm_grid2d) * g, const REAL* s, int n_src, const REAL* src,
                     const REAL* t0, REAL eps, int maxit, int weno_flags, REAL* T, REAL* change_hist, int* niterw_out) {
    const int weno = weno_flags & 1, rotated = (weno_flags >> 1) & 1;
    const size_t N = g->nnx * g->nnz;
    REAL epsilon = eps;
    epsilon *= (REAL)N;
    unsigned char* frozen = (unsigned char*)calloc(N, 1);
    REAL* times = (REAL*)malloc(N * sizeof(REAL));
    for (size_t n = 0; n < N; ++n) T[n] = REAL_MAX;
    SFX(init2d)(g, s, T, frozen, n_src, src, t0, weno ? 2 : 1);
    for (size_t n = 0; n < N; ++n) times[n] = T[n];
    REAL change = REAL_MAX;
    int niter = 0, niterw = 0;
    const int xz = !(g->dx == g->dz);
    while (change >= epsilon && niter < maxit) {
        SFX(sweep2d)(T, s, frozen, g->dx, g->dz, xz, g->nnx, g->nnz);
        if (rotated && !weno && !xz) SFX(sweep2d_45)(T, s, frozen, g->dx, g->nnx, g->nnz);
        change = 0.0;
        for (size_t n = 0; n < N; ++n) {
            REAL dt = FABS(times[n] - T[n]);
            change += dt;
            times[n] = T[n];
        }
        if (change_hist) change_hist[niter] = change;
        niter++;
    }
    if (weno) { /* ttcr/Grid2Drnfs.h:236-275 */
        change = REAL_MAX;
        while (change >= epsilon && niterw < maxit) {
            SFX(sweep2d_weno)(T, s, frozen, g->dx, g->dz, xz, g->nnx, g->nnz);
            change = 0.0;
            for (size_t n = 0; n < N; ++n) {
                REAL dt = FABS(times[n] - T[n]);
                change += dt;
                times[n] = T[n];
            }
            if (change_hist) change_hist[maxit + niterw] = change;
            niterw++;
        }
    }
    if (niterw_out) *niterw_out = niterw;
    free(times);
    free(frozen);
    return niter;
}

/* Grid2Drn::getIJ (ttcr/Grid2Drn.h:186-189) + getTraveltime (:359-414) */
REAL SFX(fsm_interp2d)(const SFX(fsm_grid2d) * g, const REAL* T, REAL px, REAL pz) {
    const size_t nnx = g->nnx, nnz = g->nnz;
    const REAL xmin = g->xmin, zmin = g->zmin, dx = g->dx, dz = g->dz;
    const uint32_t i = FSM_U32(FSM_SMALL + (px - xmin) / dx);
    const uint32_t j = FSM_U32(FSM_SMALL + (pz - zmin) / dz);
    const int onx = FABS(px - (xmin + i * dx)) < FSM_SMALL;
    const int onz = FABS(pz - (zmin + j * dz)) < FSM_SMALL;
/* index = quotient + 1e-4 (in cells), "on the line" = ABSOLUTE distance below 1e-4: with dx > 1 a point between
 * 1e-4 and 1e-4*dx below the last line of an axis gets the last node as lower index, is not "on" it, and the
 * reference reads index+1 (the next column, or past the array).  Clamped to the last node as in 3-D. */
#define T2_CL(v, n) ((size_t)(v) < (n) ? (size_t)(v) : (n) - 1)
#define T2(ii, jj) T[T2_CL(ii, nnx) * nnz + T2_CL(jj, nnz)]
    REAL tt;
    if (onx && onz) {
        return T2(i, j);
    } else if (onx) {
        REAL t1 = T2(i, j), t2 = T2(i, j + 1);
        REAL w1 = (zmin + (j + 1) * dz - pz) / dz, w2 = (pz - (zmin + j * dz)) / dz;
        tt = t1 * w1 + t2 * w2;
    } else if (onz) {
        REAL t1 = T2(i, j), t2 = T2(i + 1, j);
        REAL w1 = (xmin + (i + 1) * dx - px) / dx, w2 = (px - (xmin + i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    } else {
        REAL t1 = T2(i, j), t2 = T2(i + 1, j);
        REAL t3 = T2(i, j + 1), t4 = T2(i + 1, j + 1);
        REAL w1 = (xmin + (i + 1) * dx - px) / dx, w2 = (px - (xmin + i * dx)) / dx;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (zmin + (j + 1) * dz - pz) / dz;
        w2 = (pz - (zmin + j * dz)) / dz;
        tt = t1 * w1 + t2 * w2;
    }
#undef T2
#undef T2_CL
    return tt;
}

/* Grid2Drn::computeSlowness(pt), ttcr/Grid2Drn.h:262-330 (Grid2Drnfs and Grid2Drcfs both: interpolation of the NODE
 * slowness): on-line test = first node within small^2 (absolute), cell index = quotient + small; an index past the
 * last node is clamped like SN_CL in 3-D. */
REAL SFX(fsm_compute_slowness2d)(const SFX(fsm_grid2d) * g, const REAL* sn, REAL px, REAL pz) {
    const size_t nnx = g->nnx, nnz = g->nnz;
    const REAL xmin = g->xmin, zmin = g->zmin, dx = g->dx, dz = g->dz;
    ptrdiff_t onX = -1, onZ = -1;
    for (size_t n = 0; n < nnx; ++n)
        if (FABS(px - (xmin + n * dx)) < FSM_SMALL2) { onX = (ptrdiff_t)n; break; }
    for (size_t n = 0; n < nnz; ++n)
        if (FABS(pz - (zmin + n * dz)) < FSM_SMALL2) { onZ = (ptrdiff_t)n; break; }
#define S2_CL(v, n) ((size_t)(v) < (size_t)(n) ? (size_t)(v) : (size_t)(n) - 1)
#define S2(ii, kk) sn[S2_CL(ii, nnx) * nnz + S2_CL(kk, nnz)]
    REAL s[4], x[3], z[3];
    if (onX != -1 && onZ != -1) {
        return sn[(size_t)onX * nnz + onZ];
    } else if (onX != -1) {
        const uint32_t k = FSM_U32(FSM_SMALL + (pz - zmin) / dz);
        s[0] = S2(onX, k); s[1] = S2(onX, k + 1);
        x[0] = pz; x[1] = zmin + k * dz; x[2] = zmin + (k + 1) * dz;
        return SFX(lin1)(x, s);
    } else if (onZ != -1) {
        const uint32_t i = FSM_U32(FSM_SMALL + (px - xmin) / dx);
        s[0] = S2(i, onZ); s[1] = S2(i + 1, onZ);
        x[0] = px; x[1] = xmin + i * dx; x[2] = xmin + (i + 1) * dx;
        return SFX(lin1)(x, s);
    }
    const uint32_t i = FSM_U32(FSM_SMALL + (px - xmin) / dx);
    const uint32_t k = FSM_U32(FSM_SMALL + (pz - zmin) / dz);
    s[0] = S2(i, k); s[1] = S2(i, k + 1); s[2] = S2(i + 1, k); s[3] = S2(i + 1, k + 1);
    x[0] = px; z[0] = pz; x[1] = xmin + i * dx; z[1] = zmin + k * dz; x[2] = xmin + (i + 1) * dx; z[2] = zmin + (k + 1) * dz;
    return SFX(lin2)(x, z, s);
#undef S2
#undef S2_CL
}

/* ---- 2-D raypath family ----------------------------------------------------------------------
 * Grid2Drn::grad(g, pt, nt), ttcr/Grid2Drn.h:606-632: centred difference of interpolated traveltimes
 * over one cell, window clamped to the grid. */
static void SFX(grad2d)(const SFX(fsm_grid2d) * g, const REAL* T, REAL px, REAL pz, REAL* gx, REAL* gz) {
    const REAL dx = g->dx, dz = g->dz;
    REAL p1 = px - dx / 2.0;
    if (p1 < g->xmin) p1 = g->xmin;
    REAL p2 = p1 + dx;
    if (p2 > g->xmax) {
        p2 = g->xmax;
        p1 = g->xmax - dx;
    }
    *gx = (SFX(fsm_interp2d)(g, T, p2, pz) - SFX(fsm_interp2d)(g, T, p1, pz)) / dx;
    p1 = pz - dz / 2.0;
    if (p1 < g->zmin) p1 = g->zmin;
    p2 = p1 + dz;
    if (p2 > g->zmax) {
        p2 = g->zmax;
        p1 = g->zmax - dz;
    }
    *gz = (SFX(fsm_interp2d)(g, T, px, p2) - SFX(fsm_interp2d)(g, T, px, p1)) / dz;
}

/* Grid2Drn::getCellNo, ttcr/Grid2Drn.h:170-176 */
static uint32_t SFX(cellno2d)(const SFX(fsm_grid2d) * g, REAL px, REAL pz) {
    const REAL x = g->xmax - px < FSM_SMALL ? (REAL)(g->xmax - .5 * g->dx) : px;
    const REAL z = g->zmax - pz < FSM_SMALL ? (REAL)(g->zmax - .5 * g->dz) : pz;
    uint32_t nx = FSM_U32(FSM_SMALL + (x - g->xmin) / g->dx);
    uint32_t nz = FSM_U32(FSM_SMALL + (z - g->zmin) / g->dz);
    /* (xmax - px < small is an absolute test, the index a relative one: with dx > 1 a point between 1e-4 and
     * 1e-4*dx below xmax gets the cell past the last one and the reference reads past its cell array: clamped) */
    if (nx > (uint32_t)(g->nnx - 2)) nx = (uint32_t)(g->nnx - 2);
    if (nz > (uint32_t)(g->nnz - 2)) nz = (uint32_t)(g->nnz - 2);
    return nx * (uint32_t)(g->nnz - 1) + nz;
}

/* "advance curr along gv to the next grid line of cell (i,k)" (ttcr/Grid2Drn.h:1510-1533 and twins);
 * (i,k) is the cell index taken at the START of the walk step -- the reference does not refresh it */
static void SFX(step2d)(const SFX(fsm_grid2d) * g, ptrdiff_t i, ptrdiff_t k, REAL cur[2], const REAL gv[2]) {
    const REAL dx = g->dx, dz = g->dz;
    REAL xp = g->xmin + dx * (i + (SFX(sgn)(gv[0]) > 0.0 ? 1.0 : 0.0));
    REAL zp = g->zmin + dz * (k + (SFX(sgn)(gv[1]) > 0.0 ? 1.0 : 0.0));
    if (FABS(xp - cur[0]) < FSM_SMALL) xp += dx * SFX(sgn)(gv[0]);
    if (FABS(zp - cur[1]) < FSM_SMALL) zp += dz * SFX(sgn)(gv[1]);
    REAL tx = gv[0] != 0.0 ? (xp - cur[0]) / gv[0] : REAL_MAX;
    REAL tz = gv[1] != 0.0 ? (zp - cur[1]) / gv[1] : REAL_MAX;
    if (tx < tz) {
        cur[0] += tx * gv[0]; cur[1] += tx * gv[1];
        cur[0] = xp;
    } else {
        cur[0] += tz * gv[0]; cur[1] += tz * gv[1];
        cur[1] = zp;
    }
}

/* record == 0: Grid2Drn::getTraveltimeFromRaypath (ttcr/Grid2Drn.h:1478-1661)
 * record == 1: Grid2Drn::getRaypath(Tx, t0, Rx, r_data, tt, threadNo) (:1663-1850): the same walk with every
 *              point recorded, `back` being r_data.back() (prev_pt in the other one).
 * sn: node slowness; sc: cell slowness of a Grid2Drcfs (hasCellSlowness, ttcr/Grid2Drcfs.h:62-68) or NULL.
 * Returns 0 ok, 1 ray left the grid twice (the reference throws), 2 step limit, 3 capacity exceeded. */
int SFX(fsm_raypath2d)(const SFX(fsm_grid2d) * g, const REAL* sn, const REAL* sc, const REAL* T, int n_src,
                       const REAL* src, const REAL* t0, const REAL rx[2], int record, long max_steps, REAL* tt_out,
                       REAL* pts, long cap, long* npts) {
    REAL tt = 0.0;
    REAL s1 = 0.0, s2 = 0.0, slown = 0.0;
    long np = 0;
    int over = 0;
    REAL back[2] = {rx[0], rx[1]}, cur[2] = {rx[0], rx[1]}, gv[2];
#define FSM_PUSH2(P) do { if (record) { if (np < cap) { pts[2 * np] = (P)[0]; pts[2 * np + 1] = (P)[1]; } else over = 1; \
                                         back[0] = (P)[0]; back[1] = (P)[1]; ++np; } } while (0)
    FSM_PUSH2(rx);
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[2 * ns] && rx[1] == src[2 * ns + 1]) { *tt_out = t0[ns]; if (npts) *npts = np; return 0; }
    if (!sc) s1 = SFX(fsm_interp2d)(g, sn, cur[0], cur[1]);   /* getSlowness has the shape of getTraveltime */
    const REAL dx = g->dx, dz = g->dz;
    const REAL maxDist = (REAL)sqrt(dx * dx + dz * dz);
    int reached = 0;
    long steps = 0;
    while (!reached) {
        if (++steps > max_steps) { if (npts) *npts = np; return 2; }
        SFX(grad2d)(g, T, cur[0], cur[1], &gv[0], &gv[1]);
        gv[0] *= (REAL)-1.0; gv[1] *= (REAL)-1.0;
        const ptrdiff_t i = (ptrdiff_t)(FSM_SMALL + (cur[0] - g->xmin) / dx);
        const ptrdiff_t k = (ptrdiff_t)(FSM_SMALL + (cur[1] - g->zmin) / dz);
        SFX(step2d)(g, i, k, cur, gv);
        if (cur[0] < g->xmin || cur[0] > g->xmax || cur[1] < g->zmin || cur[1] > g->zmax) {
            /* going outside: slide along the face instead (:1536-1581) */
            /* the reference writes an unqualified abs() here, which binds to the C int overload:
             * both components are truncated to integers before the comparison (:1538) */
            if (abs((int)gv[0]) > abs((int)gv[1])) { gv[0] = (REAL)SFX(sgn)(gv[0]); gv[1] = 0.0; }
            else { gv[1] = (REAL)SFX(sgn)(gv[1]); gv[0] = 0.0; }
            cur[0] = back[0]; cur[1] = back[1];
            SFX(step2d)(g, i, k, cur, gv);
            if (cur[0] < g->xmin || cur[0] > g->xmax || cur[1] < g->zmin || cur[1] > g->zmax) { if (npts) *npts = np; return 1; }
        }
        if (sc) {
            const REAL mx = (REAL)0.5 * (back[0] + cur[0]), mz = (REAL)0.5 * (back[1] + cur[1]);
            slown = sc[SFX(cellno2d)(g, mx, mz)];
        } else {
            s2 = SFX(fsm_interp2d)(g, sn, cur[0], cur[1]);
            slown = 0.5 * (s1 + s2);
            s1 = s2;
        }
        tt += slown * SFX(dist2d)(back[0], back[1], cur[0], cur[1]);
        if (record) FSM_PUSH2(cur); else { back[0] = cur[0]; back[1] = cur[1]; }
        for (int ns = 0; ns < n_src; ++ns) {
            const REAL* tx = src + 2 * ns;
            const REAL dist = SFX(dist2d)(cur[0], cur[1], tx[0], tx[1]);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1];
                SFX(step2d)(g, i, k, cur, gv);
                if (SFX(dist2d)(cur[0], cur[1], back[0], back[1]) > dist || (cur[0] == tx[0] && cur[1] == tx[1])) {
                    if (sc) {
                        const REAL mx = (REAL)0.5 * (back[0] + tx[0]), mz = (REAL)0.5 * (back[1] + tx[1]);
                        slown = sc[SFX(cellno2d)(g, mx, mz)];
                    } else {
                        s2 = SFX(fsm_interp2d)(g, sn, tx[0], tx[1]);
                        slown = 0.5 * (s1 + s2);
                    }
                    tt += slown * SFX(dist2d)(back[0], back[1], tx[0], tx[1]);
                    FSM_PUSH2(tx);
                } else {
                    if (sc) {
                        REAL mx = (REAL)0.5 * (back[0] + cur[0]), mz = (REAL)0.5 * (back[1] + cur[1]);
                        slown = sc[SFX(cellno2d)(g, mx, mz)];
                        tt += slown * SFX(dist2d)(back[0], back[1], cur[0], cur[1]);
                        mx = (REAL)0.5 * (cur[0] + tx[0]); mz = (REAL)0.5 * (cur[1] + tx[1]);
                        slown = sc[SFX(cellno2d)(g, mx, mz)];
                        tt += slown * SFX(dist2d)(cur[0], cur[1], tx[0], tx[1]);
                        /* (the reference records neither point in this branch, :1826-1831) */
                    } else {
                        s2 = SFX(fsm_interp2d)(g, sn, cur[0], cur[1]);
                        tt += 0.5 * (s1 + s2) * SFX(dist2d)(back[0], back[1], cur[0], cur[1]);
                        FSM_PUSH2(cur);
                        s1 = s2;
                        s2 = SFX(fsm_interp2d)(g, sn, tx[0], tx[1]);
                        tt += 0.5 * (s1 + s2) * SFX(dist2d)(cur[0], cur[1], tx[0], tx[1]);
                        FSM_PUSH2(tx);
                    }
                }
                tt += t0[ns];
                reached = 1;
            }
        }
    }
#undef FSM_PUSH2
    *tt_out = tt;
    if (npts) *npts = np;
    return over ? 3 : 0;
}

/* Grid2Drn::getRaypath with l_data (the ray-projection matrix L of compute_L):
 *   with_rays != 0: getRaypath(Tx, t0, Rx, r_data, l_data, tt, threadNo), ttcr/Grid2Drn.h:1852-2021
 *   with_rays == 0: getRaypath(Tx, t0, Rx, l_data, tt, threadNo), :2023-2190
 * A walk of its own: a step that leaves the grid ends it (the reference throws; no second try along the face as in
 * fsm_raypath2d); every segment is booked as (cell of its mid-point, length) in push order -- UNSORTED here, Grid2D::raytrace sorts
 * afterwards (ttcr/Grid2D.h:608-611).  When the last two segments lie in one cell the reference pushes the first entry and then
 * one holding the sum (:1985-1992, :2163-2170); without r_data the last hop's traveltime is slowness x that sum (:2181).
 * Returns 0 ok, 1 outside the grid, 2 step limit, 3 capacity. */
int SFX(fsm_raypath2d_l)(const SFX(fsm_grid2d) * g, const REAL* sn, const REAL* sc, const REAL* T, int n_src, const REAL* src,
                         const REAL* t0, const REAL rx[2], int with_rays, long max_steps, REAL* tt_out, REAL* pts, long cap,
                         long* npts, long long* lcell, REAL* lval, long lcap, long* nlen) {
    REAL tt = 0.0, s1 = 0.0, s2 = 0.0, slown = 0.0;
    long np = 0, nl = 0;
    int over = 0;
    REAL back[2] = {rx[0], rx[1]}, cur[2] = {rx[0], rx[1]}, gv[2];
#define FSM_PUSHL(P) do { if (with_rays) { if (np < cap) { pts[2 * np] = (P)[0]; pts[2 * np + 1] = (P)[1]; } else over = 1; ++np; } \
                          back[0] = (P)[0]; back[1] = (P)[1]; } while (0)
#define FSM_BOOK(Cc, V) do { if (nl < lcap) { lcell[nl] = (long long)(Cc); lval[nl] = (V); } else over = 1; ++nl; } while (0)
#define FSM_DONE(RC) do { *tt_out = tt; if (npts) *npts = np; if (nlen) *nlen = nl; return (RC); } while (0)
    if (with_rays) { if (np < cap) { pts[0] = rx[0]; pts[1] = rx[1]; } else over = 1; ++np; }
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[2 * ns] && rx[1] == src[2 * ns + 1]) { tt = t0[ns]; FSM_DONE(0); }
    if (!sc) s1 = SFX(fsm_interp2d)(g, sn, cur[0], cur[1]);
    const REAL dx = g->dx, dz = g->dz;
    const REAL maxDist = (REAL)sqrt(dx * dx + dz * dz);
    int reached = 0;
    long steps = 0;
    while (!reached) {
        if (++steps > max_steps) FSM_DONE(2);
        SFX(grad2d)(g, T, cur[0], cur[1], &gv[0], &gv[1]);
        gv[0] *= (REAL)-1.0; gv[1] *= (REAL)-1.0;
        const ptrdiff_t i = (ptrdiff_t)(FSM_SMALL + (cur[0] - g->xmin) / dx);
        const ptrdiff_t k = (ptrdiff_t)(FSM_SMALL + (cur[1] - g->zmin) / dz);
        SFX(step2d)(g, i, k, cur, gv);
        if (cur[0] < g->xmin || cur[0] > g->xmax || cur[1] < g->zmin || cur[1] > g->zmax) FSM_DONE(1);
        {
            const REAL mx = (REAL)0.5 * (back[0] + cur[0]), mz = (REAL)0.5 * (back[1] + cur[1]);
            const uint32_t c = SFX(cellno2d)(g, mx, mz);
            const REAL v = SFX(dist2d)(cur[0], cur[1], back[0], back[1]);
            FSM_BOOK(c, v);
            if (sc) slown = sc[c];
            else { s2 = SFX(fsm_interp2d)(g, sn, cur[0], cur[1]); slown = 0.5 * (s1 + s2); s1 = s2; }
            tt += slown * v;
            FSM_PUSHL(cur);
        }
        for (int ns = 0; ns < n_src; ++ns) {
            const REAL* tx = src + 2 * ns;
            const REAL dist = SFX(dist2d)(cur[0], cur[1], tx[0], tx[1]);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1];
                SFX(step2d)(g, i, k, cur, gv);
                if (SFX(dist2d)(cur[0], cur[1], back[0], back[1]) > dist || (cur[0] == tx[0] && cur[1] == tx[1])) {
                    const uint32_t c = SFX(cellno2d)(g, tx[0], tx[1]);
                    const REAL v = SFX(dist2d)(tx[0], tx[1], back[0], back[1]);
                    FSM_BOOK(c, v);
                    if (sc) slown = sc[c];
                    else { s2 = SFX(fsm_interp2d)(g, sn, tx[0], tx[1]); slown = 0.5 * (s1 + s2); }
                    tt += slown * v;
                    if (with_rays) FSM_PUSHL(tx);
                } else {
                    const REAL mx = (REAL)0.5 * (back[0] + cur[0]), mz = (REAL)0.5 * (back[1] + cur[1]);
                    uint32_t c = SFX(cellno2d)(g, mx, mz);
                    REAL v = SFX(dist2d)(cur[0], cur[1], back[0], back[1]);
                    FSM_BOOK(c, v);
                    if (sc) slown = sc[c];
                    else { s2 = SFX(fsm_interp2d)(g, sn, cur[0], cur[1]); slown = 0.5 * (s1 + s2); s1 = s2; }
                    tt += slown * v;
                    FSM_PUSHL(cur);
                    const uint32_t c2 = SFX(cellno2d)(g, tx[0], tx[1]);
                    const REAL hop = SFX(dist2d)(tx[0], tx[1], back[0], back[1]);
                    if (c == c2) v += hop; else { c = c2; v = hop; }
                    FSM_BOOK(c, v);
                    if (sc) slown = sc[c];
                    else { s2 = SFX(fsm_interp2d)(g, sn, tx[0], tx[1]); slown = 0.5 * (s1 + s2); }
                    tt += slown * (with_rays ? hop : v);
                    if (with_rays) FSM_PUSHL(tx);
                }
                tt += t0[ns];
                reached = 1;
            }
        }
    }
    FSM_DONE(over ? 3 : 0);
#undef FSM_PUSHL
#undef FSM_BOOK
#undef FSM_DONE
}

#undef FSM_SMALL
#undef FSM_SMALL2
