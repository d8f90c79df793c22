// ttcr_amd/csrc/fsm_kernels.h -- CDNA4 (gfx950) kernels of the fast-sweeping eikonal solver.
//
// What the reference does (ttcr/Grid3Drn.h:2816-2959, ttcr/Grid2Drn.h:713-954): in-place
// Gauss-Seidel sweeps in 8 (3-D) / 4 (2-D) lexicographic directions; each node solves a
// local first-order Godunov quadratic from the minima of its axis neighbours.
//
// How it is done here (not a port): a sweep in direction d is a partial order -- node
// (i',j',k') (indices oriented along d) needs its three upwind neighbours already updated
// and its three downwind neighbours not yet.  Any linear extension of that order gives the
// serial result bit for bit.  We use a *skewed pencil march*:
//
//   * level  L = i' + j' + k'   (i' is the memory-contiguous axis)
//   * a workgroup owns a PJ x PK patch of (j',k') columns; thread (tj,tk) owns one column and
//     at level L updates its node i' = L - j' - k'.  All threads of a level are independent,
//     so a patch advances one level per step with PJ*PK-way parallelism and no ramp.
//   * a launch processes, for every patch, one TILE of BL consecutive levels.  Patch
//     m = TJ+TK gets the level window [BL*w - m*(BL-1), +BL) in launch w: with that shift a
//     tile depends only on tiles of launch w-1 (own previous window and the two upwind
//     patches), and every tile it could race with is provably disjoint, so
//     all tiles of a launch run concurrently with no inter-workgroup communication.
//   * the tile (own columns + 1-column halo, BL+2 levels) is staged in LDS in skewed form:
//     row = column, entry q = level, so global loads/stores run along i (coalesced, rows are
//     contiguous in HBM) while the march reads one LDS column per level (conflict-free,
//     odd row stride).
//   * convergence: each thread accumulates the decrease of its nodes in fp64; a __shfl_down
//     wavefront reduction + one atomicAdd per wave gives sum|T_old-T_new| of the iteration
//     (updates only ever decrease T, so the per-sweep decreases telescope to the L1 change the
//     reference computes from a snapshot, ttcr/Grid3Drnfs.h:141-152) without a snapshot array.
//   * a tile that changed nothing skips its write-back.
//
// Arithmetic mirrors the reference exactly (see update3/update2): for float grids the
// quadratic branches are evaluated in double and rounded once, because the reference's double
// literals promote them.  Compile with -ffp-contract=off: fused a1+s*dx would change rounding;
// the fma() calls below are deliberate (the products are exact in double, so fma(x,y,acc) ==
// round(x*y+acc) == the reference's separately rounded add).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ttcr_amd {

// edge of the bricks (natural coordinates) whose last-change sweep number is tracked
constexpr int FSM_BRICK = 16;

template <typename T> struct real_traits;
template <> struct real_traits<float> {
    static __host__ __device__ constexpr float max() { return 3.402823466e+38f; }
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
};
template <> struct real_traits<double> {
    static __host__ __device__ constexpr double max() { return 1.7976931348623157e+308; }
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
};

// Index of a coordinate (ray walks, interpolation).  The reference converts a double to its unsigned index type; for a point
// OUTSIDE the grid -- the end game of a ray with several source points within a cell diagonal moves curr_pt without a bounds check
// (e.g. ttcr/Grid2Drn.h:1596-1655) -- that conversion of a negative value is undefined in C++ and the compiled reference reads far
// outside its arrays.  Here, as in the oracle (FSM_U32): negative -> 0, followed by the callers' upper clamps (the nearest cell /
// node); identical to the plain conversion for every point inside the grid.
__device__ __forceinline__ uint32_t idx_u32(double v) { return v < 0 ? 0u : (v >= 4294967295.0 ? 4294967295u : (uint32_t)v); }

__device__ __forceinline__ float rmin(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ double rmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ float rmax(float a, float b) { return a < b ? b : a; }
__device__ __forceinline__ double rmax(double a, double b) { return a < b ? b : a; }

// wave-uniform "any lane" tests.  A ballot of a general bool goes through a VALU select + compare;
// a compare builtin writes the lane mask straight into a scalar pair, and masks combine on the SALU.
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ unsigned long long lanes_gt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 2); }   // FCMP_OGT
__device__ __forceinline__ unsigned long long lanes_gt(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 2); }

// ---- 3-D local solver: Grid3Drn::update_node, ttcr/Grid3Drn.h:2936-2956 -------------------
// inputs: the three axis minima (any order), node slowness s, cell size dx. Returns candidate t.
// fp64 square root for the discriminants: the same Goldschmidt sequence the compiler emits for
// llvm.sqrt.f64 (v_rsq_f64 seed, two refinements, correctly rounded) WITHOUT the 2^+-256 range
// scaling, which only matters for inputs below 2^-767.  Our arguments are sums of exact products
// of floats: either >= ~1e-90, or exactly 0 (handled), or negative / NaN (-> NaN, like sqrt()).
__device__ __forceinline__ double sqrt_disc(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x == 0.0 ? x : g;
}
// the same for arguments that are positive wherever the result is used (NaN elsewhere is fine): no zero fix-up
__device__ __forceinline__ double sqrt_disc_pos(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// `live_lanes`: lane mask of the results that are used.  When no live lane of the wavefront leaves the 1-D branch
// (t1 <= a2: unreached regions, where every neighbour is still FLT_MAX, and grazing fronts) the
// fp64 work is skipped for the whole wave -- a wave-uniform branch, values unchanged.
__device__ __forceinline__ float update3(float ax, float ay, float az, float s, float dx, unsigned long long live_lanes) {
    // sort (std::swap network of :2936-2938; values only, so min/max/med3 is equivalent)
    const float a1 = __builtin_fminf(__builtin_fminf(ax, ay), az);
    const float a3 = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az);
    const float a2 = __builtin_amdgcn_fmed3f(ax, ay, az);
    const float fh = s * dx;
    const float t1 = a1 + fh;
    float t = t1;
    const unsigned long long beyond1d = lanes_gt(t1, a2) & live_lanes;
    if (beyond1d != 0ull) {
        const double d1 = a1, d2 = a2, d3 = a3, dfh = fh;
        // -2.*a1*a1 + 2.*a1*a2 - 2.*a2*a2 + 2.*a1*a3 + 2.*a2*a3 - 2.*a3*a3 + 3.*fh*fh, left to
        // right.  Every product is exact in double, so fma(x,y,acc) rounds exactly like the
        // reference's add; the common factor 2 of the first six terms is a power of two, so the
        // chain is run on the halved terms (same roundings, scaled) and doubled in the last fma.
        double r = -d1 * d1;
        r = __builtin_fma(d1, d2, r);
        r = __builtin_fma(-d2, d2, r);
        r = __builtin_fma(d1, d3, r);
        r = __builtin_fma(d2, d3, r);
        r = __builtin_fma(-d3, d3, r);
        r = __builtin_fma(r, 2.0, (3.0 * dfh) * dfh);
        const float s12 = a1 + a2;
        const float s123 = s12 + a3;
        // (no zero fix-up of the root: wherever t3 is taken the 2-D value lies above a3, i.e. fh^2 > u^2 + v^2 in the
        // notation below, and r = 3 fh^2 - (u-v)^2 - u^2 - v^2 > (u+v)^2 >= 0 then; r == 0 needs u = v = 0, where
        // r = 3 fh^2.  A lane with r <= 0 gets NaN and takes t1 or t2, as it would with sqrt(r).)
        const float t3 = (float)((1. / 3.) * ((double)s123 + sqrt_disc_pos(r)));
        // Is the 2-D quadratic t2 needed at all?  The reference takes the 3-D value iff t1 > a2 and
        // t2 > a3.  With u = a3-a1, v = a3-a2:  t2* > a3  <=>  fh^2 > u^2 + v^2 (exact arithmetic),
        // and t2* - a3 >= (fh^2-u^2-v^2)/(3.42 fh).  The computed t2 differs from t2* by less than
        // 2.5e-7 (|a1|+|a3|+fh) and the fp32 evaluation of fh^2-u^2-v^2 is off by < 5e-7 fh^2, so
        //     fh^2 - u^2 - v^2  >  4e-6 fh (|a1|+|a3|+fh)        ("clearly 3-D")
        // implies t1 > a2 and t2 > a3 with a > 2x margin (derivation: DESIGN.md section 4).  When
        // every live lane is either 1-D or clearly 3-D the wave skips t2 and its fp64 sqrt.
        const float u = a3 - a1, v = a3 - a2;
        const float slack = fh * fh - (u * u + v * v);
        const float thr = 4e-6f * fh * (__builtin_fabsf(a1) + __builtin_fabsf(a3) + fh);
        if ((beyond1d & ~lanes_gt(slack, thr)) != 0ull) {
            // 2.*fh*fh - (a1-a2)*(a1-a2): fh*fh exact in double, (a1-a2)^2 rounded in float
            const float df = a1 - a2;
            const float df2 = df * df;
            const double disc2 = __builtin_fma(dfh * dfh, 2.0, -(double)df2);
            // (disc2 == 0 needs |a1-a2| = sqrt(2) fh, where t1 <= a2 and t2 is dropped: no zero fix-up either)
            const float t2 = (float)(0.5 * ((double)s12 + sqrt_disc_pos(disc2)));
            t = t1 > a2 ? (t2 > a3 ? t3 : t2) : t1;
        } else {
            t = t1 > a2 ? t3 : t1;
        }
    }
    return t;
}

__device__ __forceinline__ double update3(double ax, double ay, double az, double s, double dx, unsigned long long live_lanes) {
    const double a1 = __builtin_fmin(__builtin_fmin(ax, ay), az);
    const double a3 = __builtin_fmax(__builtin_fmax(ax, ay), az);
    const double a2 = __builtin_fmax(__builtin_fmin(ax, ay), __builtin_fmin(__builtin_fmax(ax, ay), az));
    const double fh = s * dx;
    const double t1 = a1 + fh;
    double t = t1;
    if ((lanes_gt(t1, a2) & live_lanes) != 0ull) {
        const double t2 = 0.5 * (a1 + a2 + __builtin_sqrt(2. * fh * fh - (a1 - a2) * (a1 - a2)));
        const double t3 = 1. / 3. * ((a1 + a2 + a3) + __builtin_sqrt(-2. * a1 * a1 + 2. * a1 * a2 - 2. * a2 * a2 +
                                                                      2. * a1 * a3 + 2. * a2 * a3 -
                                                                      2. * a3 * a3 + 3. * fh * fh));
        t = t1 > a2 ? (t2 > a3 ? t3 : t2) : t1;
    }
    return t;
}

// ---- 2-D local solvers ---------------------------------------------------------------------
// Grid2Drn::update_node, ttcr/Grid2Drn.h:945-950 (square cells). a: x-axis minimum, b: z-axis.
__device__ __forceinline__ float update2_fh(float a, float b, float fh) {
    // both branches evaluated, one select: the one-wave 2-D march is a single dependent chain and a divergent branch
    // costs it two exec-mask round trips per level.  Where the 1-D value is taken (|d| >= fh) the discriminant may be
    // <= 0 and the quadratic value NaN -- it is dropped; where it is used, disc = 2 fh^2 - d^2 > fh^2 > 0.
    const float d = a - b;
    const float t1 = (a < b ? a : b) + fh;
    const float d2 = d * d;
    const double dfh = fh;
    const double disc = __builtin_fma(2.0 * dfh, dfh, -(double)d2);
    const float sab = a + b;
    const float t2 = (float)(0.5 * ((double)sab + sqrt_disc_pos(disc)));
    return __builtin_fabsf(d) >= fh ? t1 : t2;
}
__device__ __forceinline__ double update2_fh(double a, double b, double fh) {
    if (__builtin_fabs(a - b) >= fh) return (a < b ? a : b) + fh;
    return 0.5 * (a + b + __builtin_sqrt(2. * fh * fh - (a - b) * (a - b)));
}
__device__ __forceinline__ float update2(float a, float b, float s, float dx) { return update2_fh(a, b, s * dx); }
__device__ __forceinline__ double update2(double a, double b, double s, double dx) { return update2_fh(a, b, s * dx); }
// Grid2Drn::update_node45, ttcr/Grid2Drn.h:1005-1007: fh = sqrt(2) s dx with a double literal, so the
// float instantiation forms both products in double and rounds once
__device__ __forceinline__ float fh45(float s, float dx) { return (float)((1.414213562373095 * (double)s) * (double)dx); }
__device__ __forceinline__ double fh45(double s, double dx) { return (1.414213562373095 * s) * dx; }

// Grid2Drn::update_node_xz, ttcr/Grid2Drn.h:1041-1054 (dx != dz)
__device__ __forceinline__ float update2_xz(float a, float b, float sn, float dx, float dz) {
    if (a < b && ((b - a) / dx) > sn) return a + sn * dx;
    if (a > b && ((a - b) / dz) > sn) return b + sn * dz;
    const float dx2 = dx * dx, dz2 = dz * dz, s2 = sn * sn;
    // 2.0*a*b*dx2*dz2 is a double chain; the other four terms are float chains, promoted on use
    const double t1 = (((2.0 * (double)a) * (double)b) * (double)dx2) * (double)dz2;
    const float t2 = ((a * a) * dx2) * dz2;
    const float t3 = ((b * b) * dx2) * dz2;
    const float t4 = ((dx2 * dx2) * dz2) * s2;
    const float t5 = ((dx2 * dz2) * dz2) * s2;
    const double num = (((t1 - (double)t2) - (double)t3) + (double)t4) + (double)t5;
    const float den = (dx2 + dz2) * (dx2 + dz2);
    const float lin = (b * dx2 + a * dz2) / (dx2 + dz2);
    return (float)((double)lin + __builtin_sqrt(num / (double)den));
}
__device__ __forceinline__ double update2_xz(double a, double b, double sn, double dx, double dz) {
    if (a < b && ((b - a) / dx) > sn) return a + sn * dx;
    if (a > b && ((a - b) / dz) > sn) return b + sn * dz;
    const double dx2 = dx * dx, dz2 = dz * dz, s2 = sn * sn;
    return (b * dx2 + a * dz2) / (dx2 + dz2) +
           __builtin_sqrt((2.0 * a * b * dx2 * dz2 - a * a * dx2 * dz2 - b * b * dx2 * dz2 + dx2 * dx2 * dz2 * s2 +
                           dx2 * dz2 * dz2 * s2) /
                          ((dx2 + dz2) * (dx2 + dz2)));
}

// ---- tolerance-grade local solvers (option "arith" = 1; template AR = 1 of the persistent kernels) ---------------------------
// Same quadratics (ttcr/Grid3Drn.h:2936-2956, ttcr/Grid2Drn.h:945-950), evaluated in fp32 on DIFFERENCES from the smallest
// neighbour, scaled by 1/fh: with p2 = (a2-a1)/fh, p3 = (a3-a1)/fh
//     2-D:  t = a1 + fh/2 (p2 + sqrt(2 - p2^2))                        taken when p2 < 1 (the reference's t1 > a2)
//     3-D:  t = a1 + fh/3 (p2 + p3 + sqrt(3 - p2^2 - p3^2 - (p3-p2)^2))  taken when p3^2 + (p3-p2)^2 < 1, which in exact
//           arithmetic IS the reference's t2 > a3 (update3 above) -- decided before any root, so ONE v_sqrt_f32 per update
// Both discriminants are > 1 wherever their branch is taken (no cancellation), the two branches meet continuously at the
// switch (t2 = t3 = a3), and the scaling makes the result independent of the units of slowness and distance (fh^2 may underflow
// fp32).  The increment t - a1 is good to a few ulp OF ITSELF, the sum rounds once like the reference's final conversion: a
// result differs from the reference's by at most an ulp of t, and only where the increment's error straddles a rounding
// boundary.  v_rcp_f32 / v_sqrt_f32 are the 1-ulp hardware approximations; the ~480 cycles of the fp64 chain of a level
// become ~100.  NOT bit-identical to the reference: opt-in, within north_star's 1e-5 s RMS by orders of magnitude
// (tests/test_arith_mode_gpu.py).  fh == 0: p = NaN or inf, the 1-D value a1 + 0 is taken like in the reference.
__device__ __forceinline__ float update3_fast(float ax, float ay, float az, float s, float dx) {
    const float a1 = __builtin_fminf(__builtin_fminf(ax, ay), az);
    const float a3 = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az);
    const float a2 = __builtin_amdgcn_fmed3f(ax, ay, az);
    const float fh = s * dx;
    const float rfh = __builtin_amdgcn_rcpf(fh);
    const float p2 = (a2 - a1) * rfh, p3 = (a3 - a1) * rfh;
    const float e = p3 - p2;
    const float q = __builtin_fmaf(p3, p3, e * e);
    const float n2 = __builtin_fmaf(-p2, p2, 2.0f);
    const bool s3 = q < 1.0f;
    const float disc = s3 ? (n2 + 1.0f) - q : n2;
    const float root = __builtin_amdgcn_sqrtf(disc);
    const float psum = s3 ? p2 + p3 : p2;
    const float w = fh * (s3 ? (1.0f / 3.0f) : 0.5f);
    const float t = __builtin_fmaf(w, psum + root, a1);
    return p2 < 1.0f ? t : a1 + fh;
}
// 2-D (square cells): the reference's own sequence of fp32 operations (ttcr/Grid2Drn.h:945-950 as update2_fh above evaluates it) -- a - b,
// its square and a + b rounded in fp32 -- with the discriminant in fp32 (one fma; > fh^2 where the quadratic is taken) and a v_sqrt_f32.
// 0.5 ((a + b) + root) in fp32 IS the reference's double expression rounded to float (the fp32 sum of two floats rounds their exact
// sum, the halving is exact), so a result differs from the reference's only where the root's error (~1e-7 fh) moves that rounding: the
// scaled-difference form of the 3-D solver, which rounds differently at every update, drifts to 2e-5 s RMS over the 4096 nodes of a
// C5 field; this one stays at 3e-6 s.  (The same treatment of the 3-D solver -- its (1/3)(s123 + root) in fp64 like the reference --
// costs 9 % of the batch's time for an RMS of 0.6e-6 instead of 0.8e-6 s at 256^3: not taken.)
__device__ __forceinline__ float update2_fast(float a, float b, float s, float dx) {
    const float fh = s * dx;
    const float d = a - b;
    const float t1 = (a < b ? a : b) + fh;
    const float root = __builtin_amdgcn_sqrtf(__builtin_fmaf(2.0f * fh, fh, -(d * d)));
    const float t2 = 0.5f * ((a + b) + root);
    return __builtin_fabsf(d) >= fh ? t1 : t2;
}

// ---- sweep-tile kernel ---------------------------------------------------------------------
// Unified 3-D / 2-D geometry: "fast" axis F (memory stride 1), "mid" axis J (stride NF),
// "slow" axis K (stride NF*NJ).  3-D: F=x, J=y, K=z.  2-D (z-fastest): F=z, J=x, NK=1.
struct SweepGeom {
    int NF, NJ, NK;        // node counts
    int npj, npk;          // patches along J and K
    int M;                 // shear modulus: max(NF, NJ) rounded up to an even number
    int MP;                // rows of a k plane of a sheared copy: M + FSM_XPAD (the first FSM_XPAD rows once more behind the last one)
    int SR;                // sheared slowness copies: pitch of a PAIR of rows, ceil(NJ/16) blocks of 2 levels x 16 columns;
                           // 0: plain rows of NJ elements
    uint32_t n_nodes;      // NF*NJ*NK
};
constexpr int FSM_XPAD = 16;   // >= the longest chunk - 1: the rows of a chunk are read without a wrap

// Sheared slowness.  A thread marches along F but a wavefront is laid out along J, so at a given
// level the 64 lanes of a wave read nodes that are NF-1 elements apart in the natural layout.
// set_slowness therefore keeps, per direction family, a copy indexed by (k, x = (i'+j'+k) mod M, j') -- i', j' oriented like the
// family, k natural -- in which the nodes of one level and one k are contiguous in j': the per-level slowness
// load of a wave is a coalesced row read straight into registers (no LDS staging, no transposition).
// x is the LEVEL of the node modulo M (round 5; until then (i'+j') mod M): at one level every thread of a workgroup reads row
// x = L mod M of its k plane, so the row part of a load address is a scalar -- the sweep kernels compute it once per level on
// the scalar unit and a load is one instruction (uniform base + per-thread 32-bit offset), where each of the C loads of a chunk
// used to cost ~20 vector instructions of walking x with its wrap and its in-range test (issue_static).  Rows M .. M+FSM_XPAD-1
// repeat rows 0 .. FSM_XPAD-1: the C consecutive rows of a chunk never wrap.
// Element order: A[k][x / 2][j' / 16][x % 2][j' % 16] -- one 128-byte line holds TWO consecutive levels of the 16
// columns a patch row covers.  With plain rows A[k][x][j'] the 64 bytes a patch row needs per level were half of a
// line whose other half belongs to the neighbouring patch (another workgroup, another time): every line was fetched
// twice.  Now the second half is the same lanes' next level, a fraction of a microsecond later.
// A direction and its opposite traverse the same array backwards, so 4 (3-D) / 2 (2-D) copies.
// (3-D grids.  The 64-column rows of the one-wave 2-D patches fill whole lines as plain rows A[x][j'] and are 3 % slower
// with the paired form -- four half lines per load instead of two full ones --, so 2-D grids keep plain rows: SR == 0.)
__host__ __device__ __forceinline__ size_t shear_plane(const SweepGeom& g) {   // elements of a k plane
    return g.SR == 0 ? (size_t)g.MP * (size_t)g.NJ : (size_t)(g.MP >> 1) * (size_t)g.SR;
}
__host__ __device__ __forceinline__ size_t shear_index(const SweepGeom& g, int kx, int x, int jx) {
    if (g.SR == 0) return ((size_t)kx * (size_t)g.MP + (size_t)x) * (size_t)g.NJ + (size_t)jx;
    return ((size_t)kx * (size_t)(g.MP >> 1) + (size_t)(x >> 1)) * (size_t)g.SR + (size_t)((jx >> 4) * 32 + (x & 1) * 16 + (jx & 15));
}
template <typename T>
__global__ void fsm_shear_slowness(const T* __restrict__ s, T* __restrict__ out, SweepGeom g, int rf, int rj) {
    const size_t N = g.n_nodes;
    for (size_t n = blockIdx.x * (size_t)blockDim.x + threadIdx.x; n < N; n += (size_t)gridDim.x * blockDim.x) {
        const int i = n % g.NF, j = (n / g.NF) % g.NJ, k = n / ((size_t)g.NF * g.NJ);
        const int ip = rf ? g.NF - 1 - i : i, jp = rj ? g.NJ - 1 - j : j;
        const int x = (ip + jp + k) % g.M;
        for (int xr = x; xr < g.MP; xr += g.M) out[shear_index(g, k, xr, jp)] = s[n];   // (rows M and above: the first rows again)
    }
}

// The same copy written a line at a time: a workgroup produces 16 consecutive x of one 16-column block of one k plane,
// lanes ordered (x, j') like the copy itself, so that a wavefront stores whole 128-byte lines (fsm_shear_slowness above walks
// the natural array and scatters 4-byte stores over as many lines: the fabric counted 8x the payload in write requests,
// profiles/r02/traffic.json).  The loads walk 16 rows of the natural array along an anti-diagonal band -- a compact region
// that stays in the caches across the workgroup.  (Rows M and above: the copies of rows 0 ..)
template <typename T>
__global__ __launch_bounds__(256) void fsm_shear_slowness_lines(const T* __restrict__ s, T* __restrict__ out, SweepGeom g, int rf, int rj) {
    const int jj = threadIdx.x & 15, xx = threadIdx.x >> 4;
    const int xrow = blockIdx.x * 16 + xx, jp = blockIdx.y * 16 + jj, k = blockIdx.z;
    if (xrow >= g.MP || jp >= g.NJ) return;
    const int x = xrow % g.M;
    int ip = (x - jp - k) % g.M;
    ip = ip < 0 ? ip + g.M : ip;
    if (ip >= g.NF) return;   // (no node at this level in this column: the entry is read -- rows are loaded whole -- and never used)
    const int i = rf ? g.NF - 1 - ip : ip, j = rj ? g.NJ - 1 - jp : jp;
    out[shear_index(g, k, xrow, jp)] = s[((size_t)k * g.NJ + j) * g.NF + i];
}

template <typename T>
struct SweepArgs {
    T* tt;                    // [n_slots][n_nodes] traveltime fields
    const T* s_sheared;       // sheared node slowness of this direction's family
    const uint32_t* frozen;   // [n_slots][mask_words] frozen bit per node
    const int* bbox;          // [n_slots][6] natural-index bounding box of frozen nodes (lo/hi per F,J,K)
    double* change;           // [n_slots] L1 decrease accumulated over the iteration
    const int* slots;         // [batch] slot (ts == 1) or slot group (ts == 2) handled by blockIdx.z / ticket
    const int* lmask;         // [batch] ts == 2: bit l set = source l of the group is still being solved
    int ts;                   // traveltime fields are interleaved in groups of ts sources: T[group][node][ts]
    const uint32_t* tiles;    // (TJ | TK<<16) of the patches that have nodes in launch w (blockIdx.x)
    SweepGeom g;
    uint32_t mask_words;
    T dx, dz;                 // cell size along J-axis(x)/F-axis(z) in 2-D; dx only in 3-D
    int rf, rj, rk;           // 1: axis swept in decreasing index
    int rev;                  // 1: s_sheared is traversed backwards (opposite direction of its family)
    int w;                    // launch index within the sweep
    int variant;              // 0: 3-D, 1: 2-D square cells, 2: 2-D dx != dz
    unsigned long long* prof; // debug (TTCR_FSM_PROF=1): per-phase wall-clock sums, else nullptr
};

// debug phase timer: 100 MHz constant clock, thread 0 of every block accumulates phase sums
#ifndef FSM_ENABLE_PROF
#define FSM_ENABLE_PROF 0   // build with -DFSM_ENABLE_PROF=1 to get the per-phase timers (TTCR_FSM_PROF=1)
#endif
#define FSM_PROF_MARK(slot_)                                                      \
    if (FSM_ENABLE_PROF && a.prof && tid == 0) {                                  \
        const unsigned long long now_ = wall_clock64();                           \
        atomicAdd(a.prof + (slot_), now_ - prof_t);                               \
        prof_t = now_;                                                            \
    }

// v_min_f32 directly: llvm.minnum would first canonicalise both (loaded) operands for signalling
// NaNs, one extra VALU op each; the traveltime fields never hold NaNs.
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmin(double a, double b) { return __builtin_fmin(a, b); }

template <typename T, int PJ, int PK, int BL, bool IS3D>
__global__ __launch_bounds__(PJ* PK) void fsm_sweep_tile(const SweepArgs<T> a) {
    constexpr int NT = PJ * PK;
    constexpr int RJ = PJ + 2;                      // tile rows along J incl. halo
    constexpr int NROWS = IS3D ? RJ * (PK + 2) : RJ;
    constexpr int NQ = BL + 2;                      // levels incl. halo
    constexpr int RS = NQ | 1;                      // odd LDS row stride (bank-conflict-free column reads)
    static_assert(IS3D || PK == 1, "2-D uses PK = 1");

    __shared__ T Tt[NROWS * RS];

    const int tid = threadIdx.x;
    unsigned long long prof_t = (FSM_ENABLE_PROF && a.prof) ? wall_clock64() : 0ull;
    const uint32_t tile = a.tiles[blockIdx.x];
    const int TJ = tile & 0xffffu, TK = tile >> 16;
    const int NF = a.g.NF, NJ = a.g.NJ, NK = a.g.NK;
    const int j0 = TJ * PJ, k0 = TK * PK;
    const int jmaxp = (j0 + PJ < NJ ? j0 + PJ : NJ) - 1;
    const int kmaxp = (k0 + PK < NK ? k0 + PK : NK) - 1;
    const int L0 = BL * a.w - (TJ + TK) * (BL - 1);
    // levels at which this patch has nodes (the host only lists such tiles; kept as a guard)
    if (L0 + BL - 1 < j0 + k0 || L0 > jmaxp + kmaxp + NF - 1) return;

    const int slot = a.slots[blockIdx.z];
    if (slot < 0) return;  // source already converged: its blocks are masked out of the batch
    // slot -> (group, lane) of the interleaved field layout
    const int ts = a.ts;
    T* __restrict__ Tg = a.tt + (size_t)(slot / ts) * a.g.n_nodes * ts + slot % ts;
    const T INF = real_traits<T>::inf();
    const int rf = a.rf, rj = a.rj, rk = a.rk;

    // ---- this thread's column and the levels (q = 1..BL <-> level L0-1+q) at which it has a node
    const int tj = tid % PJ, tk = tid / PJ;
    const int jp = j0 + tj, kp = k0 + tk;
    const bool col_ok = jp < NJ && kp < NK;
    const int row = IS3D ? (tk + 1) * RJ + tj + 1 : tj + 1;
    const int qoff = jp + kp - L0 + 1;                       // q at which i' == 0
    const int qa = col_ok ? (qoff > 1 ? qoff : 1) : BL + 1;  // first / last active q
    const int qb = qoff + NF - 1 < BL ? qoff + NF - 1 : BL;

    // ---- node slowness of the own column, straight from the sheared copy into registers
    T sv[BL];
    {
        const T* __restrict__ Sg = a.s_sheared;
        const int M = a.g.M;
        const int jx = a.rev ? NJ - 1 - jp : jp;
        const int kx = a.rev ? NK - 1 - kp : kp;
#pragma unroll
        for (int q = 1; q <= BL; ++q) {
            int x = L0 - 1 + q;  // the level: i' + j' + k' (oriented like this sweep)
            x = (a.rev ? NF + NJ + NK - 3 - x : x) % M;   // ... like the family of the copy
            x = x < 0 ? x + M : x;
            T v = 0;
            if (q >= qa && q <= qb) v = Sg[shear_index(a.g, kx, x, jx)];
            sv[q - 1] = v;
        }
    }

    FSM_PROF_MARK(0)  // setup + slowness loads issued
    // ---- stage the T tile: rows = columns (with halo), entries = levels L0-1 .. L0+BL.
    // All global loads are issued first (registers), then written to LDS, so the HBM/L2 latency
    // is paid once per tile instead of once per element.
    constexpr int TOT = NROWS * NQ;
    constexpr int NLD = (TOT + NT - 1) / NT;
    {
        T tv[NLD];
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int f = tid + it * NT;
            const int r = f / NQ, q = f - r * NQ;
            const int hj = r % RJ - 1;
            const int hk = IS3D ? r / RJ - 1 : 0;
            const bool halo_j = (hj < 0) | (hj >= PJ);
            const bool halo_k = IS3D && ((hk < 0) | (hk >= PK));
            // corner rows are never read; upwind halo rows are read at q-1 (q in 0..BL-1),
            // downwind ones at q+1 (2..BL+1): do not touch what a concurrent tile may be writing
            bool need = (f < TOT) & !(halo_j & halo_k);
            need &= !((((hj < 0) | (hk < 0)) & (q > BL - 1)) | (((hj >= PJ) | (IS3D && hk >= PK)) & (q < 2)));
            const int jq = j0 + hj, kq = k0 + hk;
            const int ip = L0 - 1 + q - jq - kq;
            T v = INF;
            if (need && jq >= 0 && jq < NJ && kq >= 0 && kq < NK && ip >= 0 && ip < NF) {
                const int i = rf ? NF - 1 - ip : ip;
                const int j = rj ? NJ - 1 - jq : jq;
                const int k = rk ? NK - 1 - kq : kq;
                v = Tg[(size_t)(((uint32_t)k * NJ + j) * NF + i) * ts];
            }
            tv[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int f = tid + it * NT;
            const int r = f / NQ, q = f - r * NQ;
            if (f < TOT) Tt[r * RS + q] = tv[it];
        }
    }

    // ---- does this tile touch the frozen (source) neighbourhood?  block-uniform test
    const int* bb = a.bbox + 6 * slot;
    bool near_src;
    {
        int ilo = L0 - jmaxp - kmaxp, ihi = L0 + BL - 1 - j0 - k0;
        ilo = ilo < 0 ? 0 : ilo;
        ihi = ihi > NF - 1 ? NF - 1 : ihi;
        const int flo = rf ? NF - 1 - ihi : ilo, fhi = rf ? NF - 1 - ilo : ihi;
        const int jlo = rj ? NJ - 1 - jmaxp : j0, jhi = rj ? NJ - 1 - j0 : jmaxp;
        const int klo = rk ? NK - 1 - kmaxp : k0, khi = rk ? NK - 1 - k0 : kmaxp;
        near_src = !(fhi < bb[0] || flo > bb[1] || jhi < bb[2] || jlo > bb[3] || khi < bb[4] || klo > bb[5]);
    }
    const uint32_t* __restrict__ Fz = a.frozen + (size_t)slot * a.mask_words;
    const int jn = rj ? NJ - 1 - jp : jp;
    const int kn = rk ? NK - 1 - kp : kp;
    const uint32_t colbase = ((uint32_t)kn * NJ + jn) * NF;
    const T dx = a.dx, dz = a.dz;
    const int variant = a.variant;
    T dec = 0;  // decrease of this thread's nodes in this tile
    bool changed = false;

    __syncthreads();
    FSM_PROF_MARK(1)  // T tile staged

    // own column: levels L0-1 .. L0+BL live in registers; LDS only carries values between columns
    T own[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) own[q] = Tt[row * RS + q];

#pragma unroll
    for (int q = 1; q <= BL; ++q) {
        bool active = (q >= qa) & (q <= qb);
        if (near_src && active) {
            const int ip = L0 - 1 + q - jp - kp;
            const uint32_t n = colbase + (rf ? NF - 1 - ip : ip);
            active = !((Fz[n >> 5] >> (n & 31)) & 1u);
        }
        const T c = own[q];
        // axis minima: a missing neighbour is +inf in the tile == the reference's one-sided pick
        const T af = vmin(own[q - 1], own[q + 1]);
        const T aj = vmin(Tt[(row - 1) * RS + q - 1], Tt[(row + 1) * RS + q + 1]);
        const T s = sv[q - 1];
        T t;
        if (IS3D) {
            const T ak = vmin(Tt[(row - RJ) * RS + q - 1], Tt[(row + RJ) * RS + q + 1]);
            t = update3(ak, aj, af, s, dx, __builtin_amdgcn_ballot_w64(active));
        } else {
            // 2-D: J axis is x (a), F axis is z (b)
            t = variant == 1 ? update2(aj, af, s, dx) : update2_xz(aj, af, s, dx, dz);
        }
        const bool acc = active & (t < c);
        const T nv = acc ? t : c;
        dec += acc ? c - t : (T)0;
        changed |= acc;
        own[q] = nv;
        Tt[row * RS + q] = nv;
        __syncthreads();
    }

    FSM_PROF_MARK(2)  // level march
    if (FSM_ENABLE_PROF && a.prof && tid == 0) atomicAdd(a.prof + 4, 1ull);
    // ---- write back (only when something in the tile changed)
    if (__syncthreads_or(changed)) {
        for (int f = tid; f < NT * BL; f += NT) {
            const int c = f / BL, q = f - c * BL;
            const int cj = c % PJ, ck = c / PJ;
            const int jq = j0 + cj, kq = k0 + ck;
            const int ip = L0 + q - jq - kq;
            if (jq < NJ && kq < NK && ip >= 0 && ip < NF) {
                const int i = rf ? NF - 1 - ip : ip;
                const int j = rj ? NJ - 1 - jq : jq;
                const int k = rk ? NK - 1 - kq : kq;
                const int r2 = IS3D ? (ck + 1) * RJ + cj + 1 : cj + 1;
                Tg[(size_t)(((uint32_t)k * NJ + j) * NF + i) * ts] = Tt[r2 * RS + q + 1];
            }
        }
        // wavefront (64-lane) reduction of the decrease, one atomic per wave
        double accd = (double)dec;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) accd += __shfl_down(accd, off, 64);
        if ((tid & 63) == 0 && accd != 0.0) atomicAdd(a.change + slot, accd);
        FSM_PROF_MARK(3)  // write-back issued
    }
}

// ---- persistent sweep kernel: one launch per directional sweep -----------------------------
// Same skewed pencil march, but a workgroup keeps its patch for the whole sweep and walks its
// levels in chunks of C; the ordering between patches is carried by per-patch progress counters
// in HBM instead of by kernel boundaries:
//   * work units (patch, source) are handed out by an atomic ticket in an order that is a
//     topological order of the patch dependencies (anti-diagonal m = TJ+TK first), so a unit only
//     ever waits for units that already started -- no co-residency assumption, no deadlock;
//   * a unit may run chunk [Lc, Lc+C) once both upwind patches have published every level
//     <= Lc+C-2 (prog[patch] = number of final levels in HBM);
//   * hand-off through HBM, MI355X_MICROARCH.md recipe R1 (XCD L2s are not coherent with each
//     other, a CU's L1 is never refreshed by other CUs' stores): the columns a downstream patch
//     reads are stored write-through (relaxed agent-scope atomic store = `global_store ... sc1`),
//     every storing wave drains `s_waitcnt vmcnt(0)`, barrier, ONE lane stores the counter
//     (relaxed, agent); the consumer polls that one word (relaxed, agent, one lane, s_sleep),
//     barrier, then reads the upwind halo columns with sc1 loads (bypass L1).
//   * every spin is bounded (wall clock); on time-out a global abort word makes every waiter
//     leave, and the host turns it into an error.
template <typename T>
struct PersistArgs {
    SweepArgs<T> s;           // w, tiles unused
    const uint32_t* order;    // ticket order.  One launch per sweep: patches (TJ | TK<<16) by anti-diagonal m = TJ+TK;
                              // whole-iteration launch (XS): units (TJ | TK<<14 | dir<<28) of all directions
    int* sync;                // [0..3]: ticket counters (launch seq & 3), [4]: abort flag, from [8] on the progress words
                              // ((1 + seq % 3) << 30 | value) per (direction,) batch entry and patch
    int n_patches, batch;
    unsigned long long timeout_ticks;  // 100 MHz wall-clock ticks
    // dirty-brick tracking (exact skipping of chunks that cannot change anything)
    int* stamp;               // [n_slots][nbf*nbj*nbk] global sweep number of the last change in a brick, -1: never
    const int* iter_ptr;      // device words: [0] index of the current iteration, [1] launch epoch of its (first) sweep launch
    unsigned long long* evals;  // [n_slots] node updates actually evaluated
    int nbf, nbj, nbk;        // bricks of FSM_BRICK^3 nodes (natural coordinates)
    int dir, ndir;            // direction index within the iteration, directions per iteration
    int skip;                 // 0: evaluate every chunk
    // exact skipping (SKIP kernels), see "scheduler" in fsm_sweep_persistent
    unsigned long long* cmap; // [unit][2][cw] per-chunk change flags of a unit's J-edge / K-edge columns: (launch epoch << 32) | 32 flag bits
    int cw;                   // words per edge and unit
    unsigned long long* sw;   // [global sweep number][n_sw_groups]: units finished (low 32) | units that changed a node (high 32);
                              // zeroed per solve; nullptr: no whole-sweep shortcut
    int n_sw_groups, n_sw_sweeps;
    // whole-iteration launch (XS): all directions in one grid, sheared copies by family
    const T* ssh;             // [families][ssh_stride]
    size_t ssh_stride;
};



__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load((const unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __longlong_as_double(__hip_atomic_load((const long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store((unsigned int*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store((long long*)p, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- third-order WENO stage (second stage of the reference's default solver) ------------
// Grid3Drn::weno3_upwind (ttcr/Grid3Drn.h:3047-3075) / the inlined 2-D form (ttcr/Grid2Drn.h:
// 1078-1125).  Every named intermediate of the reference is a T1 and its double literals make the
// right-hand sides double: reproduced operation by operation (no fused shortcuts: the divisions
// are IEEE divisions).  v1..v4 / v0..v3 are in NATURAL index order along the axis.
// x / h2 for h2 = 2*(double)h with h a float, r2 = RN(1/h2) (one true division per kernel).  h2 has a
// 24-bit significand, so the residual x - q0*h2 is exact and x/h2 can never sit closer than 2^-25 ulp to
// a rounding boundary, while q0 + res*r2 is off by < 2^-52 ulp: ONE fused correction gives the
// correctly rounded quotient (3 ops instead of the ~17 of an IEEE division).  x is finite or NaN here
// (sums of a few float-range terms in double), so no infinity fix-up is needed.
__device__ __forceinline__ double div_h2(double x, double h2, double r2) {
    const double q0 = x * r2;
    return __builtin_fma(__builtin_fma(-q0, h2, x), r2, q0);
}

__device__ __forceinline__ float weno_fwd(float v1, float v2, float v3, float v4, float h, double r2) {
    const float eps = 1.1920928955078125e-07f;
    const float num = (float)(((double)v4 - 2.0 * (double)v3) + (double)v2);
    const float den = (float)(((double)v3 - 2.0 * (double)v2) + (double)v1);
    const float r = (eps + num * num) / (eps + den * den);
    const float w = (float)(1.0 / (1.0 + (2.0 * (double)r) * (double)r));
    const float d31 = v3 - v1;
    const double h2 = 2.0 * (double)h;
    const double ap = div_h2((1.0 - (double)w) * (double)d31, h2, r2) +
                      div_h2((double)w * ((-(double)v4 + 4.0 * (double)v3) - 3.0 * (double)v2), h2, r2);
    return v2 + h * (float)ap;
}
__device__ __forceinline__ float weno_bwd(float v0, float v1, float v2, float v3, float h, double r2) {
    const float eps = 1.1920928955078125e-07f;
    const float num = (float)(((double)v2 - 2.0 * (double)v1) + (double)v0);
    const float den = (float)(((double)v3 - 2.0 * (double)v2) + (double)v1);
    const float r = (eps + num * num) / (eps + den * den);
    const float w = (float)(1.0 / (1.0 + (2.0 * (double)r) * (double)r));
    const float d31 = v3 - v1;
    const double h2 = 2.0 * (double)h;
    const double am = div_h2((1.0 - (double)w) * (double)d31, h2, r2) +
                      div_h2((double)w * ((3.0 * (double)v2 - 4.0 * (double)v1) + (double)v0), h2, r2);
    return v2 - h * (float)am;
}
__device__ __forceinline__ double weno_fwd(double v1, double v2, double v3, double v4, double h, double) {
    const double eps = 2.220446049250313e-16;
    const double num = (v4 - 2.0 * v3 + v2);
    const double den = (v3 - 2.0 * v2 + v1);
    const double r = (eps + num * num) / (eps + den * den);
    const double w = 1.0 / (1.0 + 2.0 * r * r);
    const double ap = (1.0 - w) * (v3 - v1) / (2.0 * h) + w * (-v4 + 4.0 * v3 - 3.0 * v2) / (2.0 * h);
    return v2 + h * ap;
}
__device__ __forceinline__ double weno_bwd(double v0, double v1, double v2, double v3, double h, double) {
    const double eps = 2.220446049250313e-16;
    const double num = (v2 - 2.0 * v1 + v0);
    const double den = (v3 - 2.0 * v2 + v1);
    const double r = (eps + num * num) / (eps + den * den);
    const double w = 1.0 / (1.0 + 2.0 * r * r);
    const double am = (1.0 - w) * (v3 - v1) / (2.0 * h) + w * (3.0 * v2 - 4.0 * v1 + v0) / (2.0 * h);
    return v2 - h * am;
}

// One axis of update_node_weno3 (ttcr/Grid3Drn.h:3084-3196): m2..p2 are the values at natural
// offsets -2..+2 along the axis, idx the node's natural index, n the last index.  The reference's
// if / else-if chain (first node, second, last, last but one, interior) as selects on ONE forward and
// ONE backward stencil: the stencils of the cases that do not apply are evaluated on whatever the
// out-of-range entries hold and dropped (no traps on the GPU), every case keeps its own operands and
// its own `a < t ? a : t`.  One copy of the stencil arithmetic instead of four keeps the unrolled level
// loop of the WENO kernel inside the instruction cache.
template <typename T>
__device__ __forceinline__ T weno_axis(T m2, T m1, T c, T p1, T p2, int idx, int n, T h, double r2) {
    const bool c0 = idx == 0;
    const bool c1 = !c0 && idx == 1;
    const bool cn = !c0 && !c1 && idx == n;
    const bool cm = !c0 && !c1 && !cn && idx == n - 1;
    const T F = weno_fwd(m1, c, p1, p2, h, r2);
    const T B = weno_bwd(m2, m1, c, p1, h, r2);
    T a = cm ? B : F;
    const T t = c1 ? m1 : (cm ? p1 : B);
    a = a < t ? a : t;
    return c0 ? p1 : (cn ? m1 : a);
}

// fp32 grids: the same two stencils with the work they share done once, and the divisions of TAME operands written out.
//  * second differences and one-sided differences: 2 v, 3 v and 4 v are exact in double for a float v, so
//    fma(-2, b, a) rounds like the reference's a - 2.0 * b (one rounding either way), fma(2 r, r, 1) like 1.0 + 2.0 * r * r;
//  * both stencils divide by the same eps + den^2: one reciprocal refinement serves both quotients;
//  * the operation sequence of an IEEE division on this target is  scale, rcp, two (f64: four) fused refinement steps,
//    quotient, residual, [f32: a second correction,] final fused correction, fix-up;  for operands whose quotient and
//    reciprocal stay far from the ends of the exponent range the scale and fix-up steps are identities (fp32: numerator
//    and denominator in [2^-23, 2^73) -- they are eps + a square, so only the upper bound is tested; fp64: the divisor
//    1 + 2 r^2 is in [1, 2^193) then).  The wavefront takes the written-out sequence when every lane is tame, the
//    compiler's division otherwise (next to unreached or out-of-grid entries): bit-identical either way, 24 instructions
//    instead of 44 for the four divisions of an axis.
#ifndef FSM_WENO_PLAIN
#define FSM_WENO_PLAIN 0   // 1: the operation-by-operation form above for fp32 too (A/B builds)
#endif
__device__ __forceinline__ float weno_quot_tame(float A, float B, float y) {
    float q = A * y;
    float r = __builtin_fmaf(-B, q, A);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-B, q, A);
    return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ double weno_recip_tame(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double r = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(r, y, y);
}
// One axis in three steps, so that ONE wave-uniform branch covers the divisions of all axes of a node and the independent
// chains of the axes stay in one basic block (a lone wavefront issues in order: they fill each other's latencies).
struct WenoAxisF {
    double dm2, dm1, dc, dp1, dp2;
    float Bq, AF, AB;   // eps + den^2, eps + num^2 of the forward / backward stencil
    float wF, wB;
};
__device__ __forceinline__ WenoAxisF weno_pre(float m2, float m1, float c, float p1, float p2) {
    const float eps = 1.1920928955078125e-07f;
    WenoAxisF x;
    x.dm2 = m2; x.dm1 = m1; x.dc = c; x.dp1 = p1; x.dp2 = p2;
    const float den = (float)(__builtin_fma(-2.0, x.dc, x.dp1) + x.dm1);   // (p1 - 2 c) + m1: `den` of both stencils
    const float nF = (float)(__builtin_fma(-2.0, x.dp1, x.dp2) + x.dc);    // forward:  (p2 - 2 p1) + c
    const float nB = (float)(__builtin_fma(-2.0, x.dm1, x.dc) + x.dm2);    // backward: (c - 2 m1) + m2
    x.Bq = eps + den * den; x.AF = eps + nF * nF; x.AB = eps + nB * nB;
    return x;
}
__device__ __forceinline__ float weno_span(const WenoAxisF& x) { return __builtin_fmaxf(__builtin_fmaxf(x.AF, x.AB), x.Bq); }
__device__ __forceinline__ void weno_weights_tame(WenoAxisF& x) {
    float y = __builtin_amdgcn_rcpf(x.Bq);
    const float e = __builtin_fmaf(-x.Bq, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    const double rF = weno_quot_tame(x.AF, x.Bq, y), rB = weno_quot_tame(x.AB, x.Bq, y);
    x.wF = (float)weno_recip_tame(__builtin_fma(2.0 * rF, rF, 1.0));
    x.wB = (float)weno_recip_tame(__builtin_fma(2.0 * rB, rB, 1.0));
}
// The compiler's divisions for the wavefronts that are not tame: ONE copy for the whole kernel, reached by a call -- inlined
// into the eight unrolled levels it costs the level march a fifth more code, and the loop no longer fits the instruction cache
// (measured: 256^3 single source 421 ms -> 488 ms with the inlined copies, 384 ms without them).
struct WenoW6 { float w[6]; };
__device__ __attribute__((noinline)) WenoW6 weno_weights_ieee(float AF0, float AB0, float B0, float AF1, float AB1, float B1,
                                                              float AF2, float AB2, float B2) {
    const float A[6] = {AF0, AB0, AF1, AB1, AF2, AB2};
    const float B[3] = {B0, B1, B2};
    WenoW6 o;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double r = A[i] / B[i >> 1];
        o.w[i] = (float)(1.0 / __builtin_fma(2.0 * r, r, 1.0));
    }
    return o;
}
__device__ __forceinline__ float weno_post(const WenoAxisF& x, float m1, float c, float p1, int idx, int n, float h, double r2) {
    const double d31 = p1 - m1;   // (float difference, then widened: T1 d31 = v3 - v1)
    const double h2 = 2.0 * (double)h;
    const double ap = div_h2((1.0 - (double)x.wF) * d31, h2, r2) +
                      div_h2((double)x.wF * __builtin_fma(-3.0, x.dc, __builtin_fma(4.0, x.dp1, -x.dp2)), h2, r2);
    const double am = div_h2((1.0 - (double)x.wB) * d31, h2, r2) +
                      div_h2((double)x.wB * (__builtin_fma(-4.0, x.dm1, 3.0 * x.dc) + x.dm2), h2, r2);
    const float F = c + h * (float)ap;
    const float B = c - h * (float)am;
    const bool c0 = idx == 0;
    const bool c1 = !c0 && idx == 1;
    const bool cn = !c0 && !c1 && idx == n;
    const bool cm = !c0 && !c1 && !cn && idx == n - 1;
    float a = cm ? B : F;
    const float t = c1 ? m1 : (cm ? p1 : B);
    a = a < t ? a : t;
    return c0 ? p1 : (cn ? m1 : a);
}
// the axis values of one node: K, J, F (2-D: J, F)
template <bool IS3D>
__device__ __forceinline__ void weno_axes(const float (&vF)[5], const float (&vJ)[5], const float (&vK)[5], int iF, int nF, int iJ, int nJ,
                                          int iK, int nK, float hF, double r2F, float hJ, double r2J, float& aF, float& aJ, float& aK) {
    WenoAxisF xF = weno_pre(vF[0], vF[1], vF[2], vF[3], vF[4]);
    WenoAxisF xJ = weno_pre(vJ[0], vJ[1], vJ[2], vJ[3], vJ[4]);
    WenoAxisF xK = xJ;
    float span = __builtin_fmaxf(weno_span(xF), weno_span(xJ));
    if (IS3D) {
        xK = weno_pre(vK[0], vK[1], vK[2], vK[3], vK[4]);
        span = __builtin_fmaxf(span, weno_span(xK));
    }
    const bool tame = span < 0x1p73f;   // (a NaN operand gives NaN on both paths)
#ifndef FSM_WENO_EXP
#define FSM_WENO_EXP 0   // TIMING builds: 1 = no test, every wavefront takes the written-out divisions
#endif
    if (FSM_WENO_EXP == 1 || __builtin_expect(__builtin_amdgcn_ballot_w64(!tame) == 0ull, 1)) {
        weno_weights_tame(xF); weno_weights_tame(xJ);
        if (IS3D) weno_weights_tame(xK);
    } else {
        const WenoW6 o = weno_weights_ieee(xF.AF, xF.AB, xF.Bq, xJ.AF, xJ.AB, xJ.Bq, xK.AF, xK.AB, xK.Bq);
        xF.wF = o.w[0]; xF.wB = o.w[1]; xJ.wF = o.w[2]; xJ.wB = o.w[3]; xK.wF = o.w[4]; xK.wB = o.w[5];
    }
    aF = weno_post(xF, vF[1], vF[2], vF[3], iF, nF, hF, r2F);
    aJ = weno_post(xJ, vJ[1], vJ[2], vJ[3], iJ, nJ, hJ, r2J);
    aK = IS3D ? weno_post(xK, vK[1], vK[2], vK[3], iK, nK, hJ, r2J) : 0.0f;
}

// Local solver of the WENO stage: the reference's literal compare/swap network and nested ifs
// (ttcr/Grid3Drn.h:3432-3452) -- WENO axis values may be NaN/inf next to unreached nodes, and
// min/max/med3 would not propagate them the way the swaps do.  Discriminants as in update3.
__device__ __forceinline__ float solve3_literal(float a1, float a2, float a3, float fh) {
    if (a1 > a2) { const float w = a1; a1 = a2; a2 = w; }
    if (a1 > a3) { const float w = a1; a1 = a3; a3 = w; }
    if (a2 > a3) { const float w = a2; a2 = a3; a3 = w; }
    float t = a1 + fh;
    if (t > a2) {
        const double d1 = a1, d2 = a2, dfh = fh;
        const float df = a1 - a2;
        const float df2 = df * df;
        t = (float)(0.5 * ((double)(a1 + a2) + __builtin_sqrt(__builtin_fma(dfh * dfh, 2.0, -(double)df2))));
        if (t > a3) {
            const double d3 = a3;
            double r = (-2.0 * d1) * d1;
            r = __builtin_fma(2.0 * d1, d2, r);
            r = __builtin_fma(-2.0 * d2, d2, r);
            r = __builtin_fma(2.0 * d1, d3, r);
            r = __builtin_fma(2.0 * d2, d3, r);
            r = __builtin_fma(-2.0 * d3, d3, r);
            r = __builtin_fma(3.0 * dfh, dfh, r);
            t = (float)((1. / 3.) * ((double)((a1 + a2) + a3) + __builtin_sqrt(r)));
        }
    }
    return t;
}
__device__ __forceinline__ double solve3_literal(double a1, double a2, double a3, double fh) {
    if (a1 > a2) { const double w = a1; a1 = a2; a2 = w; }
    if (a1 > a3) { const double w = a1; a1 = a3; a3 = w; }
    if (a2 > a3) { const double w = a2; a2 = a3; a3 = w; }
    double t = a1 + fh;
    if (t > a2) {
        t = 0.5 * (a1 + a2 + __builtin_sqrt(2. * fh * fh - (a1 - a2) * (a1 - a2)));
        if (t > a3) {
            t = 1. / 3. * ((a1 + a2 + a3) + __builtin_sqrt(-2. * a1 * a1 + 2. * a1 * a2 - 2. * a2 * a2 + 2. * a1 * a3 +
                                                            2. * a2 * a3 - 2. * a3 * a3 + 3. * fh * fh));
        }
    }
    return t;
}

// ---- WENO stage with tolerance-grade arithmetic (option "arith" = 1; the AR = 1 instantiations with H = 2) ----------------------
// One axis of update_node_weno3 (ttcr/Grid3Drn.h:3084-3196 with weno3_upwind :3047-3075) in fp32.  What keeps it within the tolerance:
// the smoothness ratios are quotients of squared SECOND differences of traveltimes that differ in their last digits -- they are formed from
// first differences of neighbours (exact in fp32 wherever the two values lie within a factor of two of each other: everywhere but next to
// the source), so num and den are the rounded second differences of the stored values, as in the reference's double evaluation.  The
// one-sided derivative times h is   c +- ((1 - w) (p1 - m1) + w (3 d_near - d_far)) / 2   -- the reference's division by 2 h and
// multiplication by h cancel.  Quotients by v_rcp_f32 (1 ulp): the weights move by ~1e-7, the derivative by that times the difference of
// its two stencils.
__device__ __forceinline__ float weno_axis_fast(float m2, float m1, float c, float p1, float p2, int idx, int n) {
    const float eps = 1.1920928955078125e-07f;
    const float d0 = c - m1, d1 = p1 - c, d2 = p2 - p1, dm = m1 - m2;
    const float den = d1 - d0, nF = d2 - d1, nB = d0 - dm;
    const float y = __builtin_amdgcn_rcpf(__builtin_fmaf(den, den, eps));
    const float rF = __builtin_fmaf(nF, nF, eps) * y, rB = __builtin_fmaf(nB, nB, eps) * y;
    const float wF = __builtin_amdgcn_rcpf(__builtin_fmaf(2.0f * rF, rF, 1.0f));
    const float wB = __builtin_amdgcn_rcpf(__builtin_fmaf(2.0f * rB, rB, 1.0f));
    const float d31 = p1 - m1;
    // (1 - w) d31 + w s = d31 + w (s - d31)
    const float F = __builtin_fmaf(0.5f, __builtin_fmaf(wF, __builtin_fmaf(3.0f, d1, -d2) - d31, d31), c);
    const float B = __builtin_fmaf(-0.5f, __builtin_fmaf(wB, __builtin_fmaf(3.0f, d0, -dm) - d31, d31), c);
    const bool c0 = idx == 0;
    const bool c1 = !c0 && idx == 1;
    const bool cn = !c0 && !c1 && idx == n;
    const bool cm = !c0 && !c1 && !cn && idx == n - 1;
    float a = cm ? B : F;
    const float t = c1 ? m1 : (cm ? p1 : B);
    a = a < t ? a : t;
    return c0 ? p1 : (cn ? m1 : a);
}
// Local solver of the WENO stage (ttcr/Grid3Drn.h:3432-3452): the literal compare / swap network and nested conditions of
// solve3_literal (NaN and inf axis values travel as they do there), discriminants in fp32 -- the 2-D one as the reference forms it, the
// 3-D one on differences from the smallest value --, hardware roots.
__device__ __forceinline__ float solve3_literal_fast(float a1, float a2, float a3, float fh) {
    if (a1 > a2) { const float w = a1; a1 = a2; a2 = w; }
    if (a1 > a3) { const float w = a1; a1 = a3; a3 = w; }
    if (a2 > a3) { const float w = a2; a2 = a3; a3 = w; }
    const float t1 = a1 + fh;
    const float df = a1 - a2, d3 = a3 - a1;
    const float df2 = df * df;
    const float t2 = 0.5f * ((a1 + a2) + __builtin_amdgcn_sqrtf(__builtin_fmaf(2.0f * fh, fh, -df2)));
    const float e = d3 + df;
    const float disc3 = __builtin_fmaf(3.0f * fh, fh, -(df2 + __builtin_fmaf(d3, d3, e * e)));
    const float t3 = __builtin_fmaf(1.0f / 3.0f, (d3 - df) + __builtin_amdgcn_sqrtf(disc3), a1);
    return t1 > a2 ? (t2 > a3 ? t3 : t2) : t1;
}

// ---- persistent sweep kernel, halo width H (1: first-order stage, 2: WENO3 stage) -----------
#ifndef FSM_POLL_SLEEP
#define FSM_POLL_SLEEP 2   // s_sleep argument (x64 clocks) between two polls of a progress counter
#endif
#ifndef FSM_MINW
#define FSM_MINW 1
#endif
// SKIP kernels: F bricks (of FSM_BRICK nodes) a grid may have along the fast axis, in 32-bit words of the slab mask
#ifndef FSM_SLAB_WORDS
#define FSM_SLAB_WORDS 32
#endif
// SKIP kernels publish progress as (levels << 8) | change flags of the last four chunks (two bits each: J-edge, K-edge columns);
// a finished unit publishes FSM_FIN | (ever changed its J edge) | (ever changed its K edge) << 1
#define FSM_FIN 0x3ffffff0
#ifndef FSM_SKIP_ABL
#define FSM_SKIP_ABL 0   // tuning builds: leave parts of the skip bookkeeping out (wrong results unless every brick is dirty)
#endif
#ifndef FSM_EARLY_PUB
#define FSM_EARLY_PUB 0   // first chunks of a unit whose progress is published right after their write-back
#endif
// NS sources of a slot group marched together (field layout T[group][node][NS]): one element type
// value held by the lane below / above (DPP wave shift, one VALU op; the end lanes get `edge`)
__device__ __forceinline__ float lane_below(float x, float edge) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(edge), (int)__float_as_uint(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_above(float x, float edge) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(edge), (int)__float_as_uint(x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_below(double x, double edge) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x), e = (unsigned long long)__double_as_longlong(edge);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)e, (int)(unsigned)u, 0x138, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(e >> 32), (int)(unsigned)(u >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double lane_above(double x, double edge) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x), e = (unsigned long long)__double_as_longlong(edge);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)e, (int)(unsigned)u, 0x130, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(e >> 32), (int)(unsigned)(u >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <typename T, int NS> struct Pack;
template <typename T> struct Pack<T, 1> { T v[1]; };
template <typename T> struct alignas(2 * sizeof(T)) Pack<T, 2> { T v[2]; };

template <typename T, int NS>
__device__ __forceinline__ Pack<T, NS> pack_fill(T x) {
    Pack<T, NS> p;
#pragma unroll
    for (int l = 0; l < NS; ++l) p.v[l] = x;
    return p;
}
__device__ __forceinline__ Pack<float, 1> ld_sc1(const Pack<float, 1>* p) { Pack<float, 1> r; r.v[0] = ld_sc1(&p->v[0]); return r; }
__device__ __forceinline__ Pack<double, 1> ld_sc1(const Pack<double, 1>* p) { Pack<double, 1> r; r.v[0] = ld_sc1(&p->v[0]); return r; }
__device__ __forceinline__ Pack<float, 2> ld_sc1(const Pack<float, 2>* p) {
    const unsigned long long u = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Pack<float, 2> r;
    r.v[0] = __uint_as_float((unsigned)u);
    r.v[1] = __uint_as_float((unsigned)(u >> 32));
    return r;
}
__device__ __forceinline__ Pack<double, 2> ld_sc1(const Pack<double, 2>* p) {
    Pack<double, 2> r;
    r.v[0] = ld_sc1(&p->v[0]);
    r.v[1] = ld_sc1(&p->v[1]);
    return r;
}
__device__ __forceinline__ void st_sc1(Pack<float, 1>* p, Pack<float, 1> x) { st_sc1(&p->v[0], x.v[0]); }
__device__ __forceinline__ void st_sc1(Pack<double, 1>* p, Pack<double, 1> x) { st_sc1(&p->v[0], x.v[0]); }
__device__ __forceinline__ void st_sc1(Pack<float, 2>* p, Pack<float, 2> x) {
    const unsigned long long u = (unsigned long long)__float_as_uint(x.v[0]) | ((unsigned long long)__float_as_uint(x.v[1]) << 32);
    __hip_atomic_store((unsigned long long*)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(Pack<double, 2>* p, Pack<double, 2> x) {
    st_sc1(&p->v[0], x.v[0]);
    st_sc1(&p->v[1], x.v[1]);
}

// XS = true: ONE launch per sweep-iteration.  Tickets run direction-major, so the patches of sweep
// d+1 start while sweep d is still draining its last anti-diagonals: a unit (d, patch) begins as
// soon as sweep d-1 has finished every patch that owns a column within 2H of its own columns
// (those are all the writers of what it reads and all the readers of what it writes).  Every
// traveltime access is then an agent-scope (sc1) access: values written by another XCD within
// the same launch must come from memory, not from a stale L1 line or a dirty remote L2.
// PRE = true: lane 0 samples the upwind progress counters one chunk ahead (see the wait at the top of the
// chunk loop).  Pays off with several units in flight per patch position (64 sources: +3.8 %) and for the one-wave
// 2-D patches (a single 4096^2 solve: +24 %); a lone 3-D source is 4 % better off without.
// Which instantiations keep their workgroups for more than one unit: the first-order 3-D kernels (see fsm_sweep_persistent).
__host__ __device__ constexpr bool fsm_looped(bool is3d, int h) { return is3d && h == 1; }

__device__ __forceinline__ int kmaxp_of(int k0, int NK, int PK) { return (k0 + PK < NK ? k0 + PK : NK) - 1; }
// One work unit: the body of fsm_sweep_persistent below.  Returns false when the tickets of the launch have run out.
template <typename T, int PJ, int PK, int C, bool IS3D, bool SKIP, int H, int NS, bool XS, bool PRE, int AR = 0>
__device__ __forceinline__ bool fsm_sweep_unit(const PersistArgs<T>& pa) {
    static_assert(AR == 0 || std::is_same<T, float>::value, "tolerance-grade arithmetic: fp32 kernels");
    constexpr bool LOOPED = fsm_looped(IS3D, H);   // the workgroup comes back for another unit (see fsm_sweep_persistent)
    using P = Pack<T, NS>;
    constexpr int NT = PJ * PK;
    constexpr int RJ = PJ + 2 * H;
    constexpr int NROWS = IS3D ? RJ * (PK + 2 * H) : RJ;
    constexpr int NQ = C + 2 * H;               // tile levels: L0-H .. L0+C+H-1  <->  q = 0 .. NQ-1
    constexpr int RS = NQ | 1;
    constexpr int NUP = IS3D ? H * (PJ + PK) : H;   // upwind halo columns
    constexpr int NDH = IS3D ? H * (PJ + PK) : H;   // downwind halo columns
    // streaming map: C consecutive lanes walk C consecutive levels of one column, NT/C columns per pass
    constexpr int RPI = NT / C;
    constexpr int NOWN = C;                      // passes over the own columns (NT / RPI)
    constexpr int NHI = (NDH + RPI - 1) / RPI;   // passes over the downwind halo columns
    constexpr int NUPI = (NUP + RPI - 1) / RPI;  // passes over the upwind halo columns
    static_assert(NT % C == 0 && (IS3D || PK == 1) && C <= FSM_BRICK && (H == 1 || H == 2), "tile shape");
    static_assert(C - 1 <= FSM_XPAD, "the rows of a chunk are read from the sheared copies without a wrap");
    // (the early publish would let a unit's final progress value go out before its brick stamps: round-3 advice)
    static_assert(!(SKIP && FSM_EARLY_PUB > 0), "FSM_EARLY_PUB is a tuning option of the kernels without exact skipping");
    const SweepArgs<T>& a = pa.s;
    // Launch epoch.  Nothing the kernel synchronises on is reset between the launches of a solve: every word carries the number
    // of the launch that wrote it, and a word of another launch reads as "nothing published" -- what a word held before (the
    // final values of the previous launch) can never be taken for this launch's progress, whatever the order in which a
    // reset would have become visible.  (Round 4: under the HIP runtime that PyTorch bundles, replays of a hipGraph with
    // memset nodes in front of the kernel node let the kernel work with the words of the previous launch once the process
    // had called hipDeviceSynchronize() -- profiles/r04/niter_root_cause.txt.)
    //   * progress words: 32 bits, (e2 << 30) | value with e2 = 1 + seq % 3 -- every word of a launch is rewritten by the
    //     next one (each unit publishes a final value), so within a solve a stale word is exactly one launch old; the host
    //     wipes them once per solve with an ordinary stream memset (zero reads as epoch 0: never);
    //   * change maps: 64 bits, (seq << 32) | 32 flag bits -- a unit only writes the words it sets a bit in, the rest
    //     may be of any age;
    //   * ticket counters: four, launch seq draws from counter seq & 3 and whoever draws ticket 0 zeroes counter (seq + 2) & 3.
    // One launch per directional sweep: one seq per sweep.
    const unsigned epoch = (unsigned)pa.iter_ptr[1] + (XS ? 0u : (unsigned)pa.dir);
    const int e2 = (int)(epoch % 3u) + 1;
    auto dec_prog = [&](int raw_) -> int { return (int)((unsigned)raw_ >> 30) == e2 ? (raw_ & 0x3fffffff) : 0; };
    auto ld_raw = [&](const int* p_) -> int { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld_prog = [&](const int* p_) -> int { return dec_prog(ld_raw(p_)); };
    auto st_prog = [&](int* p_, int v_) {
        __hip_atomic_store(p_, (int)(((unsigned)e2 << 30) | (unsigned)v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto ld_cmap = [&](const unsigned long long* p_) -> unsigned {
        const unsigned long long v_ = __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (unsigned)(v_ >> 32) == epoch ? (unsigned)v_ : 0u;
    };
    auto st_cmap = [&](unsigned long long* p_, unsigned v_) {
        __hip_atomic_store(p_, ((unsigned long long)epoch << 32) | v_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    __shared__ P Tt[NROWS * RS];
    __shared__ int s_ticket, s_abort;
    __shared__ int s_anychg;
    // scheduler state of the SKIP kernels (thread 0 only, apart from s_slab's set-up and s_next / s_act)
    constexpr int SLABW = SKIP ? FSM_SLAB_WORDS : 1;
    __shared__ unsigned s_stamped[SKIP ? 9 : 1][SLABW];   // bricks of the read set (J/K position, F index) this unit has already stamped
    __shared__ unsigned s_slab[SLABW];   // bit bf: some brick of the unit's read set with F index bf changed in sweep sigma-1 or in this one
    __shared__ int s_next, s_act;        // chunk the scheduler stopped at, and what to do there (1: evaluate, 2: drain first, 0: leave)
    __shared__ int s_cwi[2], s_cwl[2];   // upwind change maps: cached word index, upwind level known when it was loaded
    __shared__ unsigned s_cwv[2];        // ... and the word
    __shared__ int s_mywi;               // own change map: word being filled
    __shared__ unsigned s_mywv[2];
    // per-unit constants and the little state of the scheduler live in LDS, not in registers: they are touched once per
    // chunk by one lane, and the level march has no register to spare (a resident wave per SIMD is at stake)
    enum { U_THR, U_RSJLO, U_RSKLO, U_RSNJ, U_RSNK, U_LCF, U_UPLCF0, U_UPLCF1, U_HIST, U_EVER, U_UCHG, U_PENDV, U_LCHG, U_N };
    __shared__ int s_u[U_N];

    int tid = threadIdx.x;
    if constexpr (LOOPED) asm volatile("" : "+v"(tid));   // (see the loop in fsm_sweep_persistent)
    // debug phase timers (FSM_ENABLE_PROF builds): thread 0 sums per phase in registers, one flush per unit
    unsigned long long prof_t = (FSM_ENABLE_PROF && a.prof) ? wall_clock64() : 0ull;
    unsigned long long pacc[6] = {0, 0, 0, 0, 0, 0};
    unsigned pchunks = 0;
    unsigned long long twait = 0;   // FSM_ENABLE_PROF == 2 (unit trace only, up to 2^20 units): ticks thread 0 spent polling
    constexpr int FSM_TRACE_UNITS = FSM_ENABLE_PROF == 2 ? (1 << 20) : 65536;
    const unsigned long long trace_t0 = prof_t;
    unsigned long long* prec = nullptr;   // set once the ticket is known
    __shared__ unsigned long long s_pacc[FSM_ENABLE_PROF == 2 ? 8 : 1];   // trace-only builds: the phase sums live in LDS
    if constexpr (FSM_ENABLE_PROF == 2) {                                  // (no register of the march is spent on them)
        if (tid == 0) {
            for (int q = 0; q < 6; ++q) s_pacc[q] = 0;
            s_pacc[6] = prof_t;
        }
    }
#define FSM_PMARK(slot_)                                                          \
    if constexpr (FSM_ENABLE_PROF == 2) {                                         \
        if (a.prof && tid == 0) {                                                 \
            const unsigned long long now_ = wall_clock64();                       \
            s_pacc[slot_] += now_ - s_pacc[6];                                    \
            s_pacc[6] = now_;                                                     \
        }                                                                         \
    }                                                                             \
    if (FSM_ENABLE_PROF == 1 && a.prof && tid == 0) {                             \
        const unsigned long long now_ = wall_clock64();                           \
        pacc[slot_] += now_ - prof_t;                                             \
        prof_t = now_;                                                            \
        /* per-chunk stamps of the first units (chunk record region of the trace buffer) */ \
        if ((slot_) < 5 && prec && pchunks < 80u) prec[pchunks * 5u + (slot_)] = now_; \
    }
    const int NF = a.g.NF, NJ = a.g.NJ, NK = a.g.NK;
    const int npj = a.g.npj;
    // counter of the patch of sweep pd (its own oriented partition) that lane `lane` (< 16) has to see finished before
    // unit (d, TJ_, TK_, z_) may touch its columns: the patches owning a column within 2H of the unit's (<= 3 x 3)
    auto prev_sweep_counter = [&](int d, int TJ_, int TK_, int z_, int lane) -> const int* {
        const int pd = d - 1;
        int crj, crk, prj, prk;
        if (IS3D) { crj = (d >> 1) & 1; crk = (d >> 2) & 1; prj = (pd >> 1) & 1; prk = (pd >> 2) & 1; }
        else { crj = (d == 1) | (d == 2); crk = 0; prj = (pd == 1) | (pd == 2); prk = 0; }
        const int j0_ = TJ_ * PJ, k0_ = TK_ * PK;
        const int jm_ = (j0_ + PJ < NJ ? j0_ + PJ : NJ) - 1, km_ = (k0_ + PK < NK ? k0_ + PK : NK) - 1;
        int ja = j0_ - 2 * H, jb = jm_ + 2 * H, ka = k0_ - 2 * H, kb = km_ + 2 * H;
        ja = ja < 0 ? 0 : ja; jb = jb > NJ - 1 ? NJ - 1 : jb;
        ka = ka < 0 ? 0 : ka; kb = kb > NK - 1 ? NK - 1 : kb;
        // oriented (this sweep) -> natural -> oriented (previous sweep)
        const int ja2 = (crj != prj) ? NJ - 1 - jb : ja, jb2 = (crj != prj) ? NJ - 1 - ja : jb;
        const int ka2 = (crk != prk) ? NK - 1 - kb : ka, kb2 = (crk != prk) ? NK - 1 - ka : kb;
        const int tja = ja2 / PJ, ntj = jb2 / PJ - tja + 1;
        const int tka = IS3D ? ka2 / PK : 0, ntk = IS3D ? kb2 / PK - tka + 1 : 1;
        const int ia = lane & 3, ib = lane >> 2;
        if (ia >= ntj || ib >= ntk) return nullptr;
        return pa.sync + 8 + ((size_t)pd * pa.batch + z_) * pa.n_patches + ((tka + ib) * npj + tja + ia);
    };
    if constexpr (LOOPED) __syncthreads();   // (the workgroup comes here once per unit: every read of the previous unit's shared state is over)
    if (tid == 0) {
        const int t_ = atomicAdd(pa.sync + (epoch & 3u), 1);
        // (the counter the launch after the next one draws from: no launch is using it now)
        if (t_ == 0) __hip_atomic_store(pa.sync + ((epoch + 2u) & 3u), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ticket = t_;
    }
    if (LOOPED && tid == 1) s_abort = __hip_atomic_load(pa.sync + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (in flight together)
    __syncthreads();
    if (LOOPED && s_abort) return false;   // a unit timed out: the solve fails on the host, nobody takes another unit
    const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);
    // XS: `order` lists the units (direction, patch) of the whole iteration in ticket order -- any order in which a
    // unit comes after its upwind patches and after the patches of the previous sweep it has to see finished
    const int oidx = ticket / pa.batch, z = ticket - oidx * pa.batch;
    if (oidx >= (XS ? pa.n_patches * pa.ndir : pa.n_patches)) return false;
    if (FSM_ENABLE_PROF == 1 && a.prof && ticket < 8192) prec = a.prof + 8 + 4 * 65536 + (size_t)ticket * 400;
    const uint32_t tile = pa.order[oidx];
    const int dir = XS ? (int)(tile >> 28) : pa.dir;
    const int TJ = XS ? (int)(tile & 0x3fffu) : (int)(tile & 0xffffu), TK = XS ? (int)((tile >> 14) & 0x3fffu) : (int)(tile >> 16);
    int* prog = pa.sync + 8 + ((size_t)(XS ? dir : 0) * pa.batch + z) * pa.n_patches;
    int* my_prog = prog + (TK * npj + TJ);
    // sweep direction: 3-D bits (F, J, K); 2-D order (+x+z, -x+z, -x-z, +x-z) with J = x, F = z
    int rf, rj, rk, rev;
    const T* __restrict__ Sg;
    if (XS) {
        int fam;
        if (IS3D) {
            rf = dir & 1; rj = (dir >> 1) & 1; rk = (dir >> 2) & 1;
            rev = rk;
            fam = (rf ^ rk) | ((rj ^ rk) << 1);
        } else {
            rj = (dir == 1) | (dir == 2); rf = dir >> 1; rk = 0;
            rev = rj;
            fam = rf ^ rj;
        }
        Sg = pa.ssh + (size_t)fam * pa.ssh_stride;
    } else {
        rf = a.rf; rj = a.rj; rk = a.rk; rev = a.rev;
        Sg = a.s_sheared;
    }
    const int grp = a.slots[z];   // slot (NS == 1) or slot group (NS == 2)
    if (grp < 0) {  // converged source(s): nothing to do, but never leave a waiter hanging
        if (tid == 0) st_prog(my_prog, SKIP ? FSM_FIN : 0x3fffffff);
        return true;
    }
    // SKIP: when the previous sweep (of this source group) has finished every patch and changed no node, this sweep cannot
    // change one either (by induction along its own order every node sees the neighbour values of its last visit): the unit
    // is done before it starts, and so is every later sweep until the host looks at the iteration's change
    if constexpr (SKIP) {
        const int sigma0 = pa.ndir * pa.iter_ptr[0] + dir;   // global sweep number, 0-based
        if (pa.sw && sigma0 >= 1 && sigma0 < pa.n_sw_sweeps) {
            if (tid == 0) {
                const unsigned long long v = __hip_atomic_load(pa.sw + (size_t)(sigma0 - 1) * pa.n_sw_groups + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_act = ((unsigned)v == (unsigned)pa.n_patches && (v >> 32) == 0ull) ? 1 : 0;
            }
            __syncthreads();
            if (s_act) {
                if (tid == 0) {
                    st_prog(my_prog, FSM_FIN);
                    atomicAdd(pa.sw + (size_t)sigma0 * pa.n_sw_groups + grp, 1ull);
                    if (FSM_ENABLE_PROF && a.prof && ticket < FSM_TRACE_UNITS) {
                        unsigned long long* tr = a.prof + 8 + 4 * (size_t)ticket;
                        tr[0] = trace_t0; tr[1] = FSM_ENABLE_PROF == 2 ? 0ull : trace_t0; tr[2] = wall_clock64();
                        tr[3] = (unsigned long long)TJ | ((unsigned long long)TK << 16) | ((unsigned long long)dir << 32) | ((unsigned long long)z << 40) | (0xffffull << 48);
                    }
                }
                return true;
            }
        }
    }
#ifndef FSM_EXP_NOWAIT
#define FSM_EXP_NOWAIT 0   // TIMING experiment (wrong results): no unit ever waits for another one
#endif
    const int* up_j = (!FSM_EXP_NOWAIT && TJ > 0) ? prog + (TK * npj + TJ - 1) : nullptr;
    const int* up_k = (!FSM_EXP_NOWAIT && IS3D && TK > 0) ? prog + ((TK - 1) * npj + TJ) : nullptr;

    const int j0 = TJ * PJ, k0 = TK * PK;
    const int jmaxp = (j0 + PJ < NJ ? j0 + PJ : NJ) - 1;
    const int kmaxp = (k0 + PK < NK ? k0 + PK : NK) - 1;
    const int Ls = j0 + k0, Le = jmaxp + kmaxp + NF - 1;  // levels at which the patch has nodes
    // 2-D, sweeps 1 and 3 (-x+z after +x+z, +x-z after -x-z): only the J direction flips, the columns are walked the same
    // way as in the sweep before.  Such a unit does not wait for the previous sweep to FINISH the patches around it (they
    // are the last ones of that sweep): it follows them up the columns, chunk by chunk.  Lanes 1..3 of the (one-wave)
    // workgroup each watch the counter of one patch of the previous sweep that owns a column within 2H of ours; a chunk
    // that touches oriented F indices <= ilim may run once that patch has published every level <= ilim + (its last
    // column index, halo included) + H + C: then every node the chunk reads is final there, and every read the previous
    // sweep makes of the nodes the chunk writes is over (its chunks are C levels long).
    constexpr bool CHASE_OK = XS && !IS3D && !SKIP && NT == 64;
    const bool chase = CHASE_OK && (dir == 1 || dir == 3);
    const int* chase_ptr = nullptr;
    int chase_add = 0;
    if (CHASE_OK && chase && tid >= 1 && tid <= 3) {
        int ja = j0 - 2 * H, jb = jmaxp + 2 * H;
        ja = ja < 0 ? 0 : ja;
        jb = jb > NJ - 1 ? NJ - 1 : jb;
        const int ja2 = NJ - 1 - jb, jb2 = NJ - 1 - ja;     // the same columns, oriented as in the previous sweep
        const int tja = ja2 / PJ, ntj = jb2 / PJ - tja + 1;
        if (tid - 1 < ntj) {
            const int q = tja + tid - 1;
            const int qmax = (q * PJ + PJ < NJ ? q * PJ + PJ : NJ) - 1;
            chase_ptr = pa.sync + 8 + ((size_t)(dir - 1) * pa.batch + z) * pa.n_patches + q;
            chase_add = qmax + 3 * H + C + 1;
        }
    }
    // wait of a chasing unit before the chunk that starts at level L0 (and the prefetch of the one after it): lane 0 the
    // upwind patch of this sweep (unless first_only), lanes 1..3 the previous sweep.  `sample`: value read ahead of time.
    auto chase_wait = [&](int L0, int sample, bool with_upwind) {
        int ilim = L0 + 2 * C + 2 * H - 1 - j0;
        ilim = ilim > NF - 1 ? NF - 1 : ilim;
        const int* wp = tid == 0 ? (with_upwind ? up_j : nullptr) : chase_ptr;
        const int need = tid == 0 ? L0 + C - 1 : ilim + chase_add;
        bool ok = !wp || sample >= need;
        if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {
            unsigned long long t0 = 0;
            int spins = 0;
            for (;;) {
                if (!ok) ok = ld_prog(wp) >= need;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (spins == 0) t0 = wall_clock64();
                if ((++spins & 63) == 0) {
                    if (__hip_atomic_load(pa.sync + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    if (wall_clock64() - t0 > pa.timeout_ticks) {
                        if (tid == 0) __hip_atomic_store(pa.sync + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(FSM_POLL_SLEEP);
            }
        }
    };
    P* __restrict__ Tg = reinterpret_cast<P*>(a.tt) + (size_t)grp * a.g.n_nodes;
    const T INF = real_traits<T>::inf();
    const P PINF = pack_fill<T, NS>(INF);
    // what the tile holds where the grid has no node.  First-order stage: INF (the minimum of a neighbour pair ignores it).  WENO
    // stage: no out-of-grid entry ever reaches a result (the first / second / last node cases of weno_axis pick their operands
    // inside the grid), so any value will do -- 0 keeps the dropped stencils of boundary nodes tame (weno_axis)
    const P PFILL = H == 2 ? pack_fill<T, NS>((T)0) : PINF;
    const int lm = NS == 1 ? 1 : a.lmask[z];   // sources of the group still being solved
    const int sf = rf ? -1 : 1;

    // march identity of this thread: one column
    const int tj = tid % PJ, tk = tid / PJ;
    const int jp = j0 + tj, kp = k0 + tk;
    const bool col_ok = jp < NJ && kp < NK;
    const int row = IS3D ? (tk + H) * RJ + tj + H : tj + H;
    const int jn = rj ? NJ - 1 - jp : jp;   // natural J / K index of the column
    const int kn = rk ? NK - 1 - kp : kp;
    const uint32_t colbase = ((uint32_t)kn * NJ + jn) * NF;
    const T dx = a.dx, dz = a.dz;
    // WENO stage: correctly rounded 1/(2h), the one true division of the axis derivatives (see div_h2)
    const double r2x = H == 2 ? 1.0 / (2.0 * (double)dx) : 0.0, r2z = H == 2 ? 1.0 / (2.0 * (double)dz) : 0.0;
    const int variant = a.variant;
    const uint32_t* __restrict__ Fz = a.frozen + (size_t)grp * NS * a.mask_words;   // + l * mask_words
    const int* bb = a.bbox + 6 * grp * NS;                                           // + 6 * l
    const int M = a.g.M;
    // sheared slowness copy: uniform base of the unit (the k plane of its first row in the copy's orientation) + this thread's
    // byte offset inside it (32 bits: the host refuses grids whose planes are too far apart) + the row of the level, a scalar
    // (issue_static).  Threads without a column read the nearest one (their values are never used).
    const int skx_min = rev ? NK - 1 - kmaxp_of(k0, NK, PK) : k0;
    uint32_t stoff;
    {
        int kc = kp < NK ? kp : NK - 1, jc = jp < NJ ? jp : NJ - 1;
        const int kx = rev ? NK - 1 - kc : kc, jx = rev ? NJ - 1 - jc : jc;
        stoff = (uint32_t)(((size_t)(kx - skx_min) * shear_plane(a.g) + (IS3D ? (size_t)((jx >> 4) * 32 + (jx & 15)) : (size_t)jx)) * sizeof(T));
    }

    // LDS row (without the level) of the column at patch-relative (cj, ck), halo included
    auto lds_row = [&](int cj, int ck) { return (IS3D ? (ck + H) * RJ + cj + H : cj + H) * RS; };
    // natural row base of the column at oriented (jq, kq)
    auto nat_row = [&](int jq, int kq) {
        return ((uint32_t)(rk ? NK - 1 - kq : kq) * NJ + (rj ? NJ - 1 - jq : jq)) * NF;
    };

    // streaming identity: lane e = tid % C walks levels, rsub = tid / C picks the column of a pass
    const int e = tid % C, rsub = tid / C;
    // static passes: levels L0+H+e of the own and downwind-halo columns  <->  tile q = e + 2H
    //   i' = L - j' - k' = L0 + ipb,  natural index = abase + sf*L0
    int ipb[NOWN + NHI];
    uint32_t abase[NOWN + NHI];
    int lrow[NOWN + NHI];
#pragma unroll
    for (int it = 0; it < NOWN + NHI; ++it) {
        int cj, ck;
        bool ok = true;
        if (it < NOWN) {
            const int rid = rsub + RPI * it;
            cj = rid % PJ;
            ck = rid / PJ;
        } else {
            const int h = rsub + RPI * (it - NOWN);
            ok = h < NDH;
            if (IS3D) {
                if (h < H * PK) { cj = PJ + h / PK; ck = h % PK; } else { cj = (h - H * PK) % PJ; ck = PK + (h - H * PK) / PJ; }
            } else {
                cj = PJ + h; ck = 0;
            }
        }
        const int jq = j0 + cj, kq = k0 + ck;
        ok = ok && jq < NJ && kq < NK;
        const int b = H + e - jq - kq;
        ipb[it] = ok ? b : -(1 << 29);  // fails the range test forever
        abase[it] = nat_row(jq, kq) + (uint32_t)(rf ? NF - 1 - b : b);
        lrow[it] = lds_row(cj, ck) + e + 2 * H;
    }
    // upwind halo passes: levels L0-1+e (tile q = e + H - 1), every level <= L0+C-2 is published
    int uipb[NUPI];
    uint32_t uabase[NUPI];
    int ulrow[NUPI];
#pragma unroll
    for (int it = 0; it < NUPI; ++it) {
        const int h = rsub + RPI * it;
        int cj, ck;
        bool ok = h < NUP;
        if (IS3D) {
            if (h < H * PK) { cj = -1 - h / PK; ck = h % PK; } else { cj = (h - H * PK) % PJ; ck = -1 - (h - H * PK) / PJ; }
        } else {
            cj = -1 - h; ck = 0;
        }
        const int jq = j0 + cj, kq = k0 + ck;
        ok = ok && jq >= 0 && jq < NJ && kq >= 0 && kq < NK;
        const int b = -1 + e - jq - kq;
        uipb[it] = ok ? b : -(1 << 29);
        uabase[it] = nat_row(jq >= 0 ? jq : 0, kq >= 0 ? kq : 0) + (uint32_t)(rf ? NF - 1 - b : b);
        ulrow[it] = (h < NUP) ? lds_row(cj, ck) + e + H - 1 : -1;
    }
    // H == 2 only: the entries the uniform windows miss -- level L0+1 (q = 3) of the downwind
    // columns at distance 1 (read at q+1 by the last column and at q+2 by the one before), and
    // level L0-2 (q = 0) of BOTH upwind columns (column -1 is read at q-2 by the second column)
    int xipb = -(1 << 29), xlrow = -1;
    uint32_t xabase = 0;
    bool x_up = false;
    if (H == 2) {
        constexpr int NX1 = IS3D ? (PJ + PK) : 1;
        static_assert(H != 2 || 3 * NX1 <= NT, "extras fit one pass");
        if (tid < 3 * NX1) {
            const bool up = tid >= NX1;
            const int dist = tid >= 2 * NX1 ? 2 : 1;   // upwind: distance 1 or 2; downwind: 1
            const int h = tid % NX1;
            int cj, ck;
            if (IS3D) {
                if (h < PK) { cj = up ? -dist : PJ; ck = h; } else { cj = h - PK; ck = up ? -dist : PK; }
            } else {
                cj = up ? -dist : PJ; ck = 0;
            }
            const int jq = j0 + cj, kq = k0 + ck;
            const bool ok = jq >= 0 && jq < NJ && kq >= 0 && kq < NK;
            const int lev = up ? -2 : 1;               // level relative to L0
            const int b = lev - jq - kq;
            xipb = ok ? b : -(1 << 29);
            xabase = nat_row(jq >= 0 ? jq : 0, kq >= 0 ? kq : 0) + (uint32_t)(rf ? NF - 1 - b : b);
            xlrow = lds_row(cj, ck) + lev + H;
            x_up = up;
        }
    }

    // chunk starts are congruent to m = TJ+TK modulo C: an upwind patch then finishes exactly the
    // levels we need (<= Lc+C-2) with the chunk it started one level before ours
    const int m = TJ + TK;
    int Lc = Ls - (((Ls - m) % C + C) % C);

    // static inputs of a chunk starting at level L: slowness of the own column (levels L..L+C-1)
    // and the not-yet-swept values at levels L+H..L+C+H-1 of the own and downwind-halo columns
    T sv[C];
    P tv[NOWN + NHI];
    int xs_next = 0, xs_level = -(1 << 30);
    auto issue_static = [&](int L) {
        // row of the sheared copy at level L: x = L mod M in the orientation of the copy's family (the opposite direction walks it
        // backwards) -- the same for every thread of the workgroup, so all of this is scalar arithmetic.  Consecutive chunks
        // continue the walk where the previous call stopped; only a fresh start pays the modulo.  The C rows of the chunk are read
        // upwards from the lowest one, through the repeated rows behind row M - 1 where they would wrap.
        int x;
        if (L == xs_level) {
            x = xs_next;
        } else {
            x = (rev ? NF + NJ + NK - 3 - L : L) % M;   // levels before the column starts give x < 0; keep the walk inside [0, M)
            x = x < 0 ? x + M : x;
        }
        x = __builtin_amdgcn_readfirstlane(x);
        int xlo = rev ? x - (C - 1) : x;
        if (xlo < 0) { xlo %= M; xlo = xlo < 0 ? xlo + M : xlo; }   // (M may be smaller than a chunk)
        // sv[] holds the rows in ascending order (the march reads it backwards for a walk that goes down the copy).  A load is one
        // instruction: the row pointer stays on the scalar side (the empty asm keeps the compiler from folding it into the thread's
        // 64-bit address again), the thread's part is a 32-bit register offset, the half of a row pair an immediate
        constexpr bool paired = IS3D;   // (the host sets SR accordingly: 3-D paired rows, 2-D plain rows)
        const uint32_t SRB = (paired ? (uint32_t)a.g.SR : (uint32_t)NJ) * (uint32_t)sizeof(T);   // bytes from a row (pair) to the next
        const char* rowp = reinterpret_cast<const char*>(Sg + (size_t)skx_min * shear_plane(a.g)) + (size_t)(paired ? (uint32_t)xlo >> 1 : (uint32_t)xlo) * SRB;
        auto ld_row = [&](int half) -> T {
            asm volatile("" : "+s"(rowp));
            return *reinterpret_cast<const T*>(rowp + half * 16 * (int)sizeof(T) + stoff);
        };
        if constexpr (paired) {
            if ((xlo & 1) == 0) {
#pragma unroll
                for (int q = 0; q < C; ++q) {
                    sv[q] = ld_row(q & 1);
                    if (q & 1) rowp += SRB;
                }
            } else {
#pragma unroll
                for (int q = 0; q < C; ++q) {
                    if (!(q & 1)) { sv[q] = ld_row(1); rowp += SRB; } else sv[q] = ld_row(0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < C; ++q) {
                sv[q] = ld_row(0);
                rowp += SRB;
            }
        }
        x = rev ? x - C : x + C;
        if (x < 0 || x >= M) { x %= M; x = x < 0 ? x + M : x; }
        xs_next = x;
        xs_level = L + C;
        const int sL = sf * L;
#pragma unroll
        for (int it = 0; it < NOWN + NHI; ++it) {
            P v = PFILL;
            if ((unsigned)(L + ipb[it]) < (unsigned)NF) v = XS ? ld_sc1(Tg + (abase[it] + sL)) : Tg[abase[it] + sL];
            tv[it] = v;
        }
    };

    // dirty-brick tracking: sigma = global sweep number (1-based).  The local update is a function of the six (2-D: four)
    // neighbours only -- the sweep direction orders the visits, it does not enter the formula -- so a node that was visited
    // in sweep sigma-1 and whose neighbours have not changed since satisfies T <= update(neighbours) already.  A chunk is a
    // no-op when no brick of its read set (own nodes + halo) changed during sweep sigma-1 or so far in this sweep.
    // (SKIP kernels: sweep number sigma = global sweep index + 1; the read set of the unit in bricks, J / K extent -- see
    //  the scheduler set-up below; all of it in LDS)
    unsigned long long nevals = 0;

    T dec[NS];
#pragma unroll
    for (int l = 0; l < NS; ++l) dec[l] = 0;
    // own column carried from the previous chunk: results at levels L0-H..L0-1, then the old values
    // at levels L0..L0+H-1
    P carry[2 * H];
#pragma unroll
    for (int q = 0; q < 2 * H; ++q) carry[q] = PFILL;
    bool have_prev = false;   // carry[] is valid (the previous chunk was evaluated)
    int pref_for = -(1 << 30);  // level for which sv/tv were prefetched
    int pending = 0;            // progress value of the previous chunk, published once its stores have drained
    int n_done = 0;             // chunks of this unit evaluated so far
    int pre_j = -1, pre_k = -1; // upwind progress counters as sampled during the previous chunk (first lane of every wave)
    int smp_j = 0, smp_k = 0;   // ... the sampled words as loaded (epoch field and all)
    P uvn[NUPI];                // upwind halo columns fetched one chunk ahead ...
    int uvn_for = -(1 << 30);   // ... for the chunk that starts at this level
    // Chunks that may hold frozen nodes of source l: the patch overlaps the bounding box of the frozen nodes in
    // J and K, and the chunk start L0 lies in [near_lo, near_hi] (the F extent of a chunk, clamped to the grid,
    // is [L0 - jmaxp - kmaxp, L0 + C-1 - j0 - k0] in oriented i').  Worked out once per unit.
    int near_lo[NS], near_hi[NS];
    {
        const int jlo = rj ? NJ - 1 - jmaxp : j0, jhi = rj ? NJ - 1 - j0 : jmaxp;
        const int klo = rk ? NK - 1 - kmaxp : k0, khi = rk ? NK - 1 - k0 : kmaxp;
#pragma unroll
        for (int l = 0; l < NS; ++l) {
            const int* b6 = bb + 6 * l;
            const int b0 = rf ? NF - 1 - b6[1] : b6[0], b1 = rf ? NF - 1 - b6[0] : b6[1];   // oriented F range of the box
            const bool jk = !(jhi < b6[2] || jlo > b6[3] || khi < b6[4] || klo > b6[5]) && b0 <= NF - 1 && b1 >= 0;
            near_lo[l] = jk ? b0 - (C - 1) + j0 + k0 : 1;
            near_hi[l] = jk ? b1 + jmaxp + kmaxp : 0;
        }
    }
    if (!FSM_EXP_NOWAIT && XS && dir > 0 && !chase) {
        // previous sweep of this iteration: wait for the patches (of ITS oriented partition) that own
        // a column within 2H of ours -- at most 3 x 3 of them, one lane each
        if (tid < 16) {
            const int* pp = prev_sweep_counter(dir, TJ, TK, z, tid);
            if (pp) {
                const unsigned long long t0 = wall_clock64();
                int spins = 0;
                while (ld_prog(pp) < (SKIP ? FSM_FIN : 0x3fffffff)) {
                    if ((++spins & 63) == 0) {
                        if (__hip_atomic_load(pa.sync + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        if (wall_clock64() - t0 > pa.timeout_ticks) {
                            __hip_atomic_store(pa.sync + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        __syncthreads();
    }
    if (CHASE_OK && chase) chase_wait(Lc - C, -1, false);   // (covers the prefetch of the first chunk)
    if (!SKIP) { issue_static(Lc); pref_for = Lc; }
    if constexpr (SKIP) {
        // Slab mask of the unit: F bricks in which some brick of the read set (own + halo columns: <= 3 x 3 bricks in J, K)
        // changed during sweep sigma-1 or already in this one.  The sweeps before this one are over for every column the
        // unit reads (the wait above; one launch per sweep: the previous launch), so their stamps are final; what the
        // upwind patches of THIS sweep change afterwards arrives with their progress values, what the unit changes
        // itself it adds to the mask.  Changes of other nodes of a straddling brick may or may not be seen: they do
        // not matter (not in the read set), seeing them only costs an evaluation.
        const int nsw = (pa.nbf + 31) >> 5;
        for (int w = tid; w < nsw; w += NT) s_slab[w] = 0u;
        for (int w = tid; w < 9 * SLABW; w += NT) s_stamped[w / SLABW][w % SLABW] = 0u;
        if (tid == 0) {
            s_cwi[0] = -1; s_cwi[1] = -1; s_mywi = -1;
            // natural J / K extent of the read set (own + halo columns) in bricks, fixed for the whole unit
            const int ja = j0 - H < 0 ? 0 : j0 - H, jb = jmaxp + H > NJ - 1 ? NJ - 1 : jmaxp + H;
            const int ka = k0 - H < 0 ? 0 : k0 - H, kb = kmaxp + H > NK - 1 ? NK - 1 : kmaxp + H;
            const int jlo = (rj ? NJ - 1 - jb : ja) / FSM_BRICK, jhi = (rj ? NJ - 1 - ja : jb) / FSM_BRICK;
            const int klo = (rk ? NK - 1 - kb : ka) / FSM_BRICK, khi = (rk ? NK - 1 - ka : kb) / FSM_BRICK;
            s_u[U_RSJLO] = jlo; s_u[U_RSNJ] = jhi - jlo + 1;
            s_u[U_RSKLO] = klo; s_u[U_RSNK] = khi - klo + 1;
            const int sg = pa.ndir * pa.iter_ptr[0] + dir + 1;   // this sweep's number, 1-based
            s_u[U_THR] = pa.skip == 2 ? -(1 << 30) : (sg - 1 > 0 ? sg - 1 : 0);   // (skip == 2, tuning: every brick counts as changed)
            s_u[U_LCF] = Lc;
            // first chunk start of the two upwind units (their chunk index = (level - start) / C)
            const int lsj = Ls - PJ, lsk = Ls - PK, mu = TJ + TK - 1;
            s_u[U_UPLCF0] = lsj - (((lsj - mu) % C + C) % C);
            s_u[U_UPLCF1] = lsk - (((lsk - mu) % C + C) % C);
            s_u[U_HIST] = 0; s_u[U_EVER] = 0; s_u[U_UCHG] = 0; s_u[U_PENDV] = 0; s_u[U_LCHG] = 0;
        }
        __syncthreads();
        {
            const int jlo = s_u[U_RSJLO], nj = s_u[U_RSNJ], klo = s_u[U_RSKLO], nk = s_u[U_RSNK], thr = s_u[U_THR];
            const int* __restrict__ stamp = pa.stamp + (size_t)grp * pa.nbf * pa.nbj * pa.nbk;   // one stamp set per group
            const int total = pa.nbf * nj * nk;
            for (int b = tid; b < total; b += NT) {
                const int bf = b % pa.nbf, jk = b / pa.nbf;
                const int bj = jlo + jk % nj, bk = klo + jk / nj;
                const int v = __hip_atomic_load(stamp + ((size_t)bk * pa.nbj + bj) * pa.nbf + bf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= thr) atomicOr(&s_slab[bf >> 5], 1u << (bf & 31));
            }
        }
        __syncthreads();
    }
    FSM_PMARK(5)   // ticket, setup, wait for the previous sweep
    const unsigned long long trace_t1 = (FSM_ENABLE_PROF && a.prof) ? wall_clock64() : 0ull;
    // ---- SKIP kernels: the scheduler -------------------------------------------------------------------------------
    // Thread 0 decides, chunk by chunk, whether a chunk can change anything, from three sources that cost it no extra
    // memory round trip: the slab mask (above), the change flags that ride on the upwind progress values it has to read
    // anyway (the last four chunks; older ones from the upwind unit's change map, one cached word per 32 chunks), and
    // its own changes.  A chunk whose read set holds no change since the unit's columns were last visited is a no-op
    // (DESIGN.md section 4a): the scheduler steps over it and publishes the progress, a run of such chunks costs one pass
    // of a scalar loop, and the workgroup only synchronises where a chunk has to be evaluated.
    // change map of unit (patch index pidx) of this launch, edge e
    auto cmap_of = [&](int pidx, int e) -> unsigned long long* {
        return pa.cmap + ((((size_t)(XS ? dir : 0) * pa.batch + z) * pa.n_patches + (size_t)pidx) * 2 + e) * pa.cw;
    };
    // any brick of the slab mask set within the natural F range [flo, fhi]
    auto slab_any = [&](int flo, int fhi) -> bool {
        const int b0 = flo / FSM_BRICK, b1 = fhi / FSM_BRICK;
        for (int w = b0 >> 5; w <= (b1 >> 5); ++w) {
            const int lo = w == (b0 >> 5) ? (b0 & 31) : 0, hi = w == (b1 >> 5) ? (b1 & 31) : 31;
            const unsigned msk = (hi == 31 ? 0xffffffffu : ((2u << hi) - 1u)) & ~((1u << lo) - 1u);
            if (s_slab[w] & msk) return true;
        }
        return false;
    };
    // natural F range of the nodes of chunk L, `ext` more levels / columns on either side (the read set: ext = H)
    auto chunk_frange = [&](int L, int ext, int& flo, int& fhi) {
        int ia = L - ext - jmaxp - kmaxp, ib = L + C - 1 + ext - j0 - k0;
        ia = ia < 0 ? 0 : ia;
        ib = ib > NF - 1 ? NF - 1 : ib;
        flo = rf ? NF - 1 - ib : ia;
        fhi = rf ? NF - 1 - ia : ib;
    };
    // did upwind unit e (0: J, 1: K) change its edge columns in the chunk(s) chunk `need - C + 1` reads?  v: its progress
    // value, known to cover `need`.  (H = 2 reads one level further back: the chunk before as well.)
    auto up_dirty = [&](int v, int need, int e) -> bool {
        if (v >= FSM_FIN && !((v >> e) & 1)) return false;   // finished, never touched that edge
        for (int p2 = 0; p2 < H; ++p2) {
            const int nd = need - p2 * C;
            const int ulcf = s_u[U_UPLCF0 + e];
            const int ciu = (nd - C - ulcf) / C;
            if (nd - C - ulcf < 0) continue;                 // before the upwind unit's first chunk
            int bit;
            const int d = v >= FSM_FIN ? 4 : ((v >> 8) - nd) / C;
            if (d < 4) {
                bit = (v >> (2 * d + e)) & 1;
            } else {
                const int w = ciu >> 5;
                if (w >= pa.cw) continue;                    // beyond its last chunk
                if (s_cwi[e] != w || s_cwl[e] < nd) {
                    // (the word is read after the progress value that makes its bits final was seen)
                    s_cwl[e] = v >= FSM_FIN ? 0x7fffffff : (v >> 8);
                    const unsigned long long* um = cmap_of(e == 0 ? TK * npj + TJ - 1 : (TK - 1) * npj + TJ, e);
                    s_cwv[e] = ld_cmap(um + w);
                    s_cwi[e] = w;
                }
                bit = (s_cwv[e] >> (ciu & 31)) & 1u;
            }
            if (bit) return true;
        }
        return false;
    };
    // wait until *ptr >= want (progress values only grow); returns the value seen, ok = false: abort / time-out
    auto poll_until = [&](const int* ptr, int want, bool& ok) -> int {
        unsigned long long t0 = 0;
        int spins = 0;
        for (;;) {
            const int v = ld_prog(ptr);
            if (v >= want) {
                if (FSM_ENABLE_PROF == 2 && spins) twait += wall_clock64() - t0;
                return v;
            }
            if (spins == 0) t0 = wall_clock64();
            if ((++spins & 63) == 0) {
                if (__hip_atomic_load(pa.sync + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; return v; }
                if (wall_clock64() - t0 > pa.timeout_ticks) {
                    __hip_atomic_store(pa.sync + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    return v;
                }
            }
            __builtin_amdgcn_s_sleep(FSM_POLL_SLEEP);
        }
    };
    for (; Lc <= Le; Lc += C) {
        if constexpr (PRE) {   // samples taken during the previous chunk (progress only grows: never below what is known already)
            // (taken after anything the scheduler polled in that chunk, so they are the newest values this lane has seen)
            pre_j = dec_prog(smp_j);
            pre_k = dec_prog(smp_k);
        }
        if constexpr (SKIP) {
            FSM_PMARK(0)
            if (tid == 0) {
                int L = Lc, act = 1, h = s_u[U_HIST];
                const int lcf = s_u[U_LCF];
                // the common case in a region that is being worked on: the chunk before this one was evaluated and changed a
                // node.  Its bricks went into the slab mask, and the F range of this chunk (widened by H) overlaps the range of
                // that one: slab_any() below would find them -- the chunk is dirty, only the upwind progress has to be there.
                const bool follow = s_u[U_LCHG] != 0 && L <= Le;
                s_u[U_LCHG] = 0;
                for (;;) {
                    if (L > Le) { act = 0; break; }
                    if (follow) {
                        const int need8 = (L + C - 1) << 8;
                        bool ok = true;
                        if (up_j && pre_j < need8) pre_j = poll_until(up_j, need8, ok);
                        if (up_k && ok && pre_k < need8) pre_k = poll_until(up_k, need8, ok);
                        if (!ok) act = 0;
                        break;
                    }
                    const int need = L + C - 1, need8 = need << 8;
                    int vj = FSM_FIN, vk = FSM_FIN;
                    bool ok = true;
                    if (up_j) { vj = pre_j >= need8 ? pre_j : poll_until(up_j, need8, ok); pre_j = vj; }
                    if (up_k && ok) { vk = pre_k >= need8 ? pre_k : poll_until(up_k, need8, ok); pre_k = vk; }
                    if (!ok) { act = 0; break; }
                    if (L == lcf && vj >= FSM_FIN && !(vj & 1) && vk >= FSM_FIN && !(vk & 2)) {
                        // both upwind units are over and never touched the columns this unit reads: with a clean slab
                        // mask the whole unit is a no-op
                        bool any = false;
                        for (int w = 0; w < ((pa.nbf + 31) >> 5); ++w) any |= s_slab[w] != 0u;
                        if (!any) {
                            st_prog(my_prog, FSM_FIN);
                            L = Le + 1;
                            act = 0;
                            break;
                        }
                    }
                    int flo, fhi;
                    chunk_frange(L, H, flo, fhi);
                    bool dirty = slab_any(flo, fhi);
                    if (!dirty && up_j) dirty = up_dirty(vj, need, 0);
                    if (!dirty && up_k) dirty = up_dirty(vk, need, 1);
                    if (dirty) break;
                    if (pending) { act = 2; break; }   // the stores of the chunk before have to drain before progress moves on
                    L += C;
                    h = (h << 2) & 0xff;
                    if (L > Le && s_u[U_UCHG]) {
                        s_u[U_PENDV] = FSM_FIN | s_u[U_EVER];   // (goes out after the unit's stamps, below the loop)
                    } else {
                        st_prog(my_prog, L > Le ? (FSM_FIN | s_u[U_EVER]) : ((L << 8) | h));
                    }
                }
                s_u[U_HIST] = h;
                s_next = L;
                s_act = act;
            }
            __syncthreads();   // also: every read of the previous chunk's LDS tile is done
            const int act = __builtin_amdgcn_readfirstlane(s_act), Ln = __builtin_amdgcn_readfirstlane(s_next);
            if (Ln != Lc) {
                have_prev = false;
                Lc = Ln;
            }
            if (act == 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) st_prog(my_prog, s_u[U_PENDV]);
                pending = 0;
                Lc -= C;   // (the loop header adds it back: the scheduler runs again from the same chunk)
                continue;
            }
            if (act == 0) break;
        }
        const int L0 = Lc;
        const int eoff = jp + kp - L0;
        const int ea = col_ok ? (eoff > 0 ? eoff : 0) : C;       // active levels e = ea..eb
        const int eb = eoff + NF - 1 < C - 1 ? eoff + NF - 1 : C - 1;
        if constexpr (!SKIP) { FSM_PMARK(0) }

        // (1) wait until both upwind patches have published every level <= L0+C-2
        //     The counters were sampled during the previous chunk's march (see below): when that old sample
        //     already suffices -- the upwind patches are normally ahead -- no load latency is paid here.
        if constexpr (!SKIP) {
        if (CHASE_OK && chase) {
            chase_wait(L0, pre_j, true);
        } else if (tid == 0 && (up_j || up_k) && !((up_j ? pre_j : 0x3fffffff) >= L0 + C - 1 && (up_k ? pre_k : 0x3fffffff) >= L0 + C - 1)) {
            const int need = L0 + C - 1;
            unsigned long long t0 = 0;   // the clock is only read once a poll has failed (the common case: none does)
            int spins = 0;
            for (;;) {
                const int vj = up_j ? ld_prog(up_j) : need;
                const int vk = up_k ? ld_prog(up_k) : need;
                if (vj >= need && vk >= need) break;
                if (spins == 0) t0 = wall_clock64();
                if ((++spins & 63) == 0) {
                    if (__hip_atomic_load(pa.sync + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    if (wall_clock64() - t0 > pa.timeout_ticks) {
                        __hip_atomic_store(pa.sync + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(FSM_POLL_SLEEP);
            }
        }
        __syncthreads();  // also: every read of the previous chunk's LDS tile is done
        }
        if (tid == 0) s_anychg = 0;   // (read last before this barrier; written again only after the staging barrier)
        FSM_PMARK(1)
        if (pref_for != L0) issue_static(L0);
        nevals += (eb >= ea) ? (unsigned)(eb - ea + 1) : 0u;   // per active source, see the end

        // (2) upwind halo columns: fresh from HBM with sc1 loads (bypass L1)
        //     -- unless this wave fetched them one chunk ahead (below), the normal case once a unit runs behind its
        //     upwind neighbours: their latency is then off the path between two chunks
        P uv[NUPI];
        if (PRE && uvn_for == L0) {
#pragma unroll
            for (int it = 0; it < NUPI; ++it) uv[it] = uvn[it];
        } else {
#pragma unroll
            for (int it = 0; it < NUPI; ++it) {
                P v = PFILL;
                if ((unsigned)(L0 + uipb[it]) < (unsigned)NF) v = ld_sc1(Tg + (uabase[it] + sf * L0));
                uv[it] = v;
            }
        }
        P xv = PFILL;
        if (H == 2 && (unsigned)(L0 + xipb) < (unsigned)NF) {
            const P* src = Tg + (xabase + sf * L0);
            if (x_up || XS) xv = ld_sc1(src); else xv = *src;
        }
        // (3) static part (prefetched) into LDS: levels L0+H..L0+C+H-1 of own + downwind halo columns;
        //     own column q < 2H comes from the previous chunk (H results, H old values)
#pragma unroll
        for (int it = 0; it < NOWN + NHI; ++it)
            if (it < NOWN || rsub + RPI * (it - NOWN) < NDH) Tt[lrow[it]] = tv[it];
        if (!have_prev) {
            // first chunk of the patch, or the previous chunk was skipped: levels L0-H..L0+H-1 of the
            // own column come from HBM (they are current: a skipped chunk changes nothing)
#pragma unroll
            for (int q = 0; q < 2 * H; ++q) {
                const int ip = L0 - H + q - jp - kp;
                P v = PFILL;
                if (col_ok && (unsigned)ip < (unsigned)NF) {
                    const P* src = Tg + (colbase + (rf ? NF - 1 - ip : ip));
                    v = XS ? ld_sc1(src) : *src;
                }
                carry[q] = v;
            }
            have_prev = true;
        }
#pragma unroll
        for (int q = 0; q < 2 * H; ++q) Tt[row * RS + q] = carry[q];
        // the halo loads have to land anyway: the same wait drains the write-back of the previous chunk,
        // whose progress value goes out right after the barrier (store drain and halo latency overlap)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < NUPI; ++it)
            if (ulrow[it] >= 0) Tt[ulrow[it]] = uv[it];
        if (H == 2 && xlrow >= 0) Tt[xlrow] = xv;
        __syncthreads();
        if (tid == 0 && pending) st_prog(my_prog, SKIP ? s_u[U_PENDV] : pending);
        pending = 0;
        // sample the upwind counters for the NEXT chunk now: the loads complete during the march (counters only
        // grow, an old sample is a safe lower bound)
        if (PRE && Lc + C <= Le) {
            // upwind halo of the NEXT chunk (levels L0+C-1 .. L0+2C-2), when the counters sampled for THIS chunk already
            // cover it (every wave decides from the sample of its own first lane; a wave that cannot yet loads at the top
            // of the next chunk as before)
            const int need_n = SKIP ? (L0 + 2 * C - 1) << 8 : L0 + 2 * C - 1;   // (SKIP: progress values carry 8 flag bits)
            const int sj = up_j ? __builtin_amdgcn_readfirstlane(pre_j) : 0x7fffffff;
            const int sk = up_k ? __builtin_amdgcn_readfirstlane(pre_k) : 0x7fffffff;
            if (sj >= need_n && sk >= need_n) {
#pragma unroll
                for (int it = 0; it < NUPI; ++it) {
                    P v = PFILL;
                    if ((unsigned)(L0 + C + uipb[it]) < (unsigned)NF) v = ld_sc1(Tg + (uabase[it] + sf * (L0 + C)));
                    uvn[it] = v;
                }
                uvn_for = L0 + C;
            }
        }
        if (PRE && (tid & 63) == 0 && Lc + C <= Le) {
            // (raw words: they are decoded at the top of the next chunk, when the loads have long landed)
            if (up_j) smp_j = ld_raw(up_j);
            if (up_k) smp_k = ld_raw(up_k);
        }
        if (PRE && CHASE_OK && chase_ptr && Lc + C <= Le)   // lanes 1..3 of a chasing unit: their patch of the previous sweep
            smp_j = ld_raw(chase_ptr);

        bool near_src[NS];
#pragma unroll
        for (int l = 0; l < NS; ++l) near_src[l] = L0 >= near_lo[l] && L0 <= near_hi[l];

        P own[NQ];
#pragma unroll
        for (int q = 0; q < 2 * H; ++q) own[q] = carry[q];
#pragma unroll
        for (int q = 2 * H; q < NQ; ++q) own[q] = Tt[row * RS + q];
        T sc[C];
#pragma unroll
        for (int q = 0; q < C; ++q) sc[q] = rev ? sv[C - 1 - q] : sv[q];   // (issue_static: rows in ascending order)
        FSM_PMARK(2)

        // (4) prefetch the next chunk's static inputs; they land during the march
        if (Lc + C <= Le) { issue_static(Lc + C); pref_for = Lc + C; }

        // 2-D first-order patches are ONE wavefront: the J neighbours are the adjacent lanes, so the march
        // exchanges values with DPP wave shifts instead of an LDS write -> read round trip per level; the
        // two halo columns are static within the chunk and sit in registers (used by the end lanes only)
        constexpr bool WAVE2D = !IS3D && NT == 64 && H == 1;
        P hup[WAVE2D ? C : 1], hdn[WAVE2D ? C : 1];
        if (WAVE2D) {
#pragma unroll
            for (int ee = 0; ee < C; ++ee) {
                hup[ee] = Tt[(H - 1) * RS + ee + H - 1];        // column -1 at tile q-1
                hdn[ee] = Tt[(PJ + H) * RS + ee + H + 1];       // column PJ at tile q+1
            }
        }
        bool changed = false;
        // A lone wavefront issues in order, so where the march is latency bound (the one-wave 2-D patches) its
        // length IS the instruction count.  The 2-D kernels carry specialised copies of the level march
        // (fsm_march_levels.inc) for wavefronts whose lanes all have a node at every level of the chunk, far
        // from the frozen nodes (almost every chunk): no grid masks, no frozen-bit path, local solver fixed.
        // (Not for the 3-D kernels: the extra registers cost a resident wave per SIMD there -- paired: -22 %,
        // unpaired 512^3: -2 %.)
        constexpr bool DUP = H == 1 && !IS3D;
        if constexpr (!DUP) {
#define FSM_SIMPLE false
#define FSM_VAR 0
#include "fsm_march_levels.inc"
#undef FSM_SIMPLE
#undef FSM_VAR
        } else {
            bool near_any = false;
#pragma unroll
            for (int l = 0; l < NS; ++l) near_any |= near_src[l];
            const bool full_wave = !near_any && __builtin_amdgcn_ballot_w64(ea == 0 && eb == C - 1) == ~0ull;
            if (!full_wave) {
#define FSM_SIMPLE false
#define FSM_VAR 0
#include "fsm_march_levels.inc"
#undef FSM_SIMPLE
#undef FSM_VAR
            } else if (variant == 1) {
#define FSM_SIMPLE true
#define FSM_VAR 1
#include "fsm_march_levels.inc"
#undef FSM_SIMPLE
#undef FSM_VAR
            } else {
#define FSM_SIMPLE true
#define FSM_VAR 2
#include "fsm_march_levels.inc"
#undef FSM_SIMPLE
#undef FSM_VAR
            }
        }
#pragma unroll
        for (int q = 0; q < 2 * H; ++q) carry[q] = own[C + q];
        FSM_PMARK(3)
        if (FSM_ENABLE_PROF) ++pchunks;

        // (5) write back levels L0..L0+C-1 (tile q = H..H+C-1); the columns a downstream patch reads
        //     go out write-through (sc1)
        // block-wide "did anything change": one LDS flag (cleared before the staging barrier), a ballot per wave.
        // (__syncthreads_or is a library routine with a dispatch-packet load, DPP and LDS reductions and two
        // barriers: measurable at once per chunk.)
        // SKIP kernels: bit 1 / bit 2 = a column within H of the patch's downwind J / K edge changed; and the bricks a
        // thread changed are flagged for the stamps (a thread's C nodes lie in at most two bricks along F)
        if constexpr (SKIP) {
            const bool ej = changed && jp + H > jmaxp, ek = IS3D && changed && kp + H > kmaxp;
            const int f = (wave_any(changed) ? 1 : 0) | (wave_any(ej) ? 2 : 0) | (wave_any(ek) ? 4 : 0);
            if (f && (tid & 63) == 0) atomicOr(&s_anychg, f);
            if (!(FSM_SKIP_ABL & 2) && changed) {
                // note the bricks this thread changed (its nodes of the chunk lie in at most two bricks along F): one bit per
                // brick of the unit's read set; they become stamps when the unit is over (the units that read them -- the
                // next sweep's -- wait for that; this sweep's units get the news with the progress values)
                const int ip0 = L0 - jp - kp;                                  // i' of this thread's node at e = 0
                const int ia = ip0 + ea, ib = ip0 + eb;                        // its nodes in the chunk (ea <= eb: it changed one)
                const int ba = (rf ? NF - 1 - ib : ia) / FSM_BRICK, bb2 = (rf ? NF - 1 - ia : ib) / FSM_BRICK;
                const int my_bj = (rj ? NJ - 1 - jp : jp) / FSM_BRICK - s_u[U_RSJLO], my_bk = (rk ? NK - 1 - kp : kp) / FSM_BRICK - s_u[U_RSKLO];
                const int jk = my_bk * s_u[U_RSNJ] + my_bj;
                atomicOr(&s_stamped[jk][ba >> 5], 1u << (ba & 31));
                if (bb2 != ba) atomicOr(&s_stamped[jk][bb2 >> 5], 1u << (bb2 & 31));
            }
        } else {
            if (wave_any(changed) && (tid & 63) == 0) s_anychg = 1;
        }
        __syncthreads();
        const int chg_flags = SKIP ? __builtin_amdgcn_readfirstlane(s_anychg) : s_anychg;
        const bool any_changed = chg_flags != 0;
        if (any_changed) {
#pragma unroll
            for (int it = 0; it < NOWN; ++it) {
                // same streaming map, H levels earlier: level L0+e  <->  i' = L0 + ipb - H
                const int ipm = L0 + ipb[it] - H;
                if ((unsigned)ipm < (unsigned)NF) {
                    const int rid = rsub + RPI * it;
                    const int cj = rid % PJ, ck = rid / PJ;
                    P* dst = Tg + (abase[it] + sf * (L0 - H));
                    const P v = Tt[lrow[it] - H];
                    if (XS || cj >= PJ - H || (IS3D && ck >= PK - H)) st_sc1(dst, v); else *dst = v;
                }
            }
        }
        // (5b) the unit's own changes: slab mask, change map, flags of the progress word
        if constexpr (SKIP) {
            if (!(FSM_SKIP_ABL & 4) && tid == 0) {
                if (any_changed) {
                    // own changes into the slab mask (the next chunks read these bricks) and into the unit's change map
                    int flo, fhi;
                    chunk_frange(L0, 0, flo, fhi);
                    for (int b = flo / FSM_BRICK; b <= fhi / FSM_BRICK; ++b) s_slab[b >> 5] |= 1u << (b & 31);
                    const int ci = (L0 - s_u[U_LCF]) / C, w = ci >> 5;
                    if ((chg_flags & 6) && w < pa.cw) {
                        if (s_mywi != w) { s_mywi = w; s_mywv[0] = 0u; s_mywv[1] = 0u; }
                        for (int e2 = 0; e2 < 2; ++e2)
                            if (chg_flags & (2 << e2)) {
                                s_mywv[e2] |= 1u << (ci & 31);
                                st_cmap(cmap_of(TK * npj + TJ, e2) + w, s_mywv[e2]);
                            }
                    }
                    s_u[U_UCHG] = 1;
                    s_u[U_LCHG] = 1;
                }
                const int h2 = ((s_u[U_HIST] << 2) | (chg_flags >> 1)) & 0xff, ev2 = s_u[U_EVER] | (chg_flags >> 1);
                s_u[U_HIST] = h2;
                s_u[U_EVER] = ev2;
                s_u[U_PENDV] = Lc + C > Le ? (FSM_FIN | ev2) : (((Lc + C) << 8) | h2);
            }
        }
        // (6) publish later: the counter moves once every wave has drained these stores -- at the
        //     staging barrier of the next chunk, or right after the loop
        if constexpr (SKIP) pending = 1;   // (the value is in s_u[U_PENDV])
        else pending = Lc + C > Le ? 0x3fffffff : Lc + C;
        if (FSM_EARLY_PUB > 0 && n_done < FSM_EARLY_PUB) {
            // the first chunks of a unit sit on the ramp of the patch wavefront (the downstream patches are waiting for
            // exactly these levels): publish at once instead of at the next staging barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st_prog(my_prog, SKIP ? s_u[U_PENDV] : pending);
            pending = 0;
        }
        ++n_done;
        FSM_PMARK(4)
    }
    if constexpr (SKIP) {
        // stamp the bricks the unit changed with this sweep's number -- before the final progress value goes out: the units
        // of the next sweep wait for that value and then read the stamps
        __syncthreads();
        if (s_u[U_UCHG]) {
            const int nsw = (pa.nbf + 31) >> 5;
            const int rs_jlo = s_u[U_RSJLO], rs_nj = s_u[U_RSNJ], rs_klo = s_u[U_RSKLO], njk = rs_nj * s_u[U_RSNK];
            int* __restrict__ stamp = pa.stamp + (size_t)grp * pa.nbf * pa.nbj * pa.nbk;
            const int sigma = pa.ndir * pa.iter_ptr[0] + dir + 1;
            for (int q = tid; q < njk * nsw * 32; q += NT) {
                const int jk = q / (nsw * 32), bf = q % (nsw * 32);
                if (bf < pa.nbf && ((s_stamped[jk][bf >> 5] >> (bf & 31)) & 1u))
                    atomicMax(stamp + ((size_t)(rs_klo + jk / rs_nj) * pa.nbj + rs_jlo + jk % rs_nj) * pa.nbf + bf, sigma);
            }
            pending = 1;   // (the drain below covers the atomics; the value to publish is already in s_u[U_PENDV])
        }
    }
    if (pending) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) st_prog(my_prog, SKIP ? s_u[U_PENDV] : pending);
    }
    if constexpr (SKIP) {   // this sweep's tally for the whole-sweep shortcut of the next one
        const int sigma0 = pa.ndir * pa.iter_ptr[0] + dir;
        if (tid == 0 && pa.sw && sigma0 < pa.n_sw_sweeps)
            atomicAdd(pa.sw + (size_t)sigma0 * pa.n_sw_groups + grp, 1ull | (s_u[U_UCHG] ? (1ull << 32) : 0ull));
    }

    if (FSM_ENABLE_PROF && a.prof && tid == 0) {
        if constexpr (FSM_ENABLE_PROF == 2) {
            for (int q = 0; q < 6; ++q) atomicAdd(a.prof + q, s_pacc[q]);
            atomicAdd(a.prof + 6, 1ull);
            atomicAdd(a.prof + 7, (unsigned long long)pchunks);
        }
        if (FSM_ENABLE_PROF == 1) {
#pragma unroll
            for (int q = 0; q < 6; ++q) atomicAdd(a.prof + q, pacc[q]);
            atomicAdd(a.prof + 6, 1ull);
            atomicAdd(a.prof + 7, (unsigned long long)pchunks);
        }
        // per-unit trace (last iteration wins): entry, first chunk, exit on the 100 MHz clock; patch and direction
        // (trace buffer: 65536 unit entries of 4 words, then 80 chunk records of 5 stamps for each of the first 8192 units)
        if (ticket < FSM_TRACE_UNITS) {
            unsigned long long* tr = a.prof + 8 + 4 * (size_t)ticket;
            // (trace-only builds: the second word holds first chunk - entry in its low half, the polling ticks in its high half)
            tr[0] = trace_t0; tr[1] = FSM_ENABLE_PROF == 2 ? ((trace_t1 - trace_t0) & 0xffffffffull) | (twait << 32) : trace_t1; tr[2] = wall_clock64();
            tr[3] = (unsigned long long)TJ | ((unsigned long long)TK << 16) | ((unsigned long long)dir << 32) | ((unsigned long long)z << 40) |
                    ((unsigned long long)pchunks << 48);
        }
    }
#undef FSM_PMARK
    // L1 decrease of every source of the unit: wavefront reduction, one atomic per wave and source
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nevals += __shfl_down(nevals, off, 64);
#pragma unroll
    for (int l = 0; l < NS; ++l) {
        double accd = (double)dec[l];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) accd += __shfl_down(accd, off, 64);
        if ((tid & 63) == 0 && accd != 0.0) atomicAdd(a.change + grp * NS + l, accd);
        if ((tid & 63) == 0 && nevals && ((lm >> l) & 1)) atomicAdd(pa.evals + grp * NS + l, nevals);
    }
    return true;
}

// The sweep kernel.  First-order 3-D instantiations: a workgroup takes work units (tickets) until none is left.  Units used
// to be one workgroup each; the dispatcher then had to place a new workgroup whenever one retired -- in launch order,
// round-robin over the XCDs -- and with units of very different length (exact skipping: 10 us to 700 us) 10-18 % of the
// workgroup slots stood empty (unit trace, profiles/r03/unit_trace.txt).  Any ticket order that is a valid order for fresh
// workgroups is one for resident ones: a workgroup only waits for units with lower tickets, and those are running or done.
// The WENO stage and the 2-D kernels keep one workgroup per unit: their units are long and alike, and the loop costs them
// registers (WENO, one 256^3 source: 365 -> 406 ms looped; 2-D 4096^2 x 64: 17.6 -> 18.2 ms).
template <typename T, int PJ, int PK, int C, bool IS3D, bool SKIP, int H, int NS, bool XS, bool PRE = false, int AR = 0>
#ifndef FSM_WENO_MINW
#define FSM_WENO_MINW FSM_MINW   // resident workgroups per SIMD asked of the fp32 3-D WENO kernel (one field per workgroup)
#endif
__global__ __launch_bounds__(PJ* PK, (SKIP && IS3D && H == 1 && NS == 2 && sizeof(T) == 4) ? 3
                                     : (IS3D && H == 2 && NS == 1 && sizeof(T) == 4) ? FSM_WENO_MINW : FSM_MINW) void fsm_sweep_persistent(const PersistArgs<T> pa) {
    if constexpr (fsm_looped(IS3D, H)) {
        // Nothing is to be carried from one unit to the next: the arguments are read again from the kernel argument segment
        // through a pointer the compiler cannot see through (it would otherwise hoist everything that depends on them out of
        // the loop and keep it in registers -- the march has none to spare), and so is the thread index inside the unit.
        (void)pa;
        for (;;) {
            auto kp = __builtin_amdgcn_kernarg_segment_ptr();   // (constant address space)
            asm volatile("" : "+s"(kp));
            // (cast to a generic pointer in sight of the compiler: it still knows the loads are scalar loads of constant memory)
            if (!fsm_sweep_unit<T, PJ, PK, C, IS3D, SKIP, H, NS, XS, PRE, AR>(*(const PersistArgs<T>*)kp)) break;
        }
    } else {
        fsm_sweep_unit<T, PJ, PK, C, IS3D, SKIP, H, NS, XS, PRE, AR>(pa);
    }
}

// ---- rotated template (2-D, rotated_template=true): Grid2Drn::sweep45, ttcr/Grid2Drn.h:756-794 ----
// The stencil uses the four diagonal neighbours, so a node depends on the WHOLE previous row of the
// outer loop (x'-1, z'-1 and z'+1): the level function of this ordering is L = 2 x' + z' (the two
// upwind neighbours sit at L-3 and L-1, the two downwind ones at L+1 and L+3).  An optional stage of
// the reference (off by default): one workgroup per source (sources run concurrently on different
// CUs; a single source is latency bound).
template <typename T>
struct Sweep45Args {
    T* tt;                    // [group][node][ts]
    const T* s;               // node slowness, natural layout (z fastest)
    const uint32_t* frozen;   // [n_slots][mask_words]
    double* change;           // [n_slots]
    const int* slots;         // [batch] slot, or slot group when by_group
    const int* lmask;         // [batch] by_group: sources of the group still being solved
    int ts, by_group;
    int nnx, nnz;
    size_t n_nodes;
    uint32_t mask_words;
    T dx;
};

// One workgroup per source; ring row r <-> thread r.  The columns x' (outer loop index of the
// reference) are taken in strips of NT-2: a row only depends on the row before it, so a strip can
// run through all its levels once the strip before it is complete.  Rows 0 and NT-1 of the ring are
// the halo columns (finished values of the previous strip / untouched values of the next one).
// Per column the ring holds 16 consecutive levels; chunks of 8 levels: load levels L0+4..L0+11 of
// every row (the node of row r at level L is z' = L - 2(r-1)), then 8 LDS-only levels.
template <typename T, int NT>
__global__ __launch_bounds__(NT) void fsm_sweep45(const Sweep45Args<T> a) {
    constexpr int W = 16, C = 8;
    __shared__ T ring[NT * W];
    int slot;
    if (a.by_group) {
        const int z = blockIdx.x / a.ts, l = blockIdx.x % a.ts;
        const int grp = a.slots[z];
        if (grp < 0 || !((a.lmask[z] >> l) & 1)) return;
        slot = grp * a.ts + l;
    } else {
        slot = a.slots[blockIdx.x];
        if (slot < 0) return;
    }
    T* __restrict__ Tg = a.tt + ((size_t)(slot / a.ts) * a.n_nodes * a.ts + slot % a.ts);
    const uint32_t* __restrict__ Fz = a.frozen + (size_t)slot * a.mask_words;
    const int nnx = a.nnx, nnz = a.nnz, ts = a.ts;
    const T TMAX = real_traits<T>::max();
    const int r = threadIdx.x;
    constexpr int S = NT - 2;
    T dec = 0;
    for (int dir = 0; dir < 4; ++dir) {
        const int ri = (dir == 1) | (dir == 2), rj = dir >> 1;
        for (int x0 = 0; x0 < nnx; x0 += S) {
            const int ncol = nnx - x0 < S ? nnx - x0 : S;
            const int nlev = 2 * (ncol - 1) + nnz;     // local levels L = 2 (r-1) + z'
            const int xp = x0 + r - 1;                 // oriented column of this ring row
            const bool col_in = xp >= 0 && xp < nnx && r <= ncol + 1;
            const bool own = r >= 1 && r <= ncol;
            const int i = ri ? nnx - 1 - xp : xp;
            const uint32_t rowbase = (uint32_t)(col_in ? i : 0) * nnz;
            // levels [lo, lo+C) of this row, old values (+ slowness / frozen bits when they are own nodes)
            auto load_levels = [&](int lo, T* v) {
#pragma unroll
                for (int e = 0; e < C; ++e) {
                    const int zp = lo + e - 2 * (r - 1);
                    const bool ok = col_in && zp >= 0 && zp < nnz;
                    const uint32_t n = rowbase + (rj ? nnz - 1 - zp : zp);
                    v[e] = ok ? ld_sc1(Tg + (size_t)n * ts) : TMAX;
                }
            };
            auto load_static = [&](int lo, T* sv, unsigned& fz) {
                fz = 0;
#pragma unroll
                for (int e = 0; e < C; ++e) {
                    const int zp = lo + e - 2 * (r - 1);
                    const bool ok = own && zp >= 0 && zp < nnz;
                    const uint32_t n = rowbase + (rj ? nnz - 1 - zp : zp);
                    sv[e] = ok ? a.s[n] : (T)0;
                    const bool frozen = ok ? ((Fz[n >> 5] >> (n & 31)) & 1u) : true;
                    fz |= (frozen ? 1u : 0u) << e;
                }
            };
            auto to_ring = [&](int lo, const T* v) {
#pragma unroll
                for (int e = 0; e < C; ++e) ring[r * W + ((lo + e) & (W - 1))] = v[e];
            };
            T vn[C], svn[C];
            unsigned fzn;
            load_levels(-4, vn);
            to_ring(-4, vn);
            load_levels(4, vn);
            load_static(0, svn, fzn);
            for (int L0 = 0; L0 < nlev; L0 += C) {
                T sv[C];
#pragma unroll
                for (int e = 0; e < C; ++e) sv[e] = svn[e];
                const unsigned fz = fzn;
                to_ring(L0 + 4, vn);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                // next chunk's inputs: issued now, they land during the LDS-only level march
                if (L0 + C < nlev) {
                    load_levels(L0 + C + 4, vn);
                    load_static(L0 + C, svn, fzn);
                }
                T res[C];
                unsigned chg = 0;
#pragma unroll
                for (int e = 0; e < C; ++e) {
                    const int L = L0 + e;
                    if (!((fz >> e) & 1u)) {
                        const T tmm = ring[(r - 1) * W + ((L - 3) & (W - 1))];   // (x'-1, z'-1)
                        const T tmp = ring[(r - 1) * W + ((L - 1) & (W - 1))];   // (x'-1, z'+1)
                        const T tpm = ring[(r + 1) * W + ((L + 1) & (W - 1))];   // (x'+1, z'-1)
                        const T tpp = ring[(r + 1) * W + ((L + 3) & (W - 1))];   // (x'+1, z'+1)
                        const T c = ring[r * W + (L & (W - 1))];
                        // natural diagonals: a = min over (+1,+1)/(-1,-1), b = min over (+1,-1)/(-1,+1);
                        // flipping exactly one axis swaps the two diagonals
                        const bool sw = ri != rj;
                        const T a1 = sw ? tpm : tpp, a2 = sw ? tmp : tmm;
                        const T b1 = sw ? tpp : tpm, b2 = sw ? tmm : tmp;
                        const T av = a1 < a2 ? a1 : a2, bv = b1 < b2 ? b1 : b2;
                        const T t = update2_fh(av, bv, fh45(sv[e], a.dx));
                        if (t < c) {
                            ring[r * W + (L & (W - 1))] = t;
                            res[e] = t;
                            chg |= 1u << e;
                            dec += c - t;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                // write back what changed; nobody reads these nodes from memory before the strip ends
#pragma unroll
                for (int e = 0; e < C; ++e) {
                    if ((chg >> e) & 1u) {
                        const int zp = L0 + e - 2 * (r - 1);
                        st_sc1(Tg + (size_t)(rowbase + (rj ? nnz - 1 - zp : zp)) * ts, res[e]);
                    }
                }
            }
            // the next strip (and the next direction) reads what this one stored
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    double accd = (double)dec;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) accd += __shfl_down(accd, off, 64);
    if ((threadIdx.x & 63) == 0 && accd != 0.0) atomicAdd(a.change + slot, accd);
}

// Row-parallel form of the same sweep.  update_node45 reads rows i-1 and i+1 only -- never its own row --
// so inside one pass of the outer loop the nodes of a row do not depend on each other: the Gauss-Seidel
// order of sweep45 is "rows in sequence, any order within a row" (the inner-loop direction of the four
// sweeps is immaterial).  One workgroup per source walks the rows with four row buffers in LDS: A = the row
// before (already updated), B = this row (updated in place), C = the next row (old values), D = being
// filled with the row after next.  One barrier per row; global accesses are whole rows (z is contiguous).
template <typename T>
__global__ __launch_bounds__(1024) void fsm_sweep45_rows(const Sweep45Args<T> a) {
    extern __shared__ unsigned char fsm_smem45[];
    int slot;
    if (a.by_group) {
        const int z = blockIdx.x / a.ts, l = blockIdx.x % a.ts;
        const int grp = a.slots[z];
        if (grp < 0 || !((a.lmask[z] >> l) & 1)) return;
        slot = grp * a.ts + l;
    } else {
        slot = a.slots[blockIdx.x];
        if (slot < 0) return;
    }
    T* __restrict__ Tg = a.tt + ((size_t)(slot / a.ts) * a.n_nodes * a.ts + slot % a.ts);
    const uint32_t* __restrict__ Fz = a.frozen + (size_t)slot * a.mask_words;
    const int nnx = a.nnx, nnz = a.nnz, ts = a.ts;
    const int W = nnz + 2;                       // one padding entry (max()) at either end of a row
    T* buf = reinterpret_cast<T*>(fsm_smem45);
    const T TMAX = real_traits<T>::max();
    const int tid = threadIdx.x, NT = blockDim.x;
    auto load_row = [&](int b, int i) {          // row i of the field (or max() outside the grid) into buffer b
        T* dst = buf + (size_t)b * W;
        if (i < 0 || i >= nnx) {
            for (int j = tid; j < W; j += NT) dst[j] = TMAX;
        } else {
            const T* src = Tg + (size_t)i * nnz * ts;
            for (int j = tid; j < nnz; j += NT) dst[j + 1] = ld_sc1(src + (size_t)j * ts);
            if (tid == 0) { dst[0] = TMAX; dst[W - 1] = TMAX; }
        }
    };
    T dec = 0;
    for (int dir = 0; dir < 4; ++dir) {
        const int di = (dir == 1 || dir == 2) ? -1 : 1;          // outer loop direction (ttcr/Grid2Drn.h:760-793)
        int i = di > 0 ? 0 : nnx - 1;
        int bA = 0, bB = 1, bC = 2, bD = 3;
        load_row(bA, -1);
        load_row(bB, i);
        load_row(bC, i + di);
        __syncthreads();
        for (int r = 0; r < nnx; ++r, i += di) {
            load_row(bD, i + 2 * di);
            const T* __restrict__ A = buf + (size_t)bA * W;
            T* __restrict__ B = buf + (size_t)bB * W;
            const T* __restrict__ Cc = buf + (size_t)bC * W;
            for (int j = tid; j < nnz; j += NT) {
                const uint32_t n = (uint32_t)i * nnz + j;
                if ((Fz[n >> 5] >> (n & 31)) & 1u) continue;
                // the two diagonals: (next row, j+1) with (previous row, j-1), and (next row, j-1) with (previous
                // row, j+1); which of them the reference calls a and b depends on the direction, the solver is symmetric
                const T d1 = Cc[j + 2] < A[j] ? Cc[j + 2] : A[j];
                const T d2 = Cc[j] < A[j + 2] ? Cc[j] : A[j + 2];
                const T c = B[j + 1];
                const T t = update2_fh(d1, d2, fh45(a.s[n], a.dx));
                if (t < c) {
                    B[j + 1] = t;
                    st_sc1(Tg + (size_t)n * ts, t);
                    dec += c - t;
                }
            }
            __syncthreads();
            const int o = bA; bA = bB; bB = bC; bC = bD; bD = o;
        }
        // the next sweep reads rows this one stored
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    double accd = (double)dec;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) accd += __shfl_down(accd, off, 64);
    if ((threadIdx.x & 63) == 0 && accd != 0.0) atomicAdd(a.change + slot, accd);
}

// ---- small kernels -------------------------------------------------------------------------

template <typename T>
__global__ void fsm_fill(T* p, size_t n, T v, int stride) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i * stride] = v;
}

// The reference's stopping rule as it stands (ttcr/Grid3Drnfs.h:141-152, ttcr/Grid2Drnfs.h:265-290): change = sum over the nodes,
// in node order, in T1, of abs(times[n] - T[n]) with `times` the field before the sweep-iteration.  A sequential sum in the grid's
// own precision is not the fp64 sum the sweep kernels accumulate (at 1.3e8 fp32 nodes small addends vanish below half an ulp of
// the accumulator or round up to a whole one); where the two could fall on different sides of eps * N the host asks for this
// kernel: one workgroup per field, old[] = a snapshot taken before the iteration.  The adds are done in node order by ONE wavefront
// (lane-uniform arithmetic); adding 0 does not change an IEEE sum, so only the nodes that changed are visited -- the other waves of
// the workgroup stream the two fields through LDS a tile ahead and mark the 64-node blocks that hold a change.
template <typename T>
struct RefChangeArgs {
    const T* cur;        // [n_fields][node][stride] current fields (entry b: cur + off[b])
    const T* old;        // the snapshots, same layout
    const size_t* off;   // element offset of field b in both
    T* out;              // [n_fields] the reference's `change`
    size_t n_nodes;
    int stride;
};
template <typename T>
__global__ __launch_bounds__(256) void fsm_reference_change(const RefChangeArgs<T> a) {
    constexpr int TILE = 4096;                 // nodes per tile: 16 rounds of 256
    __shared__ T dbuf[2][TILE];
    __shared__ unsigned long long mask[2][TILE / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const T* __restrict__ cur = a.cur + a.off[blockIdx.x];
    const T* __restrict__ old = a.old + a.off[blockIdx.x];
    const size_t n = a.n_nodes;
    const size_t ntiles = (n + TILE - 1) / TILE;
    T acc = 0;
    auto fill = [&](size_t t, int buf) {
#pragma unroll 4
        for (int r = 0; r < TILE / 256; ++r) {
            const size_t i = t * TILE + (size_t)r * 256 + tid;
            T d = 0;
            if (i < n) {
                const T df = old[i * a.stride] - cur[i * a.stride];   // times[n] - T[n], in T1
                d = df < 0 ? -df : df;
            }
            dbuf[buf][r * 256 + tid] = d;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(d != (T)0);
            if (lane == 0) mask[buf][r * 4 + wave] = m;
        }
    };
    fill(0, 0);
    __syncthreads();
    for (size_t t = 0; t < ntiles; ++t) {
        const int buf = (int)(t & 1);
        // every wave helps to fill the next tile; wave 0 then walks this one
        if (t + 1 < ntiles) fill(t + 1, buf ^ 1);
        if (wave == 0) {
            for (int blk = 0; blk < TILE / 64; ++blk) {
                unsigned long long m = mask[buf][blk];
                if (m == 0ull) continue;
                const T d = dbuf[buf][blk * 64 + lane];
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    const T v = __shfl(d, l, 64);   // (lane-uniform: every lane carries the same accumulator)
                    acc = acc + v;
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) a.out[blockIdx.x] = acc;
}

// ---- the same sum, computed in parallel -- exactly -------------------------------------------------------------------------
// s_{i+1} = RN(s_i + x_i) with x_i = |times[i] - T[i]| >= 0 is sequential, but while the running sum stays in one binade its ulp u is
// fixed and every s_i is a multiple S_i u of it, so
//     RN(s_i + x_i) = (S_i + d_i) u,   d_i = round-to-nearest(x_i / u), a tie (fraction exactly 1/2) resolved so that S_i + d_i is even:
// the increment depends on the state only through the PARITY of S_i, and only at ties.  An element is therefore a map
// parity -> (increment, parity'), maps compose associatively, and a block of elements is summarised by two integers (its increment
// for either parity coming in): blocks are summarised in parallel and the summaries composed in order.  The recurrence changes where
// the sum reaches the next binade (at most a few dozen times per field: the sum only grows); the element that takes it there is added
// with the hardware's own T addition and the scan goes on from the element behind it with the new unit.  Exact for float and double
// (integer arithmetic on the significands; nothing is approximated), tests/test_stopping_rule_gpu.py checks it against the
// one-chain kernel above on fields with ties at every scale.
template <typename T> struct refsum_traits;
// (U: the integer the summaries count units in -- everything saturates at 2^(P+1), so 32 bits do for float: half the instructions)
template <> struct refsum_traits<float> { static constexpr int P = 24, EMIN = -149; using U = uint32_t; };
template <> struct refsum_traits<double> { static constexpr int P = 53, EMIN = -1074; using U = unsigned long long; };
// v = M 2^E, M the integer significand (0 for v == 0), exact (v finite, >= 0)
__device__ __forceinline__ void refsum_split(float v, uint32_t& M, int& E) {
    const unsigned b = __float_as_uint(v);
    const int e = (int)((b >> 23) & 0xffu);
    const uint32_t f = b & 0x7fffffu;
    M = e ? (f | 0x800000u) : f;
    E = (e ? e : 1) - 150;
}
__device__ __forceinline__ void refsum_split(double v, unsigned long long& M, int& E) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int e = (int)((b >> 52) & 0x7ffull);
    const unsigned long long f = b & 0xfffffffffffffull;
    M = e ? (f | 0x10000000000000ull) : f;
    E = (e ? e : 1) - 1075;
}
// element x against the unit 2^k: nn = floor(x / u) (capped at 2^P), cls = 0 / 1 / 2: fraction below / exactly / above one half
template <typename T>
__device__ __forceinline__ void refsum_element(T x, int k, typename refsum_traits<T>::U& nn, int& cls) {
    using U = typename refsum_traits<T>::U;
    constexpr U LIMIT = (U)1 << refsum_traits<T>::P;
    constexpr int BITS = 8 * (int)sizeof(U);
    U M;
    int E;
    refsum_split(x, M, E);
    cls = 0;
    if (M == (U)0) { nn = (U)0; return; }
    if (E >= k) {
        const int sh = E - k;
        // (M < 2^P: a shift that would run out of the integer is far "beyond the binade" anyway -- at P + 1 bits the value is capped)
        nn = sh >= BITS - refsum_traits<T>::P - 1 ? LIMIT : (M << sh);
    } else {
        const int sh = k - E;
        if (sh >= BITS) { nn = (U)0; return; }   // (x < u / 2^(BITS - P): far below half a unit)
        nn = M >> sh;
        const U rem = M & (((U)1 << sh) - (U)1), half = (U)1 << (sh - 1);
        cls = rem < half ? 0 : (rem == half ? 1 : 2);
    }
    nn = nn > LIMIT ? LIMIT : nn;
}
// increment of the element for the parity `par` of the running sum in units
template <typename U>
__device__ __forceinline__ U refsum_incr(U nn, int cls, unsigned par) {
    return nn + (cls == 2 ? (U)1 : (cls == 1 ? (U)((par + (unsigned)nn) & 1u) : (U)0));
}
// Summary of a block of elements for a fixed unit: d[x] = its increment when the sum in front of it has parity x; r[x] = the largest
// "sum in front of an element + the most that element can add" inside the block, relative to the block's start (0: empty block).
// An element is taken by the recurrence only while S + nn + (fraction != 0) < 2^P -- the sum provably stays in the binade --; the
// first element for which that fails ends the round (it is added with a T addition).  With the start state S of a block that is
// S + r[S & 1] >= 2^P.  Values saturate at 2^(P+1) (far beyond the limit: only "reached" matters from there on).
template <typename U> struct RefSum4T { U d[2], r[2]; };
template <typename T> using RefSum4 = RefSum4T<typename refsum_traits<T>::U>;
template <typename T>
__device__ __forceinline__ typename refsum_traits<T>::U refsum_sat(typename refsum_traits<T>::U v) {
    using U = typename refsum_traits<T>::U;
    constexpr U CAP = (U)1 << (refsum_traits<T>::P + 1);
    return v > CAP ? CAP : v;
}
template <typename T>
__device__ __forceinline__ void refsum_push(RefSum4<T>& s, typename refsum_traits<T>::U nn, int cls) {   // s <- s then one element
    using U = typename refsum_traits<T>::U;
#pragma unroll
    for (unsigned x = 0; x < 2; ++x) {
        const U reach = refsum_sat<T>(s.d[x] + nn + (cls ? (U)1 : (U)0));
        s.r[x] = s.r[x] > reach ? s.r[x] : reach;
        s.d[x] = refsum_sat<T>(s.d[x] + refsum_incr<U>(nn, cls, ((unsigned)s.d[x] + x) & 1u));
    }
}
template <typename T>
__device__ __forceinline__ RefSum4<T> refsum_then(const RefSum4<T>& A, const RefSum4<T>& B) {        // A then B
    using U = typename refsum_traits<T>::U;
    RefSum4<T> c;
#pragma unroll
    for (unsigned x = 0; x < 2; ++x) {
        const unsigned y = ((unsigned)A.d[x] + x) & 1u;
        const U reach = refsum_sat<T>(A.d[x] + B.r[y]);
        c.r[x] = A.r[x] > reach ? A.r[x] : reach;
        c.d[x] = refsum_sat<T>(A.d[x] + B.d[y]);
    }
    return c;
}
struct RefSumState {       // device-resident between the rounds of a field
    unsigned long long start;   // first element not yet added
    unsigned long long bits;    // the running sum (bit pattern of a T)
    unsigned long long window;  // elements of the next round (whole tiles): follows the distance between the places where the sum
    unsigned long long prev_q;  // left a binade (the element behind the last such place)
};
constexpr unsigned long long FSM_REFSUM_WMIN = 1ull << 16, FSM_REFSUM_WMAX = 1ull << 25;
template <typename T>
struct RefSumArgs {
    const T* const* cur;   // [field] current field (element i at cur[f][i * stride])
    const T* const* old;   // [field] the snapshot, same layout (nullptr: cur holds the terms themselves)
    const unsigned long long* n;   // [field] number of terms
    int stride;
    RefSumState* st;       // [2][field]: a round reads the states of buffer round & 1 and writes the other one -- read-only within a
                           // launch, so that a workgroup dispatched late can never see the next round's state (round-5 advice)
    int round;             // number of this round (launch)
    RefSum4<T>* tiles;     // [field][FSM_REFSUM_WMAX / TILE] summaries of the tiles of the window
    unsigned* arrived;     // [field] workgroups of the round that are done with their tiles (zero between rounds)
    T stop_at;             // the sum only grows: a field whose running sum has reached this value is done (what the caller asks is
                           // `change >= epsilon`); infinity: the whole sum
};
constexpr int FSM_REFSUM_TILE = 4096, FSM_REFSUM_PER = FSM_REFSUM_TILE / 256;
template <typename T>
__device__ __forceinline__ T refsum_value(unsigned long long bits) {
    if constexpr (sizeof(T) == 4) return __uint_as_float((unsigned)bits); else return __longlong_as_double((long long)bits);
}
template <typename T>
__device__ __forceinline__ unsigned long long refsum_bits(T v) {
    if constexpr (sizeof(T) == 4) return __float_as_uint(v); else return (unsigned long long)__double_as_longlong(v);
}
// S 2^k as a T (S < 2^P: exact)
template <typename T>
__device__ __forceinline__ T refsum_make(unsigned long long S, int k) {
    if constexpr (sizeof(T) == 4) return __builtin_ldexpf((float)S, k); else return __builtin_ldexp((double)S, k);
}
// unit exponent k and integer S of a running sum s = S 2^k (S in [2^(P-1), 2^P) for a normal s; s == 0: the smallest unit, S = 0)
template <typename T>
__device__ __forceinline__ void refsum_unit(T s, int& k, unsigned long long& S) {
    int E;
    typename refsum_traits<T>::U M;
    refsum_split(s, M, E);
    S = M;
    k = S ? E : refsum_traits<T>::EMIN;
}
template <typename T>
struct RefSumField { const T* cur; const T* old; size_t n_nodes; int stride; };
template <typename T>
__device__ __forceinline__ T refsum_x(const RefSumField<T>& a, unsigned long long i) {
    if (a.old == nullptr) return a.cur[i * a.stride];          // (the terms themselves: fsm_refsum_terms below)
    const T df = a.old[i * a.stride] - a.cur[i * a.stride];   // times[n] - T[n], in T1 (ttcr/Grid3Drnfs.h:145)
    return df < 0 ? -df : df;
}
// The terms abs(times[n] - T[n]) of the fields of one slot group, one compact array per source that was asked for: the rounds above then
// read 4 bytes per term instead of two fields with the stride of the group's layout.  Zeros add nothing to the sum -- and in the
// iterations the rule decides, all but 1e-4 ... 5e-2 of the terms are zero (nodes that did not change; profiles/r06/stopping_rule.txt):
// the ordered pass runs over the non-zero terms alone, kept in node order.  fsm_refsum_terms writes the terms of a block of
// FSM_REFSUM_CB nodes and counts the non-zero ones; fsm_refsum_scan turns the counts into offsets (one workgroup per field; the total is
// the length of the compacted field); fsm_refsum_compact writes the non-zero terms of a block behind those of the blocks in front of it.
// In the same pass the snapshot is brought up to the current field -- the copy the next iteration would otherwise take
// (GridT::snapshots_before_iteration).  With the dirty-brick stamps of the SKIP sweep kernels (stamp != nullptr: 3-D grids) only the
// bricks that changed since the snapshot was taken (stamp >= thr) are read at all: the others hold zeros and a snapshot that is already
// right; a block none of whose bricks changed costs the stamps of its bricks.  x[] = cnt[] = nullptr: the snapshot alone.
// V elements per lane (16 bytes when the group's fields are aligned to that and V divides their length).
constexpr int FSM_REFSUM_CB = 4096;
template <typename T>
struct RefTermsArgs {
    const T* cur;          // [n_nodes * ns] the fields of the group, source l of node n at n * ns + l
    T* old;                // the snapshot, same layout; becomes a copy of cur
    T* x[2];               // [l] terms of source l, dense (nullptr: not asked); defined in the blocks with cnt > 0
    unsigned* cnt[2];      // [l][block] non-zero terms
    uint32_t n_nodes;
    int ns;
    const int* stamp;      // the group's brick stamps (sweep number of the last change), nullptr: every brick counts as changed
    int thr;
    int NF, NJ, nbf, nbj;
};
template <typename T, int V>
__global__ __launch_bounds__(256) void fsm_refsum_terms(const RefTermsArgs<T> a) {
    struct alignas(sizeof(T) * V) Vec { T v[V]; };
    constexpr int H = V > 1 ? V / 2 : 1;
    struct alignas(sizeof(T) * H) Half { T v[H]; };
    __shared__ unsigned s_c[2];
    __shared__ int s_any;
    const uint32_t n_el = a.n_nodes * (uint32_t)a.ns;
    const uint32_t vpb = (uint32_t)FSM_REFSUM_CB * a.ns / V;          // vectors per block
    const int subs = (int)(vpb / 256u);                               // (<= 32)
    const uint32_t nblk = (a.n_nodes + FSM_REFSUM_CB - 1) / FSM_REFSUM_CB;
    const bool want = a.x[0] != nullptr || a.x[1] != nullptr;
    for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        if (threadIdx.x == 0) { s_c[0] = 0u; s_c[1] = 0u; s_any = 0; }
        __syncthreads();
        // which of this lane's vectors lie in a brick that changed (one bit per vector; a vector never straddles two bricks: the
        // host sees to it that NF is a multiple of the nodes of a vector)
        unsigned chg = 0u;
        for (int sub = 0; sub < subs; ++sub) {
            const uint32_t q = blk * vpb + (uint32_t)sub * 256u + threadIdx.x;
            if ((uint64_t)q * V >= n_el) break;
            bool c = true;
            if (a.stamp) {
                const uint32_t node = q * (uint32_t)V / (uint32_t)a.ns;
                const uint32_t line = node / (uint32_t)a.NF, f = node - line * (uint32_t)a.NF;
                const uint32_t k = line / (uint32_t)a.NJ, j = line - k * (uint32_t)a.NJ;
                c = a.stamp[((size_t)(k / FSM_BRICK) * a.nbj + j / FSM_BRICK) * a.nbf + f / FSM_BRICK] >= a.thr;
            }
            chg |= c ? 1u << sub : 0u;
        }
        if (__ballot(chg != 0u) && (threadIdx.x & 63) == 0) s_any = 1;
        __syncthreads();
        if (!s_any) {   // (uniform) nothing changed in the block: no terms, the snapshot stands
            if (threadIdx.x < 2 && a.cnt[threadIdx.x]) a.cnt[threadIdx.x][blk] = 0u;
            __syncthreads();
            continue;
        }
        unsigned c0 = 0u, c1 = 0u;   // (wave-uniform tallies)
        for (int sub = 0; sub < subs; ++sub) {
            const uint32_t q = blk * vpb + (uint32_t)sub * 256u + threadIdx.x;
            const bool in = (uint64_t)q * V < n_el;
            Vec x;
#pragma unroll
            for (int e = 0; e < V; ++e) x.v[e] = (T)0;
            if (in && ((chg >> sub) & 1u)) {
                const Vec c = ((const Vec*)a.cur)[q];
                if (want) {
                    const Vec o = ((const Vec*)a.old)[q];
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const T df = o.v[e] - c.v[e];   // times[n] - T[n], in T1 (ttcr/Grid3Drnfs.h:145)
                        x.v[e] = df < 0 ? -df : df;
                    }
                }
                ((Vec*)a.old)[q] = c;
            }
            if (!want) continue;
            if (a.ns == 1) {
                if (in) ((Vec*)a.x[0])[q] = x;
#pragma unroll
                for (int e = 0; e < V; ++e) c0 += (unsigned)__builtin_popcountll(__ballot(x.v[e] != (T)0));
            } else if constexpr (V == 1) {
                if (in && a.x[q & 1u]) a.x[q & 1u][q >> 1] = x.v[0];
                c0 += (unsigned)__builtin_popcountll(__ballot(x.v[0] != (T)0 && !(q & 1u)));
                c1 += (unsigned)__builtin_popcountll(__ballot(x.v[0] != (T)0 && (q & 1u)));
            } else {
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    if (!a.x[l]) continue;
                    Half h;
#pragma unroll
                    for (int e = 0; e < H; ++e) {
                        h.v[e] = x.v[2 * e + l];
                        const unsigned n1 = (unsigned)__builtin_popcountll(__ballot(h.v[e] != (T)0));
                        if (l == 0) c0 += n1; else c1 += n1;
                    }
                    if (in) ((Half*)a.x[l])[q] = h;
                }
            }
        }
        if (want) {
            if ((threadIdx.x & 63) == 0) { if (c0) atomicAdd(&s_c[0], c0); if (c1) atomicAdd(&s_c[1], c1); }
            __syncthreads();
            if (threadIdx.x < 2 && a.cnt[threadIdx.x]) a.cnt[threadIdx.x][blk] = s_c[threadIdx.x];
        }
        __syncthreads();
    }
}
template <int DUMMY = 0>   // (a template: the header is part of two translation units)
__global__ __launch_bounds__(1024) void fsm_refsum_scan(const unsigned* __restrict__ cnt, unsigned long long* __restrict__ off, unsigned nblk,
                                                        unsigned long long* __restrict__ total) {
    __shared__ unsigned long long s_t[1024];
    const unsigned* __restrict__ c = cnt + (size_t)blockIdx.x * nblk;
    unsigned long long* __restrict__ o = off + (size_t)blockIdx.x * nblk;
    const unsigned per = (nblk + 1023u) / 1024u, b0 = threadIdx.x * per < nblk ? threadIdx.x * per : nblk, b1 = b0 + per < nblk ? b0 + per : nblk;
    unsigned long long s = 0;
    for (unsigned b = b0; b < b1; ++b) s += c[b];
    s_t[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {   // (1024 values: one lane)
        unsigned long long run = 0;
        for (int t = 0; t < 1024; ++t) { const unsigned long long v = s_t[t]; s_t[t] = run; run += v; }
        total[blockIdx.x] = run;
    }
    __syncthreads();
    unsigned long long run = s_t[threadIdx.x];
    for (unsigned b = b0; b < b1; ++b) { o[b] = run; run += c[b]; }
}
template <typename T>
__global__ __launch_bounds__(256) void fsm_refsum_compact(const T* __restrict__ x, T* __restrict__ out, size_t pitch, size_t n,
                                                          const unsigned* __restrict__ cnt, const unsigned long long* __restrict__ off, unsigned nblk) {
    constexpr int PER = FSM_REFSUM_CB / 256;
    __shared__ unsigned s_w[4];
    const T* __restrict__ xf = x + (size_t)blockIdx.y * pitch;
    T* __restrict__ of = out + (size_t)blockIdx.y * pitch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        if (cnt[(size_t)blockIdx.y * nblk + blk] == 0u) continue;   // (uniform: most blocks of a late iteration)
        const size_t base = (size_t)blk * FSM_REFSUM_CB + (size_t)threadIdx.x * PER;   // PER consecutive terms per lane, in node order
        T v[PER];
        unsigned c = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            v[q] = base + q < n ? xf[base + q] : (T)0;
            c += v[q] != (T)0 ? 1u : 0u;
        }
        // exclusive scan of the lanes' counts: within the wavefront, then over the four wavefronts
        unsigned inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        unsigned before = inc - c;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        unsigned long long p = off[(size_t)blockIdx.y * nblk + blk] + before;
#pragma unroll
        for (int q = 0; q < PER; ++q)
            if (v[q] != (T)0) of[p++] = v[q];
        __syncthreads();
    }
}
// summary of the FSM_REFSUM_PER consecutive elements from i0 on
template <typename T>
__device__ __forceinline__ RefSum4<T> refsum_chunk(const RefSumField<T>& a, unsigned long long i0, int k) {
    RefSum4<T> s = {{0, 0}, {0, 0}};
    for (int q = 0; q < FSM_REFSUM_PER; ++q) {
        const unsigned long long i = i0 + q;
        if (i >= a.n_nodes) break;
        typename refsum_traits<T>::U nn;
        int cls;
        refsum_element<T>(refsum_x(a, i), k, nn, cls);
        refsum_push<T>(s, nn, cls);
    }
    return s;
}
// ordered composition of the 256 summaries in sd[] (thread t holds elements before thread t + 1's): the result is in sd[0]
template <typename T>
__device__ __forceinline__ void refsum_tree(RefSum4<T>* sd, int tid) {
    for (int off = 1; off < 256; off <<= 1) {
        RefSum4<T> c = {{0, 0}, {0, 0}};
        const bool act = (tid & (2 * off - 1)) == 0;
        if (act) c = refsum_then<T>(sd[tid], sd[tid + off]);
        __syncthreads();
        if (act) sd[tid] = c;
        __syncthreads();
    }
}
// The head of a field (blockIdx.x), added the reference's way: the running sum doubles every few terms at first -- a round per binade
// would spend a launch on a handful of elements a dozen times over --, so the first FSM_REFSUM_HEAD terms are staged in LDS and added by
// one lane, one T addition after the other (ttcr/Grid3Drnfs.h:147).  Leaves the state the rounds start from.
constexpr int FSM_REFSUM_HEAD = 4096;
template <typename T>
__global__ __launch_bounds__(256) void fsm_refsum_head(const RefSumArgs<T> a) {
    __shared__ T xs[FSM_REFSUM_HEAD];
    const int fi = blockIdx.x;
    const unsigned long long n_terms = a.n[fi];
    const RefSumField<T> f = {a.cur[fi], a.old[fi], n_terms, a.stride};
    const int m = (int)(n_terms < (unsigned long long)FSM_REFSUM_HEAD ? n_terms : (unsigned long long)FSM_REFSUM_HEAD);
    for (int i = threadIdx.x; i < FSM_REFSUM_HEAD; i += 256) xs[i] = i < m ? refsum_x(f, (unsigned long long)i) : (T)0;   // (zeros add nothing)
    __syncthreads();
    if (threadIdx.x != 0) return;
    T s = (T)0;
    int done = m;
    for (int i = 0; i < m; i += 16) {   // (16 values in flight ahead of the chain of additions; a sum that has reached stop_at is decided)
        T v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = xs[i + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) s = s + v[q];
        if (s >= a.stop_at) { done = -1; break; }
    }
    RefSumState ns;
    ns.start = done < 0 ? n_terms : (unsigned long long)done;
    ns.bits = refsum_bits<T>(s);
    ns.window = FSM_REFSUM_WMIN;
    ns.prev_q = ns.start;
    a.st[(size_t)(a.round & 1) * gridDim.x + fi] = ns;
}
// The ordered walk over n summaries in LDS (n a multiple of 8; empty summaries change nothing): the first entry at which the sum may
// leave its binade, S left at the state in front of it; n if there is none.  Eight summaries are fetched whole before the dependent chain
// of selects and additions runs over them: fetching sd[c].r[S & 1] -- an address that depends on S -- made every entry two LDS round
// trips, 14 us per 256 entries and most of a round (profiles/r06/stopping_rule.txt).
template <typename T>
__device__ __forceinline__ int refsum_walk(const RefSum4<T>* sd, int n, unsigned long long& S) {
    constexpr unsigned long long LIMIT = 1ull << refsum_traits<T>::P;
    for (int c = 0; c < n; c += 8) {
        RefSum4<T> e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = sd[c + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool odd = ((unsigned)S & 1u) != 0u;
            if (S + (unsigned long long)(odd ? e[q].r[1] : e[q].r[0]) >= LIMIT) return c + q;
            S += (unsigned long long)(odd ? e[q].d[1] : e[q].d[0]);
        }
    }
    return n;
}
// One round of a field (blockIdx.y): the workgroups summarise the tiles of the window for the unit of the current sum (grid stride);
// the one that finishes last composes the summaries in order up to the first tile in which an element may take the sum out of its
// binade, finds that element and adds it with a T addition -- or takes the whole window -- and writes the next state.
template <typename T>
__global__ __launch_bounds__(256) void fsm_refsum_round(const RefSumArgs<T> a) {
    constexpr unsigned long long LIMIT = 1ull << refsum_traits<T>::P;
    __shared__ RefSum4<T> sd[256];
    __shared__ unsigned long long s_S, s_tile;
    __shared__ int s_found, s_last;
    const int tid = threadIdx.x;
    const int fi = blockIdx.y;
    const unsigned long long n_terms = a.n[fi];
    const RefSumState st = a.st[(size_t)(a.round & 1) * gridDim.y + fi];
    RefSumState& st_next = a.st[(size_t)((a.round + 1) & 1) * gridDim.y + fi];
    const unsigned long long start = st.start;
    if (start >= n_terms) {   // (this field is done: the rounds are enqueued in bunches; its state moves on unchanged)
        if (blockIdx.x == 0 && tid == 0) st_next = st;
        return;
    }
    const RefSumField<T> f = {a.cur[fi], a.old[fi], n_terms, a.stride};
    RefSum4<T>* __restrict__ tiles = a.tiles + (size_t)fi * (FSM_REFSUM_WMAX / FSM_REFSUM_TILE);
    int k;
    unsigned long long S0;
    refsum_unit<T>(refsum_value<T>(st.bits), k, S0);
    unsigned long long n_left = n_terms - start;
    n_left = n_left < st.window ? n_left : st.window;
    const unsigned long long n_tiles = (n_left + FSM_REFSUM_TILE - 1) / FSM_REFSUM_TILE;
    if (blockIdx.x >= n_tiles) return;   // (no tile for this workgroup in this round: it is not counted either)
    const unsigned n_part = (unsigned)(n_tiles < (unsigned long long)gridDim.x ? n_tiles : (unsigned long long)gridDim.x);
    // the elements of a tile come in coalesced (thread t: elements t, t + 256, ...) and go through LDS to the threads that walk them
    // in order, 16 consecutive ones each (17 words per 16 elements: the walks of a wavefront hit different banks)
    __shared__ T xs[FSM_REFSUM_TILE + FSM_REFSUM_TILE / FSM_REFSUM_PER];
    for (unsigned long long b = blockIdx.x; b < n_tiles; b += gridDim.x) {
        const unsigned long long tb = start + b * FSM_REFSUM_TILE;
#pragma unroll 4
        for (int q = 0; q < FSM_REFSUM_PER; ++q) {
            const int e = q * 256 + tid;
            const unsigned long long i = tb + e;
            xs[e + e / FSM_REFSUM_PER] = i < n_terms ? refsum_x(f, i) : (T)0;   // (beyond the field: zeros add nothing)
        }
        __syncthreads();
        RefSum4<T> c = {{0, 0}, {0, 0}};
#pragma unroll 4
        for (int q = 0; q < FSM_REFSUM_PER; ++q) {
            typename refsum_traits<T>::U nn;
            int cls;
            refsum_element<T>(xs[tid * (FSM_REFSUM_PER + 1) + q], k, nn, cls);
            refsum_push<T>(c, nn, cls);
        }
        sd[tid] = c;
        __syncthreads();
        refsum_tree<T>(sd, tid);
        if (tid == 0) tiles[b] = sd[0];
        __syncthreads();
    }
    // the last workgroup to get here goes on (release of the summaries before the arrival, acquire after it: one lane each)
    if (tid == 0) {
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the write-back of the release is complete before the arrival: MI355X guide, compiler hazard)
        s_last = atomicAdd(a.arrived + fi, 1u) == n_part - 1u;
        if (s_last) {
            __threadfence();
            a.arrived[fi] = 0u;
        }
    }
    __syncthreads();
    if (!s_last) return;

    // ranges of tiles per thread (their loads overlap across the threads), composed by thread 0
    const unsigned long long per = (n_tiles + 255) / 256;
    {
        RefSum4<T> s = {{0, 0}, {0, 0}};
        for (unsigned long long t = (unsigned long long)tid * per; t < ((unsigned long long)tid + 1) * per && t < n_tiles; ++t) s = refsum_then<T>(s, tiles[t]);
        sd[tid] = s;
    }
    __syncthreads();
    // (everything thread 0 walks below sits in LDS by then: a walk over values it had to fetch itself was a chain of memory round trips,
    // 60-100 us per round, 2-6 ms per sum of a 512^3 field; profiles/r05/README.md)
    __shared__ int s_range;
    if (tid == 0) {
        unsigned long long S = S0;
        const int c = refsum_walk<T>(sd, 256, S);   // (the range in which the sum may leave its binade)
        s_S = S; s_range = c; s_found = c < 256;
    }
    __syncthreads();
    if (s_found) {   // the tiles of that range, one by one
        const unsigned long long t0 = (unsigned long long)s_range * per;
        __syncthreads();
        {   // (per <= WMAX / TILE / 256 = 32 tiles: one pass; padded with empty summaries to a multiple of 8 for the walk)
            const unsigned long long q = (unsigned long long)tid;
            if (q < ((per + 7ull) & ~7ull) && q < 256ull) sd[q] = (q < per && t0 + q < n_tiles) ? tiles[t0 + q] : RefSum4<T>{{0, 0}, {0, 0}};
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long S = s_S, t = t0;
            int found = 0;
            const int n2 = (int)(per < 256ull ? (per + 7ull) & ~7ull : 256ull);
            const int c = refsum_walk<T>(sd, n2, S);   // (entries behind the range or the window are empty: they cannot be it)
            if (c < n2) { t = t0 + (unsigned long long)c; found = 1; }
            // (found == 0 cannot happen -- the summary of the range said so; the walk then stands at the end of the range with the
            // exact state: go on from the next tile, or the window is over)
            if (!found) { t = t0 + per; found = t < n_tiles; }
            s_S = S; s_tile = t; s_found = found;
        }
        __syncthreads();
    }
    if (!s_found) {   // the whole window in this binade
        if (tid == 0) {
            RefSumState ns;
            ns.start = start + n_left;
            ns.bits = refsum_bits<T>(refsum_make<T>(s_S, k));
            ns.window = 2ull * st.window < FSM_REFSUM_WMAX ? 2ull * st.window : FSM_REFSUM_WMAX;
            ns.prev_q = st.prev_q;
            if (refsum_value<T>(ns.bits) >= a.stop_at) ns.start = n_terms;
            st_next = ns;
        }
        return;
    }
    // inside tile s_tile (state s_S at its start): the chunks of the threads, then the elements of one chunk
    const unsigned long long tbase = start + s_tile * FSM_REFSUM_TILE;
    __syncthreads();
    sd[tid] = refsum_chunk<T>(f, tbase + (unsigned long long)tid * FSM_REFSUM_PER, k);
    __syncthreads();
    if (tid == 0) {
        unsigned long long S = s_S;
        const int c = refsum_walk<T>(sd, 256, S);
        s_S = S; s_range = c;
    }
    __syncthreads();
    __shared__ T s_x[FSM_REFSUM_PER + 1];
    {
        const unsigned long long q0 = tbase + (unsigned long long)s_range * FSM_REFSUM_PER;
        if (tid <= FSM_REFSUM_PER) s_x[tid] = q0 + tid < n_terms ? refsum_x(f, q0 + tid) : (T)0;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long S = s_S;
        const int c = s_range;
        unsigned long long q = tbase + (unsigned long long)c * FSM_REFSUM_PER;
        const unsigned long long q0 = q, qend = q + FSM_REFSUM_PER;
        for (; q < qend && q < n_terms; ++q) {
            typename refsum_traits<T>::U nn;
            int cls;
            refsum_element<T>(s_x[q - q0], k, nn, cls);
            if (S + nn + (cls ? 1ull : 0ull) >= LIMIT) break;
            S += refsum_incr<typename refsum_traits<T>::U>(nn, cls, (unsigned)S & 1u);
        }
        // q: the element that may take the sum out of the binade, S: the state in front of it (c == 256 or q == qend cannot happen --
        // the summaries said so -- and would only cost a round: any element may be added the reference's way)
        T v = refsum_make<T>(S, k);
        RefSumState ns;
        ns.window = st.window;
        ns.prev_q = st.prev_q;
        if (q < n_terms) {
            v = v + s_x[q - q0];   // the reference's own addition (ttcr/Grid3Drnfs.h:147)
            ns.start = q + 1ull;
            // the sum left its binade here (or nearly): the next such place is about twice as far again; whole tiles (a tile is summarised to its end)
            unsigned long long w = 4ull * (q + 1ull - st.prev_q);   // (twice that distance ended just short of the next place every other time)
            w = w < FSM_REFSUM_WMIN ? FSM_REFSUM_WMIN : (w > FSM_REFSUM_WMAX ? FSM_REFSUM_WMAX : w);
            ns.window = (w + FSM_REFSUM_TILE - 1) / FSM_REFSUM_TILE * FSM_REFSUM_TILE;
            ns.prev_q = q + 1ull;
        } else {
            ns.start = n_terms;
        }
        ns.bits = refsum_bits<T>(v);
        if (v >= a.stop_at) ns.start = n_terms;
        st_next = ns;
    }
}

// de-interleave one source's field for the host (getTT)
template <typename T>
__global__ void fsm_gather_field(const T* __restrict__ p, T* __restrict__ out, size_t n, int stride) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = p[i * stride];
}

// Grid3Drcfs::setSlowness (ttcr/Grid3Drcfs.h:88-171) / Grid2Drcfs::setSlowness
// (ttcr/Grid2Drcfs.h:98-138): node slowness = mean of the touching cells, summed in the
// reference's order: k outer / j / i inner, except on the x-min/x-max faces where k is innermost.
template <typename T>
__global__ void fsm_cells_to_nodes3d(const T* __restrict__ sc, T* __restrict__ sn, int ncx, int ncy, int ncz) {
    const int nnx = ncx + 1, nny = ncy + 1, nnz = ncz + 1;
    const size_t N = (size_t)nnx * nny * nnz;
    for (size_t n = blockIdx.x * (size_t)blockDim.x + threadIdx.x; n < N; n += (size_t)gridDim.x * blockDim.x) {
        const int i = n % nnx, j = (n / nnx) % nny, k = n / ((size_t)nnx * nny);
        int ci[2], cj[2], ck[2], ni = 0, nj = 0, nk = 0;
        if (i < ncx) ci[ni++] = i;
        if (i > 0) ci[ni++] = i - 1;
        if (j < ncy) cj[nj++] = j;
        if (j > 0) cj[nj++] = j - 1;
        if (k < ncz) ck[nk++] = k;
        if (k > 0) ck[nk++] = k - 1;
        T sum = 0;
        bool first = true;
        if (ni == 1) {
            for (int b = 0; b < nj; ++b)
                for (int a = 0; a < nk; ++a) {
                    const T v = sc[((size_t)ck[a] * ncy + cj[b]) * ncx + ci[0]];
                    sum = first ? v : sum + v;
                    first = false;
                }
        } else {
            for (int a = 0; a < nk; ++a)
                for (int b = 0; b < nj; ++b)
                    for (int c = 0; c < ni; ++c) {
                        const T v = sc[((size_t)ck[a] * ncy + cj[b]) * ncx + ci[c]];
                        sum = first ? v : sum + v;
                        first = false;
                    }
        }
        const int cnt = ni * nj * nk;
        sn[n] = cnt == 1 ? sum : (cnt == 2 ? (T)(0.5 * sum) : (cnt == 4 ? (T)(0.25 * sum) : (T)(0.125 * sum)));
    }
}

template <typename T>
__global__ void fsm_cells_to_nodes2d(const T* __restrict__ sc, T* __restrict__ sn, int ncx, int ncz) {
    const int nnx = ncx + 1, nnz = ncz + 1;
    const size_t N = (size_t)nnx * nnz;
    for (size_t n = blockIdx.x * (size_t)blockDim.x + threadIdx.x; n < N; n += (size_t)gridDim.x * blockDim.x) {
        const int j = n % nnz, i = n / nnz;
        int ci[2], cj[2], ni = 0, nj = 0;
        if (i < ncx) ci[ni++] = i;
        if (i > 0) ci[ni++] = i - 1;
        if (j < ncz) cj[nj++] = j;
        if (j > 0) cj[nj++] = j - 1;
        T sum = 0;
        bool first = true;
        for (int a = 0; a < ni; ++a)
            for (int b = 0; b < nj; ++b) {
                const T v = sc[(size_t)ci[a] * ncz + cj[b]];
                sum = first ? v : sum + v;
                first = false;
            }
        const int cnt = ni * nj;
        sn[n] = cnt == 1 ? sum : (cnt == 2 ? (T)(0.5 * sum) : (T)(0.25 * sum));
    }
}

// Source initialisation: Grid3Drn::initFSM (ttcr/Grid3Drn.h:3487-3556) and Grid2Drn::initFSM
// (ttcr/Grid2Drn.h:1360-1418).  The host has already located each source point (node hit or
// enclosing cell, with the reference's tolerance tests); this kernel applies the points of ONE
// source in order (later points overwrite earlier ones) to one slot.
template <typename T>
struct InitPoint {
    T x, y, z, t0;       // source point (grid coordinates) and origin time; 2-D: x, z used
    int i, j, k;         // node (on_node) or cell indices
    int on_node;
};

template <typename T>
struct InitArgs {
    T* tt;                 // slot field (element n at tt[n*ts])
    int ts;
    const T* slowness;
    uint32_t* frozen;      // slot mask
    int* bbox;             // slot bbox (F,J,K lo/hi) -- written by thread 0
    const InitPoint<T>* pts;
    int n_pts, npts;       // npts = 1 (first order)
    int nnx, nny, nnz;     // 2-D: nnx, nnz, nny = 1
    T dx, dz, xmin, ymin, zmin;
    int dim;
    int* stamp;            // slot's dirty-brick stamps (all -1 on entry); bricks of the source box get 0
    int nbf, nbj, nbk;     // brick counts along the sweep kernel's F, J, K axes
};

template <typename T>
__device__ __forceinline__ T node_coord(T cmin, uint32_t n, T d) { return cmin + (T)n * d; }

template <typename T>
__global__ void fsm_init_source(const InitArgs<T> a) {
    // one block; box nodes are handled by the first threads, points strictly in order
    const int tid = threadIdx.x;
    int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
    for (int n = 0; n < a.n_pts; ++n) {
        const InitPoint<T> p = a.pts[n];
        const int npts = a.npts;
        const int b0 = p.on_node ? -npts : -(npts - 1);
        const int w = npts - b0 + 1;  // box edge
        const int nbox = a.dim == 3 ? w * w * w : w * w;
        if (p.on_node && tid == 0) {
            const size_t nn = a.dim == 3 ? ((size_t)p.k * a.nny + p.j) * a.nnx + p.i : (size_t)p.i * a.nnz + p.k;
            a.tt[nn * a.ts] = p.t0;
            atomicOr(&a.frozen[nn >> 5], 1u << (nn & 31));
        }
        __syncthreads();
        for (int b = tid; b < nbox; b += blockDim.x) {
            if (a.dim == 3) {
                const int ii = p.i + b0 + b % w, jj = p.j + b0 + (b / w) % w, kk = p.k + b0 + b / (w * w);
                if (ii < 0 || ii >= a.nnx || jj < 0 || jj >= a.nny || kk < 0 || kk >= a.nnz) continue;
                if (ii == p.i && jj == p.j && kk == p.k) continue;  // skipped in both branches (:3505, :3541)
                const size_t m = ((size_t)kk * a.nny + jj) * a.nnx + ii;
                const T x = node_coord(a.xmin, ii, a.dx), y = node_coord(a.ymin, jj, a.dx), z = node_coord(a.zmin, kk, a.dx);
                const T d2 = (x - p.x) * (x - p.x) + (y - p.y) * (y - p.y) + (z - p.z) * (z - p.z);
                const T d = (T)__builtin_sqrt((double)d2);  // == correctly rounded sqrt in T
                a.tt[m * a.ts] = p.t0 + d * a.slowness[m];
                atomicOr(&a.frozen[m >> 5], 1u << (m & 31));
            } else {
                const int ii = p.i + b0 + b / w, kk = p.k + b0 + b % w;  // kk: z index
                if (ii < 0 || ii >= a.nnx || kk < 0 || kk >= a.nnz) continue;
                if (p.on_node && ii == p.i && kk == p.k) continue;  // 2-D skips only in the on-node branch
                const size_t m = (size_t)ii * a.nnz + kk;
                const T x = node_coord(a.xmin, ii, a.dx), z = node_coord(a.zmin, kk, a.dz);
                const T d2 = (x - p.x) * (x - p.x) + (z - p.z) * (z - p.z);
                const T d = (T)__builtin_sqrt((double)d2);
                T tt;
                if (p.on_node) {
                    // t0 + dist*0.5*(s_nbr + s_src): the 0.5 literal makes this a double chain (:1384)
                    const size_t nn = (size_t)p.i * a.nnz + p.k;
                    const T ssum = a.slowness[m] + a.slowness[nn];
                    tt = (T)((double)p.t0 + ((double)d * 0.5) * (double)ssum);
                } else {
                    tt = p.t0 + d * a.slowness[m];
                }
                a.tt[m * a.ts] = tt;
                atomicOr(&a.frozen[m >> 5], 1u << (m & 31));
            }
        }
        __syncthreads();
        // bounding box of everything this point may have frozen (clipped)
        const int ci[3] = {p.i, p.j, p.k};
        const int nn3[3] = {a.nnx, a.dim == 3 ? a.nny : 1, a.nnz};
        for (int d = 0; d < 3; ++d) {
            if (a.dim == 2 && d == 1) { lo[1] = 0; hi[1] = 0; continue; }
            int l = ci[d] + b0, h = ci[d] + npts;
            l = l < 0 ? 0 : l;
            h = h > nn3[d] - 1 ? nn3[d] - 1 : h;
            lo[d] = l < lo[d] ? l : lo[d];
            hi[d] = h > hi[d] ? h : hi[d];
        }
    }
    if (tid == 0) {
        // stored in the sweep kernel's (F, J, K) axis order
        if (a.dim == 3) {
            a.bbox[0] = lo[0]; a.bbox[1] = hi[0]; a.bbox[2] = lo[1]; a.bbox[3] = hi[1]; a.bbox[4] = lo[2]; a.bbox[5] = hi[2];
        } else {  // F = z, J = x, K = none
            a.bbox[0] = lo[2]; a.bbox[1] = hi[2]; a.bbox[2] = lo[0]; a.bbox[3] = hi[0]; a.bbox[4] = 0; a.bbox[5] = 0;
        }
        if (a.stamp) {
            for (int bk = a.bbox[4] / FSM_BRICK; bk <= a.bbox[5] / FSM_BRICK; ++bk)
                for (int bj = a.bbox[2] / FSM_BRICK; bj <= a.bbox[3] / FSM_BRICK; ++bj)
                    for (int bf = a.bbox[0] / FSM_BRICK; bf <= a.bbox[1] / FSM_BRICK; ++bf)
                        a.stamp[((size_t)bk * a.nbj + bj) * a.nbf + bf] = 0;
        }
    }
}

// Receiver traveltimes: Grid3Drn::getTraveltime (ttcr/Grid3Drn.h:794-930)
template <typename T>
__device__ __forceinline__ T interp3d_pt(const T* __restrict__ Tn, int ts, T px, T py, T pz, int nnx, int nny, int nnz, T dx,
                                         T xmin, T ymin, T zmin) {
    const double small2 = 1.e-4 * 1.e-4;
    const T dy = dx, dz = dx;
    const uint32_t i = idx_u32(small2 + (double)((px - xmin) / dx));
    const uint32_t j = idx_u32(small2 + (double)((py - ymin) / dy));
    const uint32_t k = idx_u32(small2 + (double)((pz - zmin) / dz));
    auto ab = [](T v) { return v < 0 ? -v : v; };
    const bool onx = (double)ab(px - (xmin + (T)i * dx)) < small2;
    const bool ony = (double)ab(py - (ymin + (T)j * dy)) < small2;
    const bool onz = (double)ab(pz - (zmin + (T)k * dz)) < small2;
    // The index is a rounded quotient plus 1e-8, "on the plane" an absolute distance below 1e-8: a point a rounding
    // error below the last plane of an axis gets the last node as lower index without being on the plane, and the
    // reference reads node index+1 (past the row / the array).  Clamped to the last node, like the oracle's TT.
    auto cl = [](uint32_t v, int n) { return v < (uint32_t)n ? v : (uint32_t)n - 1u; };
#define TT(ii, jj, kk) Tn[(((size_t)cl(kk, nnz) * nny + cl(jj, nny)) * nnx + cl(ii, nnx)) * ts]
    T tt;
    if (onx && ony && onz) {
        tt = TT(i, j, k);
    } else if (onx && ony) {
        T t1 = TT(i, j, k), t2 = TT(i, j, k + 1);
        T w1 = (zmin + (T)(k + 1) * dz - pz) / dz, w2 = (pz - (zmin + (T)k * dz)) / dz;
        tt = t1 * w1 + t2 * w2;
    } else if (onx && onz) {
        T t1 = TT(i, j, k), t2 = TT(i, j + 1, k);
        T w1 = (ymin + (T)(j + 1) * dy - py) / dy, w2 = (py - (ymin + (T)j * dy)) / dy;
        tt = t1 * w1 + t2 * w2;
    } else if (ony && onz) {
        T t1 = TT(i, j, k), t2 = TT(i + 1, j, k);
        T w1 = (xmin + (T)(i + 1) * dx - px) / dx, w2 = (px - (xmin + (T)i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    } else if (onx) {
        T t1 = TT(i, j, k), t2 = TT(i, j, k + 1), t3 = TT(i, j + 1, k), t4 = TT(i, j + 1, k + 1);
        T w1 = (zmin + (T)(k + 1) * dz - pz) / dz, w2 = (pz - (zmin + (T)k * dz)) / dz;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (ymin + (T)(j + 1) * dy - py) / dy;
        w2 = (py - (ymin + (T)j * dy)) / dy;
        tt = t1 * w1 + t2 * w2;
    } else if (ony) {
        T t1 = TT(i, j, k), t2 = TT(i, j, k + 1), t3 = TT(i + 1, j, k), t4 = TT(i + 1, j, k + 1);
        T w1 = (zmin + (T)(k + 1) * dz - pz) / dz, w2 = (pz - (zmin + (T)k * dz)) / dz;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (xmin + (T)(i + 1) * dx - px) / dx;
        w2 = (px - (xmin + (T)i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    } else if (onz) {
        T t1 = TT(i, j, k), t2 = TT(i, j + 1, k), t3 = TT(i + 1, j, k), t4 = TT(i + 1, j + 1, k);
        T w1 = (ymin + (T)(j + 1) * dy - py) / dy, w2 = (py - (ymin + (T)j * dy)) / dy;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (xmin + (T)(i + 1) * dx - px) / dx;
        w2 = (px - (xmin + (T)i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    } else {
        T t1 = TT(i, j, k), t2 = TT(i, j, k + 1), t3 = TT(i, j + 1, k), t4 = TT(i, j + 1, k + 1);
        T t5 = TT(i + 1, j, k), t6 = TT(i + 1, j, k + 1), t7 = TT(i + 1, j + 1, k), t8 = TT(i + 1, j + 1, k + 1);
        T w1 = (zmin + (T)(k + 1) * dz - pz) / dz, w2 = (pz - (zmin + (T)k * dz)) / dz;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        t3 = t5 * w1 + t6 * w2;
        t4 = t7 * w1 + t8 * w2;
        w1 = (ymin + (T)(j + 1) * dy - py) / dy;
        w2 = (py - (ymin + (T)j * dy)) / dy;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (xmin + (T)(i + 1) * dx - px) / dx;
        w2 = (px - (xmin + (T)i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    }
#undef TT
    return tt;
}

template <typename T>
__global__ void fsm_interp3d(const T* __restrict__ Tn, int ts, const T* __restrict__ pts, T* __restrict__ out, int n,
                             int nnx, int nny, int nnz, T dx, T xmin, T ymin, T zmin) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    out[r] = interp3d_pt(Tn, ts, pts[3 * r], pts[3 * r + 1], pts[3 * r + 2], nnx, nny, nnz, dx, xmin, ymin, zmin);
}

// (nx, ny, nz) array in C order (z fastest, what numpy hands over) -> the solver's flat order (x fastest):
// 32 x 32 tiles of an (i, k) plane through LDS, reads coalesced along k, writes along i.  blockDim = (32, 8).
template <typename T>
__global__ void fsm_c_to_x_fastest(const T* __restrict__ in, T* __restrict__ out, int nx, int ny, int nz) {
    __shared__ T tile[32][33];
    const int j = blockIdx.z;
    const int k0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = i0 + r, k = k0 + threadIdx.x;
        if (i < nx && k < nz) tile[r][threadIdx.x] = in[((size_t)i * ny + j) * nz + k];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int k = k0 + r, i = i0 + threadIdx.x;
        if (i < nx && k < nz) out[((size_t)k * ny + j) * nx + i] = tile[threadIdx.x][r];
    }
}

// the receivers of a whole batch of sources in one launch: receiver r reads the field of slot slot_of[r]
// (fields of a group interleaved: element stride ts, group stride ts * n_nodes)
template <typename T>
__global__ void fsm_interp3d_batch(const T* __restrict__ tt0, int ts, size_t n_nodes, const int* __restrict__ slot_of,
                                   const T* __restrict__ pts, T* __restrict__ out, int n, int nnx, int nny, int nnz, T dx,
                                   T xmin, T ymin, T zmin) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int slot = slot_of[r];
    const T* Tn = tt0 + (size_t)(slot / ts) * n_nodes * ts + slot % ts;
    out[r] = interp3d_pt(Tn, ts, pts[3 * r], pts[3 * r + 1], pts[3 * r + 2], nnx, nny, nnz, dx, xmin, ymin, zmin);
}

// ---- traveltime from raypath (tt_from_rp, the 3-D default of ttcrpy) --------------------------
// Grid3Drn::getTraveltimeFromRaypath (ttcr/Grid3Drn.h:1103-1243): steepest-descent walk from the
// receiver to the source through the traveltime field -- gradient by the 4th-order centred
// operator grad (:1033-1100) on the trilinearly interpolated field, one grid plane per step,
// trapezoidal integration of the interpolated slowness computeSlowness (:2451-2676).  One thread
// per receiver; every expression keeps the reference's T1/double mix.
// receivers of several sources in one launch: receiver r belongs to batch entry rx_src[r]
struct RaySrc {
    long long tt_off;  // offset of the source's field from the first field (elements)
    int tx_off, n_tx;  // its points in src / t0
};

template <typename T>
struct RayGeom {
    int nnx, nny, nnz;
    T dx, xmin, ymin, zmin, xmax, ymax, zmax;
    int interp_vel;
};

__device__ __forceinline__ float rabs(float v) { return __builtin_fabsf(v); }
__device__ __forceinline__ double rabs(double v) { return __builtin_fabs(v); }

// first node index n with |p - (cmin + n*d)| < small2, or -1 (the reference scans all nodes)
template <typename T>
__device__ __forceinline__ int on_node(T p, T cmin, T d, int nn) {
    const double small2 = 1.e-4 * 1.e-4;
    long c = (long)__builtin_floor(((double)p - (double)cmin) / (double)d);
    long lo = c - 2 < 0 ? 0 : c - 2, hi = c + 3 > nn - 1 ? nn - 1 : c + 3;
    for (long n = lo; n <= hi; ++n)
        if ((double)rabs(p - (cmin + (T)(unsigned long)n * d)) < small2) return (int)n;
    return -1;
}

template <typename T>
__device__ T slowness_at3d(const RayGeom<T>& g, const T* __restrict__ sn, T px, T py, T pz) {
    const int nnx = g.nnx, nny = g.nny;
    const T xmin = g.xmin, ymin = g.ymin, zmin = g.zmin, dx = g.dx, dy = g.dx, dz = g.dx;
    const double small = 1.e-4;
    const int iv = g.interp_vel;
    const int onX = on_node(px, xmin, dx, g.nnx), onY = on_node(py, ymin, dy, g.nny), onZ = on_node(pz, zmin, dz, g.nnz);
    // (an index one past the last node -- the reference's cell index of a point within 1e-4 cell below the last
    // plane of an axis, where it reads past its array -- is clamped to the last node; see the oracle's SN)
    auto SN = [&](unsigned i, unsigned j, unsigned k) {
        i = i < (unsigned)g.nnx ? i : (unsigned)g.nnx - 1;
        j = j < (unsigned)g.nny ? j : (unsigned)g.nny - 1;
        k = k < (unsigned)g.nnz ? k : (unsigned)g.nnz - 1;
        const T v = sn[((size_t)k * nny + j) * nnx + i];
        return iv ? (T)(1.0 / (double)v) : v;
    };
    auto RET = [&](T v) { return iv ? (T)(1.0 / (double)v) : v; };
    auto lin1 = [](const T x[3], const T s[2]) { return (s[0] * (x[2] - x[0]) + s[1] * (x[0] - x[1])) / (x[2] - x[1]); };
    auto lin2 = [](const T x[3], const T y[3], const T s[4]) {
        return (s[0] * (x[2] - x[0]) * (y[2] - y[0]) + s[1] * (x[2] - x[0]) * (y[0] - y[1]) +
                s[2] * (x[0] - x[1]) * (y[2] - y[0]) + s[3] * (x[0] - x[1]) * (y[0] - y[1])) /
               ((x[2] - x[1]) * (y[2] - y[1]));
    };
    T s[8], x[3], y[3], z[3];
    if (onX != -1 && onY != -1 && onZ != -1) {
        return sn[((size_t)onZ * nny + onY) * nnx + onX];
    } else if (onX != -1 && onY != -1) {
        const unsigned k = idx_u32(small + (double)((pz - zmin) / dz));
        s[0] = SN(onX, onY, k); s[1] = SN(onX, onY, k + 1);
        x[0] = pz; x[1] = zmin + (T)k * dz; x[2] = zmin + (T)(k + 1) * dz;
        return RET(lin1(x, s));
    } else if (onX != -1 && onZ != -1) {
        const unsigned j = idx_u32(small + (double)((py - ymin) / dy));
        s[0] = SN(onX, j, onZ); s[1] = SN(onX, j + 1, onZ);
        x[0] = py; x[1] = ymin + (T)j * dy; x[2] = ymin + (T)(j + 1) * dy;
        return RET(lin1(x, s));
    } else if (onY != -1 && onZ != -1) {
        const unsigned i = idx_u32(small + (double)((px - xmin) / dx));
        s[0] = SN(i, onY, onZ); s[1] = SN(i + 1, onY, onZ);
        x[0] = px; x[1] = xmin + (T)i * dx; x[2] = xmin + (T)(i + 1) * dx;
        return RET(lin1(x, s));
    } else if (onX != -1) {
        const unsigned j = idx_u32(small + (double)((py - ymin) / dy));
        const unsigned k = idx_u32(small + (double)((pz - zmin) / dz));
        s[0] = SN(onX, j, k); s[1] = SN(onX, j, k + 1); s[2] = SN(onX, j + 1, k); s[3] = SN(onX, j + 1, k + 1);
        x[0] = py; y[0] = pz; x[1] = ymin + (T)j * dy; y[1] = zmin + (T)k * dz; x[2] = ymin + (T)(j + 1) * dy; y[2] = zmin + (T)(k + 1) * dz;
        return RET(lin2(x, y, s));
    } else if (onY != -1) {
        const unsigned i = idx_u32(small + (double)((px - xmin) / dx));
        const unsigned k = idx_u32(small + (double)((pz - zmin) / dz));
        s[0] = SN(i, onY, k); s[1] = SN(i, onY, k + 1); s[2] = SN(i + 1, onY, k); s[3] = SN(i + 1, onY, k + 1);
        x[0] = px; y[0] = pz; x[1] = xmin + (T)i * dx; y[1] = zmin + (T)k * dz; x[2] = xmin + (T)(i + 1) * dx; y[2] = zmin + (T)(k + 1) * dz;
        return RET(lin2(x, y, s));
    } else if (onZ != -1) {
        const unsigned i = idx_u32(small + (double)((px - xmin) / dx));
        const unsigned j = idx_u32(small + (double)((py - ymin) / dy));
        s[0] = SN(i, j, onZ); s[1] = SN(i, j + 1, onZ); s[2] = SN(i + 1, j, onZ); s[3] = SN(i + 1, j + 1, onZ);
        x[0] = px; y[0] = py; x[1] = xmin + (T)i * dx; y[1] = ymin + (T)j * dy; x[2] = xmin + (T)(i + 1) * dx; y[2] = ymin + (T)(j + 1) * dy;
        return RET(lin2(x, y, s));
    }
    const unsigned i = idx_u32(small + (double)((px - xmin) / dx));
    const unsigned j = idx_u32(small + (double)((py - ymin) / dy));
    const unsigned k = idx_u32(small + (double)((pz - zmin) / dz));
    s[0] = SN(i, j, k); s[1] = SN(i, j, k + 1); s[2] = SN(i, j + 1, k); s[3] = SN(i, j + 1, k + 1);
    s[4] = SN(i + 1, j, k); s[5] = SN(i + 1, j, k + 1); s[6] = SN(i + 1, j + 1, k); s[7] = SN(i + 1, j + 1, k + 1);
    x[0] = px; y[0] = py; z[0] = pz;
    x[1] = xmin + (T)i * dx; y[1] = ymin + (T)j * dy; z[1] = zmin + (T)k * dz;
    x[2] = xmin + (T)(i + 1) * dx; y[2] = ymin + (T)(j + 1) * dy; z[2] = zmin + (T)(k + 1) * dz;
    const T v = (s[0] * (x[2] - x[0]) * (y[2] - y[0]) * (z[2] - z[0]) + s[1] * (x[2] - x[0]) * (y[2] - y[0]) * (z[0] - z[1]) +
                 s[2] * (x[2] - x[0]) * (y[0] - y[1]) * (z[2] - z[0]) + s[3] * (x[2] - x[0]) * (y[0] - y[1]) * (z[0] - z[1]) +
                 s[4] * (x[0] - x[1]) * (y[2] - y[0]) * (z[2] - z[0]) + s[5] * (x[0] - x[1]) * (y[2] - y[0]) * (z[0] - z[1]) +
                 s[6] * (x[0] - x[1]) * (y[0] - y[1]) * (z[2] - z[0]) + s[7] * (x[0] - x[1]) * (y[0] - y[1]) * (z[0] - z[1])) /
                ((x[2] - x[1]) * (y[2] - y[1]) * (z[2] - z[1]));
    return RET(v);
}

template <typename T>
__device__ void grad3d(const RayGeom<T>& g, const T* __restrict__ Tn, int ts, T ptx, T pty, T ptz, T* gv) {
    const T k1 = (T)(1. / 24.), k2 = (T)(9. / 8.);
    const T dx = g.dx;
    auto TT = [&](T a, T b, T c) { return interp3d_pt(Tn, ts, a, b, c, g.nnx, g.nny, g.nnz, dx, g.xmin, g.ymin, g.zmin); };
    auto pts4 = [&](T p1, T cmin, T cmax, T& o1, T& o2, T& o3, T& o4) {
        T p2 = (T)((double)p1 + 0.5 * (double)dx), p3 = (T)((double)p1 + 1.5 * (double)dx), p4 = (T)((double)p1 + 2.0 * (double)dx);
        if (p1 <= cmin) {
            p1 = cmin;
            p2 = (T)((double)p1 + 0.5 * (double)dx); p3 = (T)((double)p1 + 1.5 * (double)dx); p4 = (T)((double)p1 + 2.0 * (double)dx);
        } else if (p4 >= cmax) {
            p4 = cmax;
            p3 = (T)((double)p4 - 0.5 * (double)dx); p2 = (T)((double)p4 - 1.5 * (double)dx); p1 = (T)((double)p4 - 2.0 * (double)dx);
        }
        o1 = p1; o2 = p2; o3 = p3; o4 = p4;
    };
    T p1, p2, p3, p4;
    pts4(ptx - dx, g.xmin, g.xmax, p1, p2, p3, p4);
    gv[0] = (k1 * TT(p1, pty, ptz) - k2 * TT(p2, pty, ptz) + k2 * TT(p3, pty, ptz) - k1 * TT(p4, pty, ptz)) / dx;
    pts4((T)((double)pty - (double)dx / 2.0), g.ymin, g.ymax, p1, p2, p3, p4);
    gv[1] = (k1 * TT(ptx, p1, ptz) - k2 * TT(ptx, p2, ptz) + k2 * TT(ptx, p3, ptz) - k1 * TT(ptx, p4, ptz)) / dx;
    pts4((T)((double)ptz - (double)dx / 2.0), g.zmin, g.zmax, p1, p2, p3, p4);
    gv[2] = (k1 * TT(ptx, pty, p1) - k2 * TT(ptx, pty, p2) + k2 * TT(ptx, pty, p3) - k1 * TT(ptx, pty, p4)) / dx;
}

template <typename T>
__device__ __forceinline__ int sgn_boost(T v) { return v == 0 ? 0 : (__builtin_signbit(v) ? -1 : 1); }

template <typename T>
__device__ __forceinline__ T dist3(const T* a, const T* b) {
    const T d2 = (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
    return (T)__builtin_sqrt((double)d2);
}

template <typename T>
__device__ void step_to_plane(const RayGeom<T>& g, T* cur, const T* gv) {
    const double small2 = 1.e-4 * 1.e-4;
    const T dx = g.dx;
    const long i = (long)(small2 + (double)((cur[0] - g.xmin) / dx));
    const long j = (long)(small2 + (double)((cur[1] - g.ymin) / dx));
    const long k = (long)(small2 + (double)((cur[2] - g.zmin) / dx));
    T xp = (T)((double)g.xmin + (double)dx * ((double)i + (sgn_boost(gv[0]) > 0 ? 1.0 : 0.0)));
    T yp = (T)((double)g.ymin + (double)dx * ((double)j + (sgn_boost(gv[1]) > 0 ? 1.0 : 0.0)));
    T zp = (T)((double)g.zmin + (double)dx * ((double)k + (sgn_boost(gv[2]) > 0 ? 1.0 : 0.0)));
    if ((double)rabs(xp - cur[0]) < small2) xp += dx * (T)sgn_boost(gv[0]);
    if ((double)rabs(yp - cur[1]) < small2) yp += dx * (T)sgn_boost(gv[1]);
    if ((double)rabs(zp - cur[2]) < small2) zp += dx * (T)sgn_boost(gv[2]);
    const T big = real_traits<T>::max();
    const T tx = gv[0] != 0 ? (xp - cur[0]) / gv[0] : big;
    const T ty = gv[1] != 0 ? (yp - cur[1]) / gv[1] : big;
    const T tz = gv[2] != 0 ? (zp - cur[2]) / gv[2] : big;
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

// status: 0 ok, 1 the ray left the grid (the reference throws), 2 step limit (the reference would not
// return), 3 (RAYS) more than `cap` points -- npts still counts them, the host retries with room.
// RAYS = false: Grid3Drn::getTraveltimeFromRaypath (ttcr/Grid3Drn.h:1103-1243).
// RAYS = true:  Grid3Drn::getRaypath(Tx, t0, Rx, r_data, tt, threadNo) (ttcr/Grid3Drn.h:1339-1500): the same
//               walk, every point recorded in pts[r][cap][3]; `back` is r_data.back() there, prev_pt here.
template <typename T, bool RAYS>
__global__ void fsm_raypath3d(const T* __restrict__ Tn, int ts, const T* __restrict__ sn, RayGeom<T> g, int n_src,
                              const T* __restrict__ src, const T* __restrict__ t0, const T* __restrict__ rcv, int n_rcv,
                              T* __restrict__ out, int* __restrict__ status, long max_steps, T* __restrict__ pts, long cap,
                              int* __restrict__ npts, const RaySrc* __restrict__ batch = nullptr,
                              const int* __restrict__ rx_src = nullptr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rcv) return;
    if (batch) {
        const RaySrc b = batch[rx_src[r]];
        Tn += b.tt_off; src += 3 * (size_t)b.tx_off; t0 += b.tx_off; n_src = b.n_tx;
    }
    const T rx[3] = {rcv[3 * r], rcv[3 * r + 1], rcv[3 * r + 2]};
    status[r] = 0;
    long np = 0;
    T* my_pts = RAYS ? pts + (size_t)r * cap * 3 : nullptr;
    T back[3] = {rx[0], rx[1], rx[2]}, cur[3] = {rx[0], rx[1], rx[2]}, gv[3];
    auto push = [&](const T* p) {
        if (RAYS) {
            if (np < cap) { my_pts[3 * np] = p[0]; my_pts[3 * np + 1] = p[1]; my_pts[3 * np + 2] = p[2]; }
            back[0] = p[0]; back[1] = p[1]; back[2] = p[2];
            ++np;
        }
    };
    auto finish = [&](int st, T tt) {
        if (RAYS) { npts[r] = (int)np; if (st == 0 && np > cap) st = 3; }
        status[r] = st;
        out[r] = tt;
    };
    push(rx);
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[3 * ns] && rx[1] == src[3 * ns + 1] && rx[2] == src[3 * ns + 2]) { finish(0, t0[ns]); return; }
    T tt = 0, s1, s2;
    s1 = slowness_at3d(g, sn, cur[0], cur[1], cur[2]);
    const T dx = g.dx;
    const T maxDist = (T)__builtin_sqrt((double)(dx * dx + dx * dx + dx * dx));
    bool reached = false;
    long steps = 0;
    while (!reached) {
        if (++steps > max_steps) { finish(2, tt); return; }
        grad3d(g, Tn, ts, cur[0], cur[1], cur[2], gv);
        gv[0] *= (T)-1.0; gv[1] *= (T)-1.0; gv[2] *= (T)-1.0;
        step_to_plane(g, cur, gv);
        if (cur[0] < g.xmin || cur[0] > g.xmax || cur[1] < g.ymin || cur[1] > g.ymax || cur[2] < g.zmin || cur[2] > g.zmax) {
            finish(1, tt); return;
        }
        s2 = slowness_at3d(g, sn, cur[0], cur[1], cur[2]);
        tt = (T)((double)tt + (0.5 * (double)(s1 + s2)) * (double)dist3(back, cur));
        s1 = s2;
        if (RAYS) push(cur); else { back[0] = cur[0]; back[1] = cur[1]; back[2] = cur[2]; }
        for (int ns = 0; ns < n_src; ++ns) {
            const T tx[3] = {src[3 * ns], src[3 * ns + 1], src[3 * ns + 2]};
            const T dist = dist3(cur, tx);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1]; gv[2] = tx[2] - cur[2];
                step_to_plane(g, cur, gv);
                if (dist3(cur, back) > dist || (cur[0] == tx[0] && cur[1] == tx[1] && cur[2] == tx[2])) {
                    s2 = slowness_at3d(g, sn, tx[0], tx[1], tx[2]);
                    tt = (T)((double)tt + ((double)t0[ns] + (0.5 * (double)(s1 + s2)) * (double)dist3(back, tx)));
                    push(tx);
                } else {
                    s2 = slowness_at3d(g, sn, cur[0], cur[1], cur[2]);
                    tt = (T)((double)tt + (0.5 * (double)(s1 + s2)) * (double)dist3(back, cur));
                    push(cur);
                    s1 = s2;
                    s2 = slowness_at3d(g, sn, tx[0], tx[1], tx[2]);
                    tt = (T)((double)tt + ((double)t0[ns] + (0.5 * (double)(s1 + s2)) * (double)dist3(cur, tx)));
                    push(tx);
                }
                reached = true;
            }
        }
    }
    finish(0, tt);
}


// The walks of the two overloads that fill the matrix M (compute_M), one thread per receiver; per term block of the reference
// (the eight nodes around a segment's mid-point get -s^2 ds w) one record  mid[3], ds, s(mid)  in segs[r][cap][5] -- the host
// spreads them over the nodes in the reference's push order.
// BOTH = false: Grid3Drn::getRaypath(Tx, t0, Rx, m_data, RxNo, tt, threadNo) (ttcr/Grid3Drn.h:1503-1800).  prev_pt is
//               overwritten with curr_pt BEFORE mid-point and length are formed (:1590-1597): every step of the walk is a
//               zero-length segment at its end point; the end game measures from the last step point, for EVERY source point
//               within a cell diagonal (prev_pt does not move between them).
// BOTH = true:  Grid3Drn::getRaypath(Tx, t0, Rx, r_data, m_data, RxNo, tt, threadNo) (:2144-2470), what ttcrpy calls for
//               compute_M with return_rays: prev_pt = r_data.back() BEFORE the push, the segments carry their lengths -- except
//               the plane point between walk and Tx, read AFTER the push (:2359-2366).  Points as fsm_raypath3d<T, true>.
// A receiver on a source point: tt = 0 (not t0), no records.  status as fsm_raypath3d (3: more than cap records).
template <typename T, bool BOTH>
__global__ void fsm_raypath3d_m(const T* __restrict__ Tn, int ts, const T* __restrict__ sn, RayGeom<T> g, int n_src,
                                const T* __restrict__ src, const T* __restrict__ t0, const T* __restrict__ rcv, int n_rcv,
                                T* __restrict__ out, int* __restrict__ status, long max_steps, T* __restrict__ segs, long cap,
                                int* __restrict__ nsegs) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rcv) return;
    const T rx[3] = {rcv[3 * r], rcv[3 * r + 1], rcv[3 * r + 2]};
    long nsg = 0;
    T* my = segs + (size_t)r * cap * 5;
    auto seg = [&](const T* a, const T* b) {   // mid_pt = 0.5 * (a + b), ds = a.getDistance(b), s = computeSlowness(mid_pt, true)
        if (nsg < cap) {
            T* p = my + 5 * nsg;
            p[0] = (T)0.5 * (a[0] + b[0]); p[1] = (T)0.5 * (a[1] + b[1]); p[2] = (T)0.5 * (a[2] + b[2]);
            p[3] = dist3(a, b);
            p[4] = slowness_at3d(g, sn, p[0], p[1], p[2]);
        }
        ++nsg;
    };
    auto finish = [&](int st, T tt) {
        nsegs[r] = (int)nsg;
        if (st == 0 && nsg > cap) st = 3;
        status[r] = st;
        out[r] = tt;
    };
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[3 * ns] && rx[1] == src[3 * ns + 1] && rx[2] == src[3 * ns + 2]) { finish(0, (T)0); return; }
    T back[3] = {rx[0], rx[1], rx[2]}, cur[3] = {rx[0], rx[1], rx[2]}, gv[3];
    T tt = 0, s1, s2;
    s1 = slowness_at3d(g, sn, cur[0], cur[1], cur[2]);
    const T dx = g.dx;
    const T maxDist = (T)__builtin_sqrt((double)(dx * dx + dx * dx + dx * dx));
    bool reached = false;
    long steps = 0;
    auto set_back = [&](const T* p) { back[0] = p[0]; back[1] = p[1]; back[2] = p[2]; };
    while (!reached) {
        if (++steps > max_steps) { finish(2, tt); return; }
        grad3d(g, Tn, ts, cur[0], cur[1], cur[2], gv);
        gv[0] *= (T)-1.0; gv[1] *= (T)-1.0; gv[2] *= (T)-1.0;
        step_to_plane(g, cur, gv);
        if (cur[0] < g.xmin || cur[0] > g.xmax || cur[1] < g.ymin || cur[1] > g.ymax || cur[2] < g.zmin || cur[2] > g.zmax) {
            finish(1, tt); return;
        }
        s2 = slowness_at3d(g, sn, cur[0], cur[1], cur[2]);
        tt = (T)((double)tt + (0.5 * (double)(s1 + s2)) * (double)dist3(back, cur));
        s1 = s2;
        if (BOTH) { seg(cur, back); set_back(cur); } else { set_back(cur); seg(cur, back); }
        for (int ns = 0; ns < n_src; ++ns) {
            const T tx[3] = {src[3 * ns], src[3 * ns + 1], src[3 * ns + 2]};
            const T dist = dist3(cur, tx);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1]; gv[2] = tx[2] - cur[2];
                step_to_plane(g, cur, gv);
                if (dist3(cur, back) > dist || (cur[0] == tx[0] && cur[1] == tx[1] && cur[2] == tx[2])) {
                    s2 = slowness_at3d(g, sn, tx[0], tx[1], tx[2]);
                    tt = (T)((double)tt + ((double)t0[ns] + (0.5 * (double)(s1 + s2)) * (double)dist3(back, tx)));
                    seg(tx, back);
                    if (BOTH) set_back(tx);
                } else {
                    s2 = slowness_at3d(g, sn, cur[0], cur[1], cur[2]);
                    tt = (T)((double)tt + (0.5 * (double)(s1 + s2)) * (double)dist3(back, cur));
                    s1 = s2;
                    if (BOTH) set_back(cur);
                    seg(cur, back);
                    s2 = slowness_at3d(g, sn, tx[0], tx[1], tx[2]);
                    tt = (T)((double)tt + ((double)t0[ns] + (0.5 * (double)(s1 + s2)) * (double)dist3(cur, tx)));
                    seg(tx, cur);
                    if (BOTH) set_back(tx);
                }
                reached = true;
            }
        }
    }
    finish(0, tt);
}

// rays recorded in fixed-capacity rows -> one dense array (ray r occupies points [off[r], off[r+1]))
template <typename T>
__global__ void fsm_compact_rays(const T* __restrict__ pts, long cap, const long long* __restrict__ off,
                                 T* __restrict__ out, T ox, T oy, T oz) {
    const int r = blockIdx.x;
    const long long a = off[r];
    long long n = off[r + 1] - a;
    n = n > cap ? cap : n;   // (a ray longer than its row is traced again with room and copied by a launch of its own)
    const T* src = pts + (size_t)r * cap * 3;
    for (long long i = threadIdx.x; i < 3 * n; i += blockDim.x) {
        const int c = (int)(i % 3);
        out[3 * a + i] = src[i] + (c == 0 ? ox : (c == 1 ? oy : oz));
    }
}

// Grid2Drn::getTraveltime (ttcr/Grid2Drn.h:359-414); Grid2Drn::getSlowness (:1421-1476) has the same
// shape on the node slowness (stride 1)
template <typename T>
__device__ __forceinline__ T interp2d_pt(const T* __restrict__ Tn, int ts, int nnx, int nnz, T dx, T dz, T xmin, T zmin, T px, T pz) {
    const double small = 1.e-4;
    const uint32_t i = idx_u32(small + (double)((px - xmin) / dx));
    const uint32_t j = idx_u32(small + (double)((pz - zmin) / dz));
    auto ab = [](T v) { return v < 0 ? -v : v; };
    const bool onx = (double)ab(px - (xmin + (T)i * dx)) < small;
    const bool onz = (double)ab(pz - (zmin + (T)j * dz)) < small;
    // index = quotient + 1e-4 (cells), "on the line" = absolute distance below 1e-4: with dx > 1 a point between 1e-4
    // and 1e-4*dx below the last line gets the last node as lower index without being on it, and the reference reads
    // index+1.  Clamped to the last node, like the oracle's T2.
    auto cl = [](uint32_t v, int n) { return v < (uint32_t)n ? v : (uint32_t)n - 1u; };
#define T2(ii, jj) Tn[((size_t)cl(ii, nnx) * nnz + cl(jj, nnz)) * ts]
    T tt;
    if (onx && onz) {
        tt = T2(i, j);
    } else if (onx) {
        T t1 = T2(i, j), t2 = T2(i, j + 1);
        T w1 = (zmin + (T)(j + 1) * dz - pz) / dz, w2 = (pz - (zmin + (T)j * dz)) / dz;
        tt = t1 * w1 + t2 * w2;
    } else if (onz) {
        T t1 = T2(i, j), t2 = T2(i + 1, j);
        T w1 = (xmin + (T)(i + 1) * dx - px) / dx, w2 = (px - (xmin + (T)i * dx)) / dx;
        tt = t1 * w1 + t2 * w2;
    } else {
        T t1 = T2(i, j), t2 = T2(i + 1, j);
        T t3 = T2(i, j + 1), t4 = T2(i + 1, j + 1);
        T w1 = (xmin + (T)(i + 1) * dx - px) / dx, w2 = (px - (xmin + (T)i * dx)) / dx;
        t1 = t1 * w1 + t2 * w2;
        t2 = t3 * w1 + t4 * w2;
        w1 = (zmin + (T)(j + 1) * dz - pz) / dz;
        w2 = (pz - (zmin + (T)j * dz)) / dz;
        tt = t1 * w1 + t2 * w2;
    }
#undef T2
    return tt;
}

template <typename T>
__global__ void fsm_interp2d(const T* __restrict__ Tn, int ts, const T* __restrict__ pts, T* __restrict__ out, int n,
                             int nnx, int nnz, T dx, T dz, T xmin, T zmin) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    out[r] = interp2d_pt(Tn, ts, nnx, nnz, dx, dz, xmin, zmin, pts[2 * r], pts[2 * r + 1]);
}

template <typename T>
__global__ void fsm_interp2d_batch(const T* __restrict__ tt0, int ts, size_t n_nodes, const int* __restrict__ slot_of,
                                   const T* __restrict__ pts, T* __restrict__ out, int n, int nnx, int nnz, T dx, T dz, T xmin,
                                   T zmin) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int slot = slot_of[r];
    const T* Tn = tt0 + (size_t)(slot / ts) * n_nodes * ts + slot % ts;
    out[r] = interp2d_pt(Tn, ts, nnx, nnz, dx, dz, xmin, zmin, pts[2 * r], pts[2 * r + 1]);
}

// ---- 2-D raypath family: Grid2Drn::getTraveltimeFromRaypath (ttcr/Grid2Drn.h:1478-1661) and
// Grid2Drn::getRaypath(Tx, t0, Rx, r_data, tt, threadNo) (:1663-1850) -------------------------------
template <typename T>
struct RayGeom2 {
    int nnx, nnz;
    T dx, dz, xmin, zmin, xmax, zmax;
};

// Grid2Drn::computeSlowness(pt), ttcr/Grid2Drn.h:262-330: node / linear / bilinear interpolation of the node slowness
// (FSM node and cell grids alike); on-line = first node within small^2, cell index = quotient + small (clamped like SN)
template <typename T>
__device__ T slowness_at2d(const RayGeom2<T>& g, const T* __restrict__ sn, T px, T pz) {
    const double small = 1.e-4;
    const int onX = on_node(px, g.xmin, g.dx, g.nnx), onZ = on_node(pz, g.zmin, g.dz, g.nnz);
    auto S2 = [&](unsigned i, unsigned k) {
        i = i < (unsigned)g.nnx ? i : (unsigned)g.nnx - 1;
        k = k < (unsigned)g.nnz ? k : (unsigned)g.nnz - 1;
        return sn[(size_t)i * g.nnz + k];
    };
    if (onX != -1 && onZ != -1) return sn[(size_t)onX * g.nnz + onZ];
    if (onX != -1) {
        const unsigned k = idx_u32(small + (double)((pz - g.zmin) / g.dz));
        const T s0 = S2(onX, k), s1 = S2(onX, k + 1);
        const T x0 = pz, x1 = g.zmin + (T)k * g.dz, x2 = g.zmin + (T)(k + 1) * g.dz;
        return (s0 * (x2 - x0) + s1 * (x0 - x1)) / (x2 - x1);
    }
    if (onZ != -1) {
        const unsigned i = idx_u32(small + (double)((px - g.xmin) / g.dx));
        const T s0 = S2(i, onZ), s1 = S2(i + 1, onZ);
        const T x0 = px, x1 = g.xmin + (T)i * g.dx, x2 = g.xmin + (T)(i + 1) * g.dx;
        return (s0 * (x2 - x0) + s1 * (x0 - x1)) / (x2 - x1);
    }
    const unsigned i = idx_u32(small + (double)((px - g.xmin) / g.dx));
    const unsigned k = idx_u32(small + (double)((pz - g.zmin) / g.dz));
    const T s0 = S2(i, k), s1 = S2(i, k + 1), s2 = S2(i + 1, k), s3 = S2(i + 1, k + 1);
    const T x0 = px, z0 = pz, x1 = g.xmin + (T)i * g.dx, z1 = g.zmin + (T)k * g.dz;
    const T x2 = g.xmin + (T)(i + 1) * g.dx, z2 = g.zmin + (T)(k + 1) * g.dz;
    return (s0 * (x2 - x0) * (z2 - z0) + s1 * (x2 - x0) * (z0 - z1) + s2 * (x0 - x1) * (z2 - z0) + s3 * (x0 - x1) * (z0 - z1)) /
           ((x2 - x1) * (z2 - z1));
}

// Grid3D::computeSlowness / Grid2D::computeSlowness at n points (get_s0 of the Python classes)
template <typename T>
__global__ void fsm_compute_slowness3d(RayGeom<T> g, const T* __restrict__ sn, const T* __restrict__ pts, int n, T* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = slowness_at3d(g, sn, pts[3 * r], pts[3 * r + 1], pts[3 * r + 2]);
}
template <typename T>
__global__ void fsm_compute_slowness2d(RayGeom2<T> g, const T* __restrict__ sn, const T* __restrict__ pts, int n, T* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = slowness_at2d(g, sn, pts[2 * r], pts[2 * r + 1]);
}

// Grid2Drn::grad(g, pt, nt), ttcr/Grid2Drn.h:606-632
template <typename T>
__device__ void grad2d(const RayGeom2<T>& g, const T* __restrict__ Tn, int ts, T px, T pz, T* gv) {
    T p1 = (T)((double)px - (double)g.dx / 2.0);
    if (p1 < g.xmin) p1 = g.xmin;
    T p2 = p1 + g.dx;
    if (p2 > g.xmax) {
        p2 = g.xmax;
        p1 = g.xmax - g.dx;
    }
    gv[0] = (interp2d_pt(Tn, ts, g.nnx, g.nnz, g.dx, g.dz, g.xmin, g.zmin, p2, pz) -
             interp2d_pt(Tn, ts, g.nnx, g.nnz, g.dx, g.dz, g.xmin, g.zmin, p1, pz)) / g.dx;
    p1 = (T)((double)pz - (double)g.dz / 2.0);
    if (p1 < g.zmin) p1 = g.zmin;
    p2 = p1 + g.dz;
    if (p2 > g.zmax) {
        p2 = g.zmax;
        p1 = g.zmax - g.dz;
    }
    gv[1] = (interp2d_pt(Tn, ts, g.nnx, g.nnz, g.dx, g.dz, g.xmin, g.zmin, px, p2) -
             interp2d_pt(Tn, ts, g.nnx, g.nnz, g.dx, g.dz, g.xmin, g.zmin, px, p1)) / g.dz;
}

// Grid2Drn::getCellNo, ttcr/Grid2Drn.h:170-176
template <typename T>
__device__ uint32_t cellno2d(const RayGeom2<T>& g, T px, T pz) {
    const double small = 1.e-4;
    const T x = (double)(g.xmax - px) < small ? (T)((double)g.xmax - .5 * (double)g.dx) : px;
    const T z = (double)(g.zmax - pz) < small ? (T)((double)g.zmax - .5 * (double)g.dz) : pz;
    uint32_t nx = idx_u32(small + (double)((x - g.xmin) / g.dx));
    uint32_t nz = idx_u32(small + (double)((z - g.zmin) / g.dz));
    // (absolute test above, relative index here: a cell index past the last cell -- dx > 1 -- is clamped, see the oracle)
    nx = nx > (uint32_t)(g.nnx - 2) ? (uint32_t)(g.nnx - 2) : nx;
    nz = nz > (uint32_t)(g.nnz - 2) ? (uint32_t)(g.nnz - 2) : nz;
    return nx * (uint32_t)(g.nnz - 1) + nz;
}

// advance cur along gv to the next grid line of cell (i,k); (i,k) is taken once per walk step, the
// reference does not refresh it for the retry along the face or for the last hop to the source
template <typename T>
__device__ void step2d(const RayGeom2<T>& g, long i, long k, T* cur, const T* gv) {
    const double small = 1.e-4;
    T xp = (T)((double)g.xmin + (double)g.dx * ((double)i + (sgn_boost(gv[0]) > 0 ? 1.0 : 0.0)));
    T zp = (T)((double)g.zmin + (double)g.dz * ((double)k + (sgn_boost(gv[1]) > 0 ? 1.0 : 0.0)));
    if ((double)rabs(xp - cur[0]) < small) xp += g.dx * (T)sgn_boost(gv[0]);
    if ((double)rabs(zp - cur[1]) < small) zp += g.dz * (T)sgn_boost(gv[1]);
    const T big = real_traits<T>::max();
    const T tx = gv[0] != 0 ? (xp - cur[0]) / gv[0] : big;
    const T tz = gv[1] != 0 ? (zp - cur[1]) / gv[1] : big;
    if (tx < tz) {
        cur[0] += tx * gv[0]; cur[1] += tx * gv[1];
        cur[0] = xp;
    } else {
        cur[0] += tz * gv[0]; cur[1] += tz * gv[1];
        cur[1] = zp;
    }
}

template <typename T>
__device__ __forceinline__ T dist2(const T* a, const T* b) {
    return (T)__builtin_sqrt((double)((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1])));
}

// sn: node slowness; sc: the cell slowness a Grid2Drcfs keeps (hasCellSlowness, ttcr/Grid2Drcfs.h:62-68) or null.
// status as in fsm_raypath3d (1: the ray left the grid twice in one step, the reference throws).
template <typename T, bool RAYS>
__global__ void fsm_raypath2d(const T* __restrict__ Tn, int ts, const T* __restrict__ sn, const T* __restrict__ sc,
                              RayGeom2<T> g, int n_src, const T* __restrict__ src, const T* __restrict__ t0,
                              const T* __restrict__ rcv, int n_rcv, T* __restrict__ out, int* __restrict__ status,
                              long max_steps, T* __restrict__ pts, long cap, int* __restrict__ npts,
                              const RaySrc* __restrict__ batch = nullptr, const int* __restrict__ rx_src = nullptr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rcv) return;
    if (batch) {
        const RaySrc b = batch[rx_src[r]];
        Tn += b.tt_off; src += 2 * (size_t)b.tx_off; t0 += b.tx_off; n_src = b.n_tx;
    }
    const T rx[2] = {rcv[2 * r], rcv[2 * r + 1]};
    status[r] = 0;
    long np = 0;
    T* my_pts = RAYS ? pts + (size_t)r * cap * 2 : nullptr;
    T back[2] = {rx[0], rx[1]}, cur[2] = {rx[0], rx[1]}, gv[2];
    auto push = [&](const T* p) {
        if (RAYS) {
            if (np < cap) { my_pts[2 * np] = p[0]; my_pts[2 * np + 1] = p[1]; }
            back[0] = p[0]; back[1] = p[1];
            ++np;
        }
    };
    auto finish = [&](int st, T tt) {
        if (RAYS) { npts[r] = (int)np; if (st == 0 && np > cap) st = 3; }
        status[r] = st;
        out[r] = tt;
    };
    auto slow = [&](T px, T pz) { return interp2d_pt(sn, 1, g.nnx, g.nnz, g.dx, g.dz, g.xmin, g.zmin, px, pz); };
    push(rx);
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[2 * ns] && rx[1] == src[2 * ns + 1]) { finish(0, t0[ns]); return; }
    T tt = 0, s1 = 0, s2 = 0, slown = 0;
    if (!sc) s1 = slow(cur[0], cur[1]);
    const T maxDist = (T)__builtin_sqrt((double)(g.dx * g.dx + g.dz * g.dz));
    bool reached = false;
    long steps = 0;
    while (!reached) {
        if (++steps > max_steps) { finish(2, tt); return; }
        grad2d(g, Tn, ts, cur[0], cur[1], gv);
        gv[0] *= (T)-1.0; gv[1] *= (T)-1.0;
        const double small = 1.e-4;
        const long i = (long)(small + (double)((cur[0] - g.xmin) / g.dx));
        const long k = (long)(small + (double)((cur[1] - g.zmin) / g.dz));
        step2d(g, i, k, cur, gv);
        if (cur[0] < g.xmin || cur[0] > g.xmax || cur[1] < g.zmin || cur[1] > g.zmax) {
            // going outside: follow the face instead (:1536-1581)
            // (the reference's unqualified abs() binds to the C int overload: components truncated first, :1538)
            const int agx = (int)gv[0], agz = (int)gv[1];
            if ((agx < 0 ? -agx : agx) > (agz < 0 ? -agz : agz)) { gv[0] = (T)sgn_boost(gv[0]); gv[1] = 0; }
            else { gv[1] = (T)sgn_boost(gv[1]); gv[0] = 0; }
            cur[0] = back[0]; cur[1] = back[1];
            step2d(g, i, k, cur, gv);
            if (cur[0] < g.xmin || cur[0] > g.xmax || cur[1] < g.zmin || cur[1] > g.zmax) { finish(1, tt); return; }
        }
        if (sc) {
            const T mx = (T)0.5 * (back[0] + cur[0]), mz = (T)0.5 * (back[1] + cur[1]);
            slown = sc[cellno2d(g, mx, mz)];
        } else {
            s2 = slow(cur[0], cur[1]);
            slown = (T)(0.5 * (double)(s1 + s2));
            s1 = s2;
        }
        tt += slown * dist2(back, cur);
        if (RAYS) push(cur); else { back[0] = cur[0]; back[1] = cur[1]; }
        for (int ns = 0; ns < n_src; ++ns) {
            const T tx[2] = {src[2 * ns], src[2 * ns + 1]};
            const T dist = dist2(cur, tx);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1];
                step2d(g, i, k, cur, gv);
                if (dist2(cur, back) > dist || (cur[0] == tx[0] && cur[1] == tx[1])) {
                    if (sc) {
                        const T mx = (T)0.5 * (back[0] + tx[0]), mz = (T)0.5 * (back[1] + tx[1]);
                        slown = sc[cellno2d(g, mx, mz)];
                    } else {
                        s2 = slow(tx[0], tx[1]);
                        slown = (T)(0.5 * (double)(s1 + s2));
                    }
                    tt += slown * dist2(back, tx);
                    push(tx);
                } else if (sc) {
                    T mx = (T)0.5 * (back[0] + cur[0]), mz = (T)0.5 * (back[1] + cur[1]);
                    slown = sc[cellno2d(g, mx, mz)];
                    tt += slown * dist2(back, cur);
                    mx = (T)0.5 * (cur[0] + tx[0]); mz = (T)0.5 * (cur[1] + tx[1]);
                    slown = sc[cellno2d(g, mx, mz)];
                    tt += slown * dist2(cur, tx);
                    // (the reference records neither point in this branch, :1826-1831)
                } else {
                    s2 = slow(cur[0], cur[1]);
                    tt = (T)((double)tt + (0.5 * (double)(s1 + s2)) * (double)dist2(back, cur));
                    push(cur);
                    s1 = s2;
                    s2 = slow(tx[0], tx[1]);
                    tt = (T)((double)tt + (0.5 * (double)(s1 + s2)) * (double)dist2(cur, tx));
                    push(tx);
                }
                tt += t0[ns];
                reached = true;
            }
        }
    }
    finish(0, tt);
}

// Grid2Drn::getRaypath with l_data -- the ray-projection matrix L of a cell grid (compute_L of ttcrpy):
//   RAYS = true : getRaypath(Tx, t0, Rx, r_data, l_data, tt, threadNo), ttcr/Grid2Drn.h:1852-2021
//   RAYS = false: getRaypath(Tx, t0, Rx, l_data, tt, threadNo), :2023-2190
// A walk of its own, not the one of fsm_raypath2d: a step that leaves the grid is an error at once (no second try along the
// face), every segment is booked to the cell of its mid-point (cell index, length) in push order, and the two overloads
// differ where the reference does: without r_data the traveltime of the last hop is slowness x the ENTRY's value, which
// is the sum of the two last segments when both lie in one cell (:2168-2181) -- that entry is pushed beside the entry of
// the segment before it, not instead of it.  sc: the cell slowness (a Grid2Drcfs); sn is used when sc is null.
// status 1: outside the grid, 2: step limit, 3: a row overflowed (np / nl hold what was needed).
template <typename T, bool RAYS>
__global__ void fsm_raypath2d_l(const T* __restrict__ Tn, int ts, const T* __restrict__ sn, const T* __restrict__ sc, RayGeom2<T> g, int n_src,
                                const T* __restrict__ src, const T* __restrict__ t0, const T* __restrict__ rcv, int n_rcv,
                                T* __restrict__ out, int* __restrict__ status, long max_steps, T* __restrict__ pts, long cap,
                                int* __restrict__ npts, uint32_t* __restrict__ lcell, T* __restrict__ lval, long lcap, int* __restrict__ nlen) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rcv) return;
    const T rx[2] = {rcv[2 * r], rcv[2 * r + 1]};
    long np = 0, nl = 0;
    T* my_pts = RAYS ? pts + (size_t)r * cap * 2 : nullptr;
    uint32_t* my_c = lcell + (size_t)r * lcap;
    T* my_v = lval + (size_t)r * lcap;
    T back[2] = {rx[0], rx[1]}, cur[2] = {rx[0], rx[1]}, gv[2];
    auto push = [&](const T* p) {   // r_data.push_back (RAYS) / prev_pt = (the other overload)
        if (RAYS) { if (np < cap) { my_pts[2 * np] = p[0]; my_pts[2 * np + 1] = p[1]; } ++np; }
        back[0] = p[0]; back[1] = p[1];
    };
    auto book = [&](uint32_t c, T v) { if (nl < lcap) { my_c[nl] = c; my_v[nl] = v; } ++nl; };
    auto finish = [&](int st, T tt) {
        if (RAYS) npts[r] = (int)np;
        nlen[r] = (int)nl;
        if (st == 0 && ((RAYS && np > cap) || nl > lcap)) st = 3;
        status[r] = st;
        out[r] = tt;
    };
    auto slow = [&](T px, T pz) { return interp2d_pt(sn, 1, g.nnx, g.nnz, g.dx, g.dz, g.xmin, g.zmin, px, pz); };
    if (RAYS) { if (np < cap) { my_pts[0] = rx[0]; my_pts[1] = rx[1]; } ++np; }
    for (int ns = 0; ns < n_src; ++ns)
        if (rx[0] == src[2 * ns] && rx[1] == src[2 * ns + 1]) { finish(0, t0[ns]); return; }
    T tt = 0, s1 = 0, s2 = 0, slown = 0;
    if (!sc) s1 = slow(cur[0], cur[1]);
    const T maxDist = (T)__builtin_sqrt((double)(g.dx * g.dx + g.dz * g.dz));
    bool reached = false;
    long steps = 0;
    while (!reached) {
        if (++steps > max_steps) { finish(2, tt); return; }
        grad2d(g, Tn, ts, cur[0], cur[1], gv);
        gv[0] *= (T)-1.0; gv[1] *= (T)-1.0;
        const double small = 1.e-4;
        const long i = (long)(small + (double)((cur[0] - g.xmin) / g.dx));
        const long k = (long)(small + (double)((cur[1] - g.zmin) / g.dz));
        step2d(g, i, k, cur, gv);
        if (cur[0] < g.xmin || cur[0] > g.xmax || cur[1] < g.zmin || cur[1] > g.zmax) { finish(1, tt); return; }
        {
            const T mx = (T)0.5 * (back[0] + cur[0]), mz = (T)0.5 * (back[1] + cur[1]);
            const uint32_t c = cellno2d(g, mx, mz);
            const T v = dist2(cur, back);
            book(c, v);
            if (sc) slown = sc[c];
            else { s2 = slow(cur[0], cur[1]); slown = (T)(0.5 * (double)(s1 + s2)); s1 = s2; }
            tt += slown * v;
            push(cur);
        }
        for (int ns = 0; ns < n_src; ++ns) {
            const T tx[2] = {src[2 * ns], src[2 * ns + 1]};
            const T dist = dist2(cur, tx);
            if (dist < maxDist) {
                gv[0] = tx[0] - cur[0]; gv[1] = tx[1] - cur[1];
                step2d(g, i, k, cur, gv);
                if (dist2(cur, back) > dist || (cur[0] == tx[0] && cur[1] == tx[1])) {   // no intersection, or arrived
                    const uint32_t c = cellno2d(g, tx[0], tx[1]);
                    const T v = dist2(tx, back);
                    book(c, v);
                    if (sc) slown = sc[c];
                    else { s2 = slow(tx[0], tx[1]); slown = (T)(0.5 * (double)(s1 + s2)); }
                    tt += slown * v;
                    if (RAYS) push(tx);
                } else {
                    // to the intersection ...
                    const T mx = (T)0.5 * (back[0] + cur[0]), mz = (T)0.5 * (back[1] + cur[1]);
                    uint32_t c = cellno2d(g, mx, mz);
                    T v = dist2(cur, back);
                    book(c, v);
                    if (sc) slown = sc[c];
                    else { s2 = slow(cur[0], cur[1]); slown = (T)(0.5 * (double)(s1 + s2)); s1 = s2; }
                    tt += slown * v;
                    push(cur);
                    // ... and on to the source point: the entry is kept and extended when the cell is the same
                    const uint32_t c2 = cellno2d(g, tx[0], tx[1]);
                    const T hop = dist2(tx, back);
                    if (c == c2) v += hop; else { c = c2; v = hop; }
                    book(c, v);
                    if (sc) slown = sc[c];
                    else { s2 = slow(tx[0], tx[1]); slown = (T)(0.5 * (double)(s1 + s2)); }
                    tt += slown * (RAYS ? hop : v);   // (with r_data: the hop, :2009; without: the entry's value, :2181)
                    if (RAYS) push(tx);
                }
                tt += t0[ns];
                reached = true;
            }
        }
    }
    finish(0, tt);
}

// rays recorded in fixed-capacity rows -> one dense array, 2-D (x, z) pairs
template <typename T>
__global__ void fsm_compact_rays2(const T* __restrict__ pts, long cap, const long long* __restrict__ off, T* __restrict__ out) {
    const int r = blockIdx.x;
    const long long a = off[r];
    long long n = off[r + 1] - a;
    n = n > cap ? cap : n;
    const T* src = pts + (size_t)r * cap * 2;
    for (long long i = threadIdx.x; i < 2 * n; i += blockDim.x) out[2 * a + i] = src[i];
}

}  // namespace ttcr_amd
