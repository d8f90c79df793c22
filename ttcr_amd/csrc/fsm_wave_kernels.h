// ttcr_amd/csrc/fsm_wave_kernels.h -- first-order 3-D sweeps with ONE WAVEFRONT per work unit (fp32 grids, one field per slot).
//
// What it computes: Grid3Drn::sweep + update_node (ttcr/Grid3Drn.h:2816-2959), the same partial order and the same
// arithmetic as fsm_sweep_persistent (fsm_kernels.h): level L = i' + j' + k' of the oriented indices, any linear extension of
// the sweep's partial order gives the serial Gauss-Seidel result bit for bit.
//
// Why a second kernel: a lone source (and every small batch) is bound by the dependent chain of a level, not by memory
// (DESIGN.md 8a-3).  In the four-wave patches of fsm_sweep_persistent a level is: LDS write -> barrier -> LDS read -> ~45
// dependent vector instructions, with the four waves waiting for the slowest at every level, and a third of all
// instructions go into staging the tile through LDS.  Here:
//   * a unit is a patch of 16 (J) x 4 PKR (K) columns marched by ONE wavefront: lane = (tj = lane & 15, g = lane >> 4), a lane
//     owns the PKR columns k' = k0 + g PKR + r.  No workgroup barrier exists, the four SIMDs of a CU run four different units.
//   * per level a lane updates PKR nodes that do not depend on each other: their fp64 discriminant chains interleave in the
//     one instruction stream (the latency of a level is hidden by the level itself).
//   * neighbours: F -- the lane's own registers (a column is walked along the contiguous axis); K inside a lane -- registers
//     of the neighbouring column; J -- DPP row shifts inside the 16-lane rows (the lanes the shift leaves out take the halo
//     value, the DPP `old` operand); K across the four lane rows -- ds_bpermute (no LDS memory), issued a level ahead.
//   * a column of a chunk is C consecutive nodes along F = C consecutive floats in HBM: every lane loads and stores its own
//     columns with 16-byte buffer accesses straight into / out of the registers it marches on (hardware bounds check: an
//     address outside the field reads 0 and a store there is dropped) -- no LDS staging, no transposition, no sheared copy
//     of the slowness.  Nodes outside the grid and frozen nodes get slowness +inf when a chunk is loaded: their update is
//     +inf / NaN and never accepted, so the level march carries no mask at all.
// Scheduling is that of the whole-iteration launch: units (direction, patch) drawn from a ticket counter in
// an order in which a unit comes after everything it waits for; progress words per unit in HBM (launch epoch, sc1 stores /
// loads); a unit starts once the previous sweep has finished the <= 3 x 3 patches around it.
#pragma once
#include "fsm_kernels.h"
#ifndef FSM_WAVE_EXP
#define FSM_WAVE_EXP 0   // TIMING experiments (wrong results): 1: no unit waits for another one; 2: no drain of the stores before the next chunk
#endif

namespace ttcr_amd {

struct WaveArgs {
    float* tt;                // [n_slots][n_nodes] traveltime fields, natural order (x fastest)
    const float* s;           // node slowness, natural order
    const uint32_t* frozen;   // [n_slots][mask_words]
    const int* bbox;          // [n_slots][6]
    double* change;           // [n_slots]
    const int* slots;         // [batch] slot solved by batch entry z, -1: converged
    unsigned long long* evals;  // [n_slots]
    const uint32_t* order;    // units (TJ | TK << 14 | dir << 28) in ticket order
    int* sync;                // [0..3] ticket counters, [4] abort, [8..] progress words (fsm_kernels.h, "launch epoch")
    const int* iter_ptr;      // [1]: launch sequence number
    int NF, NJ, NK, npj, npk, n_patches, batch;
    uint32_t n_nodes, mask_words;
    float dx;
    unsigned long long timeout_ticks;
};

typedef unsigned int wave_u4 __attribute__((ext_vector_type(4)));
constexpr int FSM_WAVE_DONE = 0x3fffffff;
constexpr int FSM_WAVE_GUARD = 16;   // elements the host keeps allocated in front of / behind the fields and the slowness

// value of lane - 1 / lane + 1 inside the 16-lane row; lane 0 / lane 15 of a row keeps `edge`
__device__ __forceinline__ float row_below(float x, float edge) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(edge), (int)__float_as_uint(x), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_above(float x, float edge) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(edge), (int)__float_as_uint(x), 0x101, 0xf, 0xf, false));
}

// The local solver of update3 (fsm_kernels.h; Grid3Drn::update_node, ttcr/Grid3Drn.h:2936-2956) for N nodes that do not depend on
// each other (the nodes a lane updates at one level).  Same formulas, same roundings, same two wave-uniform short cuts -- taken for
// the N nodes TOGETHER, so that between the branches the N fp64 discriminant chains are one straight piece of code the compiler
// interleaves: inside update3 every node is a chain of ~90 dependent instructions between two branches, and a lone wavefront
// issues one of them every 8-10 cycles.  Which branch is taken does not change a value: the 2-D root is exact wherever it is
// computed, and where it is not, every lane beyond the 1-D branch is provably 3-D (derivation at update3).
template <int N>
__device__ __forceinline__ void update3_multi(const float (&ax)[N], const float (&ay)[N], const float (&az)[N], const float (&s)[N], float dx,
                                              float (&t)[N]) {
    float a1[N], a2[N], a3[N], fh[N], t1[N];
    unsigned long long b1d[N], any1d = 0ull;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        a1[n] = __builtin_fminf(__builtin_fminf(ax[n], ay[n]), az[n]);
        a3[n] = __builtin_fmaxf(__builtin_fmaxf(ax[n], ay[n]), az[n]);
        a2[n] = __builtin_amdgcn_fmed3f(ax[n], ay[n], az[n]);
        fh[n] = s[n] * dx;
        t1[n] = a1[n] + fh[n];
        t[n] = t1[n];
        b1d[n] = lanes_gt(t1[n], a2[n]);
        any1d |= b1d[n];
    }
    if (any1d != 0ull) {
        float t3[N], s12[N];
        unsigned long long need2 = 0ull;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const double d1 = a1[n], d2 = a2[n], d3 = a3[n], dfh = fh[n];
            double r = -d1 * d1;
            r = __builtin_fma(d1, d2, r);
            r = __builtin_fma(-d2, d2, r);
            r = __builtin_fma(d1, d3, r);
            r = __builtin_fma(d2, d3, r);
            r = __builtin_fma(-d3, d3, r);
            r = __builtin_fma(r, 2.0, (3.0 * dfh) * dfh);
            s12[n] = a1[n] + a2[n];
            const float s123 = s12[n] + a3[n];
            t3[n] = (float)((1. / 3.) * ((double)s123 + sqrt_disc_pos(r)));
            const float u = a3[n] - a1[n], v = a3[n] - a2[n];
            const float slack = fh[n] * fh[n] - (u * u + v * v);
            const float thr = 4e-6f * fh[n] * (__builtin_fabsf(a1[n]) + __builtin_fabsf(a3[n]) + fh[n]);
            need2 |= b1d[n] & ~lanes_gt(slack, thr);
        }
        if (need2 != 0ull) {
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const double dfh = fh[n];
                const float df = a1[n] - a2[n];
                const float df2 = df * df;
                const double disc2 = __builtin_fma(dfh * dfh, 2.0, -(double)df2);
                const float t2 = (float)(0.5 * ((double)s12[n] + sqrt_disc_pos(disc2)));
                t[n] = t1[n] > a2[n] ? (t2 > a3[n] ? t3[n] : t2) : t1[n];
            }
        } else {
#pragma unroll
            for (int n = 0; n < N; ++n) t[n] = t1[n] > a2[n] ? t3[n] : t1[n];
        }
    }
}

// The unit proper: direction `dir` of patch (TJ, TK) for batch entry z (slot `slot`).  RF = the F axis is swept downwards (the
// four values of a 16-byte access are then four levels in reverse order: fixed at compile time, registers cannot be indexed).
template <int PKR, int C, bool RF>
__device__ __forceinline__ void fsm_wave_body(const WaveArgs& a, float* lds, int lane, int e2, int dir, int TJ, int TK, int z, int slot) {
    constexpr int PJ = 16, PK = 4 * PKR;
    const float INF = __builtin_huge_valf();
    auto dec_prog = [&](int raw_) -> int { return (int)((unsigned)raw_ >> 30) == e2 ? (raw_ & 0x3fffffff) : 0; };
    auto ld_raw = [&](const int* p_) -> int { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st_prog = [&](int* p_, int v_) {
        __hip_atomic_store(p_, (int)(((unsigned)e2 << 30) | (unsigned)v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const int NF = a.NF, NJ = a.NJ, NK = a.NK, npj = a.npj;
    int* prog = a.sync + 8 + ((size_t)dir * a.batch + z) * a.n_patches;
    int* my_prog = prog + (TK * npj + TJ);
    constexpr int rf = RF ? 1 : 0;
    const int rj = (dir >> 1) & 1, rk = (dir >> 2) & 1;   // ttcr/Grid3Drn.h:2816-2899
    const int* up_j = (!(FSM_WAVE_EXP & 1) && TJ > 0) ? prog + (TK * npj + TJ - 1) : nullptr;
    const int* up_k = (!(FSM_WAVE_EXP & 1) && TK > 0) ? prog + ((TK - 1) * npj + TJ) : nullptr;

    const int tj = lane & 15, g = lane >> 4;
    const int j0 = TJ * PJ, k0 = TK * PK;
    const int jmaxp = (j0 + PJ < NJ ? j0 + PJ : NJ) - 1, kmaxp = (k0 + PK < NK ? k0 + PK : NK) - 1;
    const int Ls = j0 + k0, Le = jmaxp + kmaxp + NF - 1;
    const int m = TJ + TK;
    int Lc = Ls - (((Ls - m) % C + C) % C);   // chunk starts congruent to m modulo C (fsm_kernels.h)
    const int jp = j0 + tj;
    const int jn = rj ? NJ - 1 - jp : jp;
    const int sf = rf ? -1 : 1;
    // column r: oriented k' = k0 + g PKR + r.  csum = j' + k' (a column outside the grid: far negative, its i' never in range);
    // ebase = element index of level 0 (level L lies at ebase + sf L)
    int csum[PKR], ebase[PKR];
#pragma unroll
    for (int r = 0; r < PKR; ++r) {
        const int kp = k0 + g * PKR + r;
        const bool ok = jp < NJ && kp < NK;
        const int kn = rk ? NK - 1 - kp : kp;
        const int rowb = (int)(((uint32_t)kn * NJ + jn) * NF);
        csum[r] = ok ? jp + kp : -(1 << 29);
        ebase[r] = ok ? FSM_WAVE_GUARD + rowb + (rf ? NF - 1 + jp + kp : -(jp + kp)) : 0x3ffe0000;   // (not ok: every access out of the field)
    }
    // halo columns.  The J-upwind halo node lane tj = 0 needs at level q is the node of column j0 - 1 at level q - 1: the
    // same i' as the lane's own node, one row over; likewise J-downwind for tj = 15 (column j0 + 16 at level q + 1) and the
    // K halos for the rows g = 0 / g = 3.  So a halo access is the own column's access plus a constant element offset.
    const bool jup_ex = j0 > 0, jdn_ex = j0 + PJ < NJ, kup_ex = k0 > 0, kdn_ex = k0 + PK < NK;
    const int rowJ = rj ? -NF : NF;                 // element offset of column j' + 1
    const int rowK = (rk ? -1 : 1) * NJ * NF;       // ... of column k' + 1
    const bool jh_inf = (tj == 0 && !jup_ex) || (tj == PJ - 1 && !jdn_ex);
    const bool kh_inf = (g == 0 && !kup_ex) || (g == 3 && !kdn_ex);
    const bool edge_patch = !(jup_ex && jdn_ex && kup_ex && kdn_ex);

    // buffer descriptors: the field / the slowness array with FSM_WAVE_GUARD elements in front and behind (allocated by the
    // host, or the neighbouring field).  A 16-byte access may overhang its row by up to three elements at either end; at the
    // very start of the array that would be a negative byte offset, which wraps and fails the bounds check for the WHOLE
    // access -- the three valid values would read as 0.  With the guard every overhang is an in-range read of values nobody
    // uses; columns outside the grid still lie far out of range (reads 0, stores dropped).
    const uint32_t nbytes = (a.n_nodes + 2u * FSM_WAVE_GUARD) * 4u;
    __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(a.tt + (size_t)slot * a.n_nodes - FSM_WAVE_GUARD, 0, nbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.s) - FSM_WAVE_GUARD, 0, nbytes, 0x00020000);
    const uint32_t* __restrict__ Fz = a.frozen + (size_t)slot * a.mask_words;

    // chunks that may hold frozen nodes (fsm_kernels.h: near_lo / near_hi)
    int near_lo = 1, near_hi = 0;
    {
        const int* b6 = a.bbox + 6 * slot;
        const int jlo = rj ? NJ - 1 - jmaxp : j0, jhi = rj ? NJ - 1 - j0 : jmaxp;
        const int klo = rk ? NK - 1 - kmaxp : k0, khi = rk ? NK - 1 - k0 : kmaxp;
        const int b0 = rf ? NF - 1 - b6[1] : b6[0], b1 = rf ? NF - 1 - b6[0] : b6[1];
        const bool jk = !(jhi < b6[2] || jlo > b6[3] || khi < b6[4] || klo > b6[5]) && b0 <= NF - 1 && b1 >= 0;
        near_lo = jk ? b0 - (C - 1) + j0 + k0 : 1;
        near_hi = jk ? b1 + jmaxp + kmaxp : 0;
    }

    // ---- previous sweep: the patches (of ITS oriented partition) that own a column within 2 of ours have finished
    if (!(FSM_WAVE_EXP & 1) && dir > 0) {
        const int pd = dir - 1;
        const int prj = (pd >> 1) & 1, prk = (pd >> 2) & 1;
        int ja = j0 - 2, jb = jmaxp + 2, ka = k0 - 2, kb = kmaxp + 2;
        ja = ja < 0 ? 0 : ja; jb = jb > NJ - 1 ? NJ - 1 : jb;
        ka = ka < 0 ? 0 : ka; kb = kb > NK - 1 ? NK - 1 : kb;
        const int ja2 = (rj != prj) ? NJ - 1 - jb : ja, jb2 = (rj != prj) ? NJ - 1 - ja : jb;
        const int ka2 = (rk != prk) ? NK - 1 - kb : ka, kb2 = (rk != prk) ? NK - 1 - ka : kb;
        const int tja = ja2 / PJ, ntj = jb2 / PJ - tja + 1, tka = ka2 / PK, ntk = kb2 / PK - tka + 1;
        const int ia = lane & 3, ib = (lane >> 2) & 3;
        const int* pp = (lane < 16 && ia < ntj && ib < ntk)
                            ? a.sync + 8 + ((size_t)pd * a.batch + z) * a.n_patches + ((tka + ib) * npj + tja + ia) : nullptr;
        bool ok = !pp;
        unsigned long long t0 = 0;
        int spins = 0;
        for (;;) {
            if (!ok) ok = dec_prog(ld_raw(pp)) >= FSM_WAVE_DONE;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
            if (spins == 0) t0 = wall_clock64();
            if ((++spins & 63) == 0) {
                if (__builtin_amdgcn_readfirstlane(ld_raw(a.sync + 4))) break;
                if (wall_clock64() - t0 > a.timeout_ticks) {
                    if (lane == 0) __hip_atomic_store(a.sync + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }

    // ---- cooperative accesses.  A column of a chunk is C = 8 consecutive nodes along F = 32 bytes of one row; a wave-wide access
    // in which every lane fetches 16 bytes of its OWN column is 64 separate requests to the L2 (measured: the lone 512^3 source
    // was bound by the L2's request rate, 18 ms per sweep-iteration).  So TWO adjacent lanes fetch the two halves of one piece
    // (32 pieces per instruction, one 32-byte request each), and the pieces go through LDS to the lanes that march the
    // columns -- a private region of the wave, no barrier: LDS operations of one wave execute in order.
    //   piece pc = it * 32 + (lane >> 1)  <->  column (cj = pc & 15, ck = pc >> 4);  lane & 1 = which half (memory order)
    constexpr int NIT = 2 * PKR;                      // instructions per array and chunk (64 PKR pieces)
    static_assert(C == 8, "a piece is two 16-byte halves");
    // lds: [64 PKR * 8] traveltime pieces in, memory order | [64 PKR * 8] slowness pieces in, result pieces out | [2][32 * 8] halo
    // pieces, upwind then downwind
    float* const lds_t = lds;
    float* const lds_s = lds + 64 * PKR * 8;
    float (*const lds_h)[32 * 8] = (float (*)[32 * 8])(lds + 2 * 64 * PKR * 8);
    const int half = lane & 1;
    int cbase[NIT];   // element index of level 0 of the piece's column
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pc = it * 32 + (lane >> 1);
        const int jq = j0 + (pc & 15), kq = k0 + (pc >> 4);
        const bool ok = jq < NJ && kq < NK;
        const int rowb = (int)(((uint32_t)(rk ? NK - 1 - kq : kq) * NJ + (rj ? NJ - 1 - jq : jq)) * NF);
        cbase[it] = ok ? FSM_WAVE_GUARD + rowb + (rf ? NF - 1 + jq + kq : -(jq + kq)) : 0x3ffe0000;
    }
    // halo pieces: pair hp = lane >> 1.  hp < PK: the J halo of k' = k0 + hp (upwind: column j0 - 1, downwind: column j0 + 16);
    // PK <= hp < PK + 16: the K halo of j' = j0 + hp - PK (column k0 - 1 / k0 + PK).  A halo node has the i' of the patch's
    // node next to it, so a halo piece is that column's piece plus a constant element offset (see above).
    int hbase_up, hbase_dn;
    {
        const int hp = lane >> 1;
        const bool isj = hp < PK, isk = hp >= PK && hp < PK + PJ;
        const int cj_up = isj ? 0 : hp - PK, ck_up = isj ? hp : 0;
        const int cj_dn = isj ? PJ - 1 : hp - PK, ck_dn = isj ? hp : PK - 1;
        auto eb = [&](int cj, int ck) -> int {
            const int jq = j0 + cj, kq = k0 + ck;
            if (!(jq < NJ && kq < NK)) return 0x3ffe0000;
            const int rowb = (int)(((uint32_t)(rk ? NK - 1 - kq : kq) * NJ + (rj ? NJ - 1 - jq : jq)) * NF);
            return FSM_WAVE_GUARD + rowb + (rf ? NF - 1 + jq + kq : -(jq + kq));
        };
        hbase_up = (isj || isk) ? eb(cj_up, ck_up) - (isj ? rowJ : rowK) : 0x3ffe0000;
        hbase_dn = (isj || isk) ? eb(cj_dn, ck_dn) + (isj ? rowJ : rowK) : 0x3ffe0000;
        if ((isj && !jup_ex) || (isk && !kup_ex)) hbase_up = 0x3ffe0000;
        if ((isj && !jdn_ex) || (isk && !kdn_ex)) hbase_dn = 0x3ffe0000;
    }
    // byte offset of this lane's half of the piece of levels L .. L + 7 of the column whose level 0 lies at element `base`
    auto coff = [&](int base, int L) -> uint32_t {
        const int e = rf ? base - (L + 7) + 4 * half : base + L + 4 * half;
        return (uint32_t)e * 4u;
    };
    // LDS positions (float index): the lane's half in cooperative order; piece pc in a region
    const int lds_coop = lane * 4;
    // element m (memory order) of a piece <-> level q = rf ? 7 - m : m

    // ---- registers of a chunk
    float w[PKR][C], sv[PKR][C];      // traveltime at levels L0 .. L0+C-1 (old, then new) and slowness, per column
    float prev[PKR], fnx[PKR];        // result at level L0-1; old value at level L0+C
    float hJ[PKR][C], hK[C];          // halo values (lanes tj = 0 / 15: J halos; rows g = 0 / 3: K halos), see above
    wave_u4 wn[NIT], sn[NIT];         // next chunk as loaded: traveltime pieces of levels L0+C+1 .. L0+2C, slowness pieces of L0+C ..
    wave_u4 hun = {0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u}, hdn = hun;   // next chunk's halo pieces as loaded
    int hup_for = -(1 << 30), hdn_for = -(1 << 30);   // ... for the chunk that starts at this level
#pragma unroll
    for (int r = 0; r < PKR; ++r) { prev[r] = INF; fnx[r] = INF; }
    if (FSM_WAVE_EXP & 12) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) { wn[it] = hun; sn[it] = hun; }
    }

    // the traveltime window of a chunk is shifted by one level: the pieces hold levels L+1 .. L+8, the old value at level L is
    // the last value of the window before (nobody has touched it since: the chunk before ended at level L-1)
    auto issue_static = [&](int L) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (!(FSM_WAVE_EXP & 4)) wn[it] = __builtin_amdgcn_raw_buffer_load_b128(rsT, coff(cbase[it], L + 1), 0, 16);   // sc1: another XCD may have written it in this launch
            if (!(FSM_WAVE_EXP & 8)) sn[it] = __builtin_amdgcn_raw_buffer_load_b128(rsS, coff(cbase[it], L), 0, 0);
        }
    };
    auto issue_halo_up = [&](int L) { if (!(FSM_WAVE_EXP & 32)) hun = __builtin_amdgcn_raw_buffer_load_b128(rsT, coff(hbase_up, L), 0, 16); hup_for = L; };
    auto issue_halo_dn = [&](int L) { if (!(FSM_WAVE_EXP & 32)) hdn = __builtin_amdgcn_raw_buffer_load_b128(rsT, coff(hbase_dn, L), 0, 16); hdn_for = L; };
    // own column r out of a region of pieces: the eight values in level order
    auto read_col = [&](const float* region, int pc, float (&out)[C]) {
        const wave_u4 v0 = *(const wave_u4*)(region + pc * 8), v1 = *(const wave_u4*)(region + pc * 8 + 4);
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int mm = rf ? 7 - q : q;
            out[q] = __uint_as_float(mm < 4 ? v0[mm] : v1[mm - 4]);
        }
    };

    float dec = 0.f;
    unsigned long long nevals = 0;
    int pending = 0;
    int smp = 0;   // lanes 0 / 1: progress words of the J / K upwind unit as sampled during the previous chunk
    const int* smp_ptr = lane == 0 ? up_j : (lane == 1 ? up_k : nullptr);
    const int perm_up = ((lane - 16) & 63) * 4, perm_dn = ((lane + 16) & 63) * 4;
    const int pc_own0 = (g * PKR) * 16 + tj;   // piece of the lane's column r: pc_own0 + 16 r

    // prologue: the old values at the unit's first level (the window before the first chunk)
    {
#pragma unroll
        for (int it = 0; it < NIT; ++it) wn[it] = __builtin_amdgcn_raw_buffer_load_b128(rsT, coff(cbase[it], Lc - C + 1), 0, 16);
#pragma unroll
        for (int it = 0; it < NIT; ++it) *(wave_u4*)(lds_t + it * 256 + lds_coop) = wn[it];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PKR; ++r) {
            float tmp[C];
            read_col(lds_t, pc_own0 + 16 * r, tmp);
            fnx[r] = (unsigned)(Lc - csum[r]) < (unsigned)NF ? tmp[C - 1] : INF;
        }
        __syncthreads();
    }
    issue_static(Lc);
    issue_halo_dn(Lc);
    for (; Lc <= Le; Lc += C) {
        const int L0 = Lc;
        // (1) both upwind units have published every level <= L0 + C - 2
        const int need = L0 + C - 1;
        {
            bool ok = !smp_ptr || dec_prog(smp) >= need;
            if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {
                unsigned long long t0 = 0;
                int spins = 0;
                for (;;) {
                    if (!ok) { smp = ld_raw(smp_ptr); ok = dec_prog(smp) >= need; }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if (spins == 0) t0 = wall_clock64();
                    if ((++spins & 63) == 0) {
                        if (__builtin_amdgcn_readfirstlane(ld_raw(a.sync + 4))) break;
                        if (wall_clock64() - t0 > a.timeout_ticks) {
                            if (lane == 0) __hip_atomic_store(a.sync + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        // (2) upwind halo pieces of this chunk, unless they were fetched a chunk ahead
        if (hup_for != L0) issue_halo_up(L0);
        if (hdn_for != L0) issue_halo_dn(L0);
        // (3) everything in flight lands, the write-back of the previous chunk drains: its progress goes out
        if (!(FSM_WAVE_EXP & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (pending && lane == 0) st_prog(my_prog, pending);
        pending = 0;
        // (4) the chunk's pieces through LDS to the lanes that march the columns
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            *(wave_u4*)(lds_t + it * 256 + lds_coop) = wn[it];
            *(wave_u4*)(lds_s + it * 256 + lds_coop) = sn[it];
        }
        *(wave_u4*)(&lds_h[0][0] + lds_coop) = hun;
        *(wave_u4*)(&lds_h[1][0] + lds_coop) = hdn;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PKR; ++r) {
            float tin[C];
            read_col(lds_t, pc_own0 + 16 * r, tin);   // levels L0+1 .. L0+8
            read_col(lds_s, pc_own0 + 16 * r, sv[r]);
            read_col(&lds_h[tj == PJ - 1 ? 1 : 0][0], g * PKR + r, hJ[r]);
            w[r][0] = fnx[r];
#pragma unroll
            for (int q = 1; q < C; ++q) w[r][q] = tin[q - 1];
            fnx[r] = tin[C - 1];
        }
        read_col(&lds_h[g == 3 ? 1 : 0][0], PK + tj, hK);
        // nodes outside the grid and frozen nodes: traveltime as it is (outside: +inf), slowness +inf -- never accepted
        bool inside = true;
#pragma unroll
        for (int r = 0; r < PKR; ++r) inside = inside && (unsigned)(L0 - csum[r]) < (unsigned)(NF - C);   // levels L0 .. L0+C in the column
        const bool near = L0 >= near_lo && L0 <= near_hi;
        const bool plain = __builtin_amdgcn_ballot_w64(!inside) == 0ull && !near;
        if (!plain) {
#pragma unroll
            for (int r = 0; r < PKR; ++r) {
#pragma unroll
                for (int q = 0; q < C; ++q) {
                    const int ip = L0 + q - csum[r];
                    const bool valid = (unsigned)ip < (unsigned)NF;
                    bool frz = false;
                    if (near && valid) {
                        const uint32_t n = (uint32_t)(ebase[r] + sf * (L0 + q) - FSM_WAVE_GUARD);
                        frz = (Fz[n >> 5] >> (n & 31)) & 1u;
                    }
                    if (q > 0) w[r][q] = valid ? w[r][q] : INF;   // (q = 0: carried over, settled when it was loaded)
                    sv[r][q] = (valid && !frz) ? sv[r][q] : INF;
                }
                fnx[r] = (unsigned)(L0 + C - csum[r]) < (unsigned)NF ? fnx[r] : INF;
            }
        }
        if (edge_patch) {   // halo columns beyond the grid
#pragma unroll
            for (int r = 0; r < PKR; ++r)
#pragma unroll
                for (int q = 0; q < C; ++q) hJ[r][q] = jh_inf ? INF : hJ[r][q];
#pragma unroll
            for (int q = 0; q < C; ++q) hK[q] = kh_inf ? INF : hK[q];
        }
#pragma unroll
        for (int r = 0; r < PKR; ++r) {
            int lo = L0 - csum[r], hi = lo + C;
            lo = lo < 0 ? 0 : lo;
            hi = hi > NF ? NF : hi;
            nevals += hi > lo ? (unsigned)(hi - lo) : 0u;
        }
        // (5) next chunk: static part now; its upwind halo too when the upwind units are known to be far enough
        const bool more = Lc + C <= Le;
        if (more) {
            issue_static(L0 + C);
            issue_halo_dn(L0 + C);
            const bool cov = !smp_ptr || dec_prog(smp) >= L0 + 2 * C - 1;
            if (__builtin_amdgcn_ballot_w64(!cov) == 0ull) issue_halo_up(L0 + C);
            if (smp_ptr) smp = ld_raw(smp_ptr);   // (decoded at the top of the next chunk)
        }
        // (6) level march
        unsigned long long chg = 0ull;
        // exchange across the lane rows: the K-upwind value of column 0 comes from column PKR-1 of the row below (level q-1,
        // final), the K-downwind value of column PKR-1 from column 0 of the row above (level q+1, old)
        float kup = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(perm_up, (int)__float_as_uint(prev[PKR - 1])));
        float kdn = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(perm_dn, (int)__float_as_uint(w[0][1])));
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const float kup_q = g == 0 ? hK[q] : kup;
            const float kdn_q = g == 3 ? hK[q] : kdn;
            if (q + 1 < C)   // K-downwind value of the next level: old value, fetched a level ahead
                kdn = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(perm_dn, (int)__float_as_uint(q + 2 < C ? w[0][q + 2] : fnx[0])));
            float nak[PKR], naj[PKR], naf[PKR], ns[PKR], nt[PKR];
#pragma unroll
            for (int r = 0; r < PKR; ++r) {
                const float fprev = q == 0 ? prev[r] : w[r][q - 1];
                const float fnext = q == C - 1 ? fnx[r] : w[r][q + 1];
                naf[r] = vmin(fprev, fnext);
                naj[r] = vmin(row_below(fprev, hJ[r][q]), row_above(fnext, hJ[r][q]));
                const float ku = r > 0 ? (q == 0 ? prev[r - 1] : w[r - 1][q - 1]) : kup_q;
                const float kd = r < PKR - 1 ? (q == C - 1 ? fnx[r + 1] : w[r + 1][q + 1]) : kdn_q;
                nak[r] = vmin(ku, kd);
                ns[r] = sv[r][q];
            }
            update3_multi<PKR>(nak, naj, naf, ns, a.dx, nt);
#pragma unroll
            for (int r = 0; r < PKR; ++r) {
                const float c = w[r][q];
                const unsigned long long accm = lanes_gt(c, nt[r]);
                const bool acc = c > nt[r];
                w[r][q] = acc ? nt[r] : c;
                dec += acc ? c - nt[r] : 0.f;
                chg |= accm;
            }
            if (q + 1 < C)   // K-upwind value of the next level for the row above: this level's result
                kup = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(perm_up, (int)__float_as_uint(w[PKR - 1][q])));
        }
#pragma unroll
        for (int r = 0; r < PKR; ++r) prev[r] = w[r][C - 1];
        // (7) write back
        if (!(FSM_WAVE_EXP & 16) && chg != 0ull) {
            if (plain) {
                __syncthreads();   // (every read of the slowness pieces is over)
#pragma unroll
                for (int r = 0; r < PKR; ++r) {
                    wave_u4 v0, v1;
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) {
                        v0[mm] = __float_as_uint(w[r][rf ? 7 - mm : mm]);
                        v1[mm] = __float_as_uint(w[r][rf ? 3 - mm : 4 + mm]);
                    }
                    *(wave_u4*)(lds_s + (pc_own0 + 16 * r) * 8) = v0;
                    *(wave_u4*)(lds_s + (pc_own0 + 16 * r) * 8 + 4) = v1;
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    __builtin_amdgcn_raw_buffer_store_b128(*(const wave_u4*)(lds_s + it * 256 + lds_coop), rsT, coff(cbase[it], L0), 0, 16);
            } else {
#pragma unroll
                for (int r = 0; r < PKR; ++r)
#pragma unroll
                    for (int q = 0; q < C; ++q) {
                        const int ip = L0 + q - csum[r];
                        const uint32_t o = (unsigned)ip < (unsigned)NF ? (uint32_t)(ebase[r] + sf * (L0 + q)) * 4u : 0xfffffff0u;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(w[r][q]), rsT, o, 0, 16);
                    }
            }
        }
        __syncthreads();   // (LDS regions free for the next chunk)
        pending = Lc + C > Le ? FSM_WAVE_DONE : Lc + C;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) st_prog(my_prog, FSM_WAVE_DONE);

    // L1 decrease and evaluated node updates of the unit
    double accd = (double)dec;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        accd += __shfl_down(accd, off, 64);
        nevals += __shfl_down(nevals, off, 64);
    }
    if (lane == 0) {
        if (accd != 0.0) atomicAdd(a.change + slot, accd);
        if (nevals) atomicAdd(a.evals + slot, nevals);
    }
}

// One work unit.  Returns false when the tickets of the launch have run out (or a unit timed out).
template <int PKR, int C>
__device__ __forceinline__ bool fsm_wave_unit(const WaveArgs& a) {
    static_assert(C % 4 == 0 && C >= 4 && PKR >= 1, "chunk shape");
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));   // (nothing derived from it is to be kept across units)
    const unsigned epoch = (unsigned)a.iter_ptr[1];
    const int e2 = (int)(epoch % 3u) + 1;
    auto ld_raw = [&](const int* p_) -> int { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st_prog = [&](int* p_, int v_) {
        __hip_atomic_store(p_, (int)(((unsigned)e2 << 30) | (unsigned)v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // ---- ticket
    int ticket = 0;
    if (lane == 0) {
        ticket = atomicAdd(a.sync + (epoch & 3u), 1);
        if (ticket == 0) __hip_atomic_store(a.sync + ((epoch + 2u) & 3u), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (__builtin_amdgcn_readfirstlane(ld_raw(a.sync + 4))) return false;   // a unit timed out: the solve fails on the host
    const int oidx = ticket / a.batch, z = ticket - oidx * a.batch;
    if (oidx >= a.n_patches * 8) return false;
    const uint32_t unit = a.order[oidx];
    const int dir = (int)(unit >> 28), TJ = (int)(unit & 0x3fffu), TK = (int)((unit >> 14) & 0x3fffu);
    const int slot = a.slots[z];
    if (slot < 0) {   // converged source: nothing to do, but never leave a waiter hanging
        if (lane == 0) st_prog(a.sync + 8 + ((size_t)dir * a.batch + z) * a.n_patches + (TK * a.npj + TJ), FSM_WAVE_DONE);
        return true;
    }
    __shared__ __attribute__((aligned(16))) float lds[2 * 64 * PKR * 8 + 2 * 32 * 8];
    if (dir & 1) fsm_wave_body<PKR, C, true>(a, lds, lane, e2, dir, TJ, TK, z, slot);
    else fsm_wave_body<PKR, C, false>(a, lds, lane, e2, dir, TJ, TK, z, slot);
    return true;
}

template <int PKR, int C>
__global__ __launch_bounds__(64) void fsm_sweep_wave(const WaveArgs a) {
    (void)a;
    for (;;) {
        auto kp = __builtin_amdgcn_kernarg_segment_ptr();   // (arguments re-read per unit: nothing is carried, fsm_sweep_persistent)
        asm volatile("" : "+s"(kp));
        if (!fsm_wave_unit<PKR, C>(*(const WaveArgs*)kp)) break;
    }
}

}  // namespace ttcr_amd
