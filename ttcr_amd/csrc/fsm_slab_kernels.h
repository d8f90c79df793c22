// ttcr_amd/csrc/fsm_slab_kernels.h -- first-order 3-D sweeps of fp32 grids (one field per slot) by SLABS: a work unit is a patch of
// 64 (J) x NW PKR (K) columns swept by a workgroup of NW wavefronts, every wavefront marching its own slab of 64 x PKR columns.
//
// What it computes: Grid3Drn::sweep + update_node (ttcr/Grid3Drn.h:2816-2959) -- the same partial order and arithmetic as
// fsm_sweep_persistent (fsm_kernels.h): level L = i' + j' + k' of the oriented indices; any linear extension of the sweep's
// partial order gives the serial Gauss-Seidel result bit for bit.
//
// Why (profiles/r03/lone_source_floor.txt): in the 16 x 16 four-wave patches a level is one chain of
// LDS write -> barrier -> LDS read -> ~45 dependent vector instructions (7-8 cycles each for a lone wave), the four waves wait for
// the slowest at every level, a third of all instructions stage tiles through LDS, and a 32-byte piece stored write-through drops
// its 128-byte line from the L2 before the next piece of the line is fetched.  Here:
//   * lane = j' (all 64 lanes along J): the J neighbours of a node are the adjacent lanes -- two DPP wave shifts, no LDS, no barrier;
//   * a lane owns PKR columns (consecutive k'): the K neighbours inside a slab are the lane's own registers, and the PKR node
//     updates of a level are independent chains that interleave in the one instruction stream;
//   * the slabs of a workgroup are coupled through LDS only: a wavefront reads the last row of the slab above out of that slab's
//     ring once its level counter (an LDS word) has passed -- waves run skewed by a level or two, nobody waits at a barrier;
//   * traveltimes live in an LDS RING of 32 levels per column (one 128-byte line): aligned 32-byte pieces are fetched a chunk ahead
//     by lane pairs (16-byte buffer loads), marched in place, written back as aligned pieces -- every such access is a whole
//     32-byte sector; the march reads / writes one ring entry per node (conflict free: the skew spreads the lanes over the banks),
//     and the old values it needs come in a level ahead (they are off the dependent chain);
//   * slowness comes from the sheared copies (fsm_kernels.h): at one level the 64 lanes of a row read consecutive elements, and
//     the level part of the address is the same for every lane -- a buffer load with a scalar offset, no address arithmetic;
//   * workgroup -> workgroup hand-off as in fsm_sweep_persistent (progress words with launch epochs, sc1 stores / loads), one
//     progress word per slab: the patch below (K) reads the last slab's word, the patch beside (J) the word of the slab with its
//     rows.  The columns a neighbour patch reads are stored level-aligned once more per chunk so that it can follow a chunk behind.
// Scheduling is that of the whole-iteration launch: units (direction, patch) drawn from a ticket counter in an order
// in which a unit comes after everything it waits for; a unit starts once the previous sweep has finished the <= 3 x 3 patches
// around it.  Needs NF % 8 == 0 (pieces are aligned in both sweep directions); other grids keep fsm_sweep_persistent.
#pragma once
#include "fsm_kernels.h"
#include "fsm_slab_api.h"
#ifndef FSM_SLAB_EXP
#define FSM_SLAB_EXP 0   // TIMING experiments (wrong results): 1: no unit waits for another unit; 2: no wavefront waits for another one of its
                         // workgroup; 4: no traveltime loads; 8: no traveltime stores; 16: no slowness loads; 32: publish without draining
#endif

#ifndef FSM_SLAB_PUB
#define FSM_SLAB_PUB 1   // when a chunk's progress goes out: 1 right behind its write-back (the wavefront sits out the drain of its stores), 0 in the
                         // middle of the next chunk's march (the drain is hidden, the patches downstream hear of it half a chunk later)
#endif
#ifndef FSM_SLAB_PROF
#define FSM_SLAB_PROF 0   // 1: phase timers (thread 0 of every wavefront; TTCR_FSM_PROF=1 prints them)
#endif

namespace ttcr_amd {

typedef unsigned int slab_u4 __attribute__((ext_vector_type(4)));
// LDS pointers with their address space spelled out: the accesses below are ds_read / ds_write whatever the optimiser infers
typedef __attribute__((address_space(3))) char slab_lds_char;
typedef __attribute__((address_space(3))) float slab_lds_float;
typedef __attribute__((address_space(3))) slab_u4 slab_lds_u4;
typedef __attribute__((address_space(3))) volatile int slab_lds_flag;

// The local solver of update3 (fsm_kernels.h; Grid3Drn::update_node, ttcr/Grid3Drn.h:2936-2956) for N nodes that do not depend on
// each other (the nodes a lane updates at one level).  Same formulas, same roundings, same two wave-uniform short cuts -- taken for
// the N nodes TOGETHER, so that between the branches the N fp64 discriminant chains are one straight piece of code the compiler
// interleaves.  Which branch is taken does not change a value: the 2-D root is exact wherever it is computed, and where it is
// not, every lane beyond the 1-D branch is provably 3-D (derivation at update3).
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

// One work unit: direction `dir` of patch (TJ, TK) for batch entry z (slot `slot`), the part of wavefront w.
// s_wlev[NW]: last level every wavefront of the workgroup has finished in this unit (LDS; written by lane 0 of its wavefront).
template <int PKR, int NW>
__device__ __forceinline__ void fsm_slab_body(const SlabArgs& a, slab_lds_char* lds, slab_lds_flag* s_wlev, int lane, int w, int e2, int dir, int TJ, int TK,
                                              int z, int slot, int Lc) {
    constexpr int C = FSM_SLAB_C, PJ = 64, KW = NW * PKR;
    constexpr int NIT = 2 * (PKR + 1);    // aligned-piece accesses per chunk: 32 pieces each, rows 0 .. PKR (PKR: the row below the slab)
    constexpr int NITS = 2 * PKR;         // ... of them own rows (stored)
    constexpr int NHK = 2;                // accesses of a level-aligned row of 64 runs (K-upwind halo in, K-edge row out)
    const float INF = __builtin_huge_valf();
    auto dec_prog = [&](int raw_) -> int { return (int)((unsigned)raw_ >> 30) == e2 ? (raw_ & 0x3fffffff) : 0; };
    auto ld_raw = [&](const int* p_) -> int { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st_prog = [&](int* p_, int v_) {
        __hip_atomic_store(p_, (int)(((unsigned)e2 << 30) | (unsigned)v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const int NF = a.g.NF, NJ = a.g.NJ, NK = a.g.NK, npj = a.npj;
    int* prog = a.sync + 8 + ((size_t)dir * a.batch + z) * a.n_patches * (NW + 1);
    int* my_prog = prog + (size_t)(TK * npj + TJ) * (NW + 1) + w;
    const int rf = dir & 1, rj = (dir >> 1) & 1, rk = (dir >> 2) & 1;   // ttcr/Grid3Drn.h:2816-2899
    const int rev = rk, fam = (rf ^ rk) | ((rj ^ rk) << 1);             // sheared slowness copy and its traversal (fsm_kernels.h)
    const int* up_j = (!(FSM_SLAB_EXP & 1) && TJ > 0) ? prog + (size_t)(TK * npj + TJ - 1) * (NW + 1) + w : nullptr;
    const int* up_k = (!(FSM_SLAB_EXP & 1) && TK > 0 && w == 0) ? prog + (size_t)((TK - 1) * npj + TJ) * (NW + 1) + (NW - 1) : nullptr;

    const int j0 = TJ * PJ, k0 = TK * KW, kw0 = k0 + w * PKR;
    const int jmaxp = (j0 + PJ < NJ ? j0 + PJ : NJ) - 1, kmaxp = (k0 + KW < NK ? k0 + KW : NK) - 1;
    const int Le = jmaxp + kmaxp + NF - 1;
    // (Lc: first chunk start, congruent to TJ + TK modulo C (fsm_kernels.h): an upwind patch's chunks end one level before ours)
    const int jp = j0 + lane;
    const bool jup_ex = j0 > 0, jdn_ex = j0 + PJ < NJ, kup_ex = k0 > 0, kdn_ex = k0 + KW < NK;
    const int step4 = rf ? -4 : 4;            // ring entries are indexed by the NATURAL index along F (pieces are aligned in memory)
    auto nat_rowb = [&](int jq, int kq) -> int {   // element index of natural i = 0 of the column at oriented (jq, kq)
        return (int)(((uint32_t)(rk ? NK - 1 - kq : kq) * NJ + (rj ? NJ - 1 - jq : jq)) * NF);
    };
    // natural index of oriented i'
    auto nat_i = [&](int ip) -> int { return rf ? NF - 1 - ip : ip; };

    // ---- ring: rows of 66 columns of 32 entries.  Row 0: the K-upwind halo row (columns k0 - 1); then per wavefront its PKR rows and
    // the row below them (old values only: the next slab / patch updates its own copy).  Column 0 / 65: the J halos of a row.
    auto row_base = [&](int ww, int r) -> int { return (1 + ww * (PKR + 1) + r) * FSM_SLAB_ROWB; };
    const int my_col = (lane + 1) * FSM_SLAB_COLB;
    int ring_own[PKR];   // byte address of entry 0 of the lane's column in row r
#pragma unroll
    for (int r = 0; r < PKR; ++r) ring_own[r] = row_base(w, r) + my_col;
    const int ring_below = row_base(w, PKR) + my_col;
    const int ring_above = (w == 0 ? 0 : row_base(w - 1, PKR - 1)) + my_col;
    const int hoff = lane < 32 ? -FSM_SLAB_COLB : FSM_SLAB_COLB;   // lane 0 reads the J-upwind halo column, lane 63 the J-downwind one
    auto ldsf = [&](int byte_addr) -> slab_lds_float& { return *(slab_lds_float*)(lds + byte_addr); };
    auto lds4 = [&](int byte_addr) -> slab_lds_u4& { return *(slab_lds_u4*)(lds + byte_addr); };
    const int dummy = fsm_slab_rows(PKR, NW) * FSM_SLAB_ROWB + lane * 16;   // where the pieces of no column go

    // columns of the lane.  csum = j' + k' (outside the grid: far negative, so that i' = L - csum is never in range)
    int csum[PKR];
    uint32_t frz_base[PKR];   // natural node index of the column's i = 0 (frozen bits)
    int soff[PKR];            // lane part of the position in the sheared slowness copy (bytes)
#pragma unroll
    for (int r = 0; r < PKR; ++r) {
        const int kp = kw0 + r;
        const bool ok = jp < NJ && kp < NK;
        csum[r] = ok ? jp + kp : -(1 << 29);
        frz_base[r] = ok ? (uint32_t)nat_rowb(jp, kp) : 0u;
        const int kx = rev ? NK - 1 - kp : kp, jx = rev ? NJ - 1 - jp : jp;
        soff[r] = ok ? (int)((((size_t)kx * (size_t)(a.g.M >> 1)) * (size_t)a.g.SR + (size_t)((jx >> 4) * 32 + (jx & 15))) * 4) : 0;
    }
    // buffer descriptors (hardware bounds check: an access outside reads 0 / is dropped -- how pieces of no column are masked)
    const uint32_t nbytes = (a.g.n_nodes + 2u * FSM_SLAB_GUARD) * 4u;
    __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(a.tt + (size_t)slot * a.g.n_nodes - FSM_SLAB_GUARD, 0, nbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ssh) + (size_t)fam * a.ssh_stride, 0, (uint32_t)(a.ssh_stride * 4u), 0x00020000);
    constexpr uint32_t OOB = 0xfffffff0u;
    const uint32_t* __restrict__ Fz = a.frozen + (size_t)slot * a.mask_words;

    // chunks that may hold frozen nodes (fsm_kernels.h: near_lo / near_hi), for the whole patch
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

    // bounded waits on an LDS level counter of another wavefront of the workgroup
    auto lds_wait = [&](slab_lds_flag* p_, int want) -> int {
        if (FSM_SLAB_EXP & 2) return want;
        int v = __builtin_amdgcn_readfirstlane(*p_);
        if (v >= want) return v;
        unsigned long long t0 = wall_clock64();
        int spins = 0;
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
            v = __builtin_amdgcn_readfirstlane(*p_);
            if (v >= want) return v;
            if ((++spins & 255) == 0) {
                if (__builtin_amdgcn_readfirstlane(ld_raw(a.sync + 4))) return want;
                if (wall_clock64() - t0 > a.timeout_ticks) {
                    if (lane == 0) __hip_atomic_store(a.sync + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return want;
                }
            }
        }
    };

    // ---- ring entries nobody loads: halo columns / rows beyond the grid read as +inf (the reference's one-sided neighbour pick)
    {
        if (lane < 32) {
#pragma unroll
            for (int r = 0; r < PKR; ++r) {
                if (!jup_ex) ldsf(row_base(w, r) + 0 * FSM_SLAB_COLB + lane * 4) = INF;
                if (!jdn_ex) ldsf(row_base(w, r) + 65 * FSM_SLAB_COLB + lane * 4) = INF;
            }
        }
        const slab_u4 inf4 = {0x7f800000u, 0x7f800000u, 0x7f800000u, 0x7f800000u};
        if (kw0 + PKR >= NK) {
#pragma unroll
            for (int q = 0; q < 8; ++q) lds4(ring_below + 16 * q) = inf4;
        }
        if (w == 0 && !kup_ex) {
#pragma unroll
            for (int q = 0; q < 8; ++q) lds4(my_col + 16 * q) = inf4;
        }
    }

    // ---- aligned pieces: access `it`, lane pair p = lane >> 1 <-> piece slot it * 32 + p = (row rr, column cj); lane & 1: which half
    const int half = lane & 1;
    int pc_csum[NIT], pc_rowb[NIT], pc_lds[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int sp = it * 32 + (lane >> 1);
        const int rr = sp >> 6, cj = sp & 63;
        const int jq = j0 + cj, kq = kw0 + rr;
        const bool ok = jq < NJ && kq < NK;
        pc_csum[it] = ok ? jq + kq : -(1 << 29);
        pc_rowb[it] = ok ? (FSM_SLAB_GUARD + nat_rowb(jq, kq)) * 4 + 16 * half : 0;
        pc_lds[it] = row_base(w, rr) + (cj + 1) * FSM_SLAB_COLB + 16 * half;
    }
    // byte offset inside a row of the piece with oriented index po (oriented elements 8 po .. 8 po + 7; NF % 8 == 0)
    auto piece_nat4 = [&](int po) -> int { return (rf ? NF - 8 - 8 * po : 8 * po) * 4; };
    slab_u4 ldv[NIT];
    if (FSM_SLAB_EXP & 4) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) ldv[it] = slab_u4{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
    }
    int ld_dst[NIT];   // ring address the piece in ldv goes to (the dummy area: no piece)
    // pieces needed from chunk X + C on: the one that holds oriented i' = X + 2C + 1 - csum (everything below is there already; the
    // last level of chunk X + C fetches the old value of the node two levels on, i' = X + 2C + 1 - csum)
    auto issue_pieces = [&](int X) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int t = X + 2 * C + 1 - pc_csum[it];
            const bool ok = (unsigned)t < (unsigned)NF;
            const int n4 = piece_nat4(t >> 3);
            if (!(FSM_SLAB_EXP & 4)) ldv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsT, ok ? (uint32_t)(pc_rowb[it] + n4) : OOB, 0, 16);   // sc1: another XCD may have written it in this launch
            ld_dst[it] = ok ? pc_lds[it] + (n4 & 127) : dummy;
        }
    };
    auto land_pieces = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) lds4(ld_dst[it]) = ldv[it];
    };
    // write-back of the pieces completed by chunk X (levels X .. X + 7 done): last element in [u - 7, u], u = X + 7 - csum
    auto store_pieces = [&](int X) {
#pragma unroll
        for (int it = 0; it < NITS; ++it) {
            const int u = X + C - 1 - pc_csum[it];
            const bool ok = (unsigned)(u - 7) < (unsigned)NF;
            const int n4 = piece_nat4(((u + 1) >> 3) - 1);
            const slab_u4 v = lds4(pc_lds[it] + (ok ? (n4 & 127) : 0));
            if (!(FSM_SLAB_EXP & 8)) __builtin_amdgcn_raw_buffer_store_b128(v, rsT, ok ? (uint32_t)(pc_rowb[it] + n4) : OOB, 0, 16);
        }
    };

    // ---- level-aligned runs (halos in, edge columns out): 8 consecutive levels of a column = 8 consecutive elements at any alignment;
    // a lane moves 4 of them.  ns = natural index of the lane's first element of the run whose first ORIENTED element is ia
    auto run_ns = [&](int ia) -> int { return (rf ? NF - 8 - ia : ia) + 4 * half; };
    auto ring_put4 = [&](int ring_col, int ns, slab_u4 v) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ldsf(ring_col + (int)(((unsigned)(ns + q) << 2) & 124u)) = __uint_as_float(v[q]);
    };
    auto ring_get4 = [&](int ring_col, int ns) -> slab_u4 {
        slab_u4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __float_as_uint(ldsf(ring_col + (int)(((unsigned)(ns + q) << 2) & 124u)));
        return v;
    };
    // load of a run: the 16 bytes of the lane, whatever part of them lies in the column (the rest belongs to a neighbouring row and
    // only ever reaches nodes that do not exist); nothing of it in the column: no access
    auto run_load = [&](int rowb4, int ns, bool exists) -> slab_u4 {
        const bool ok = exists && ns + 3 >= 0 && ns < NF;
        if (FSM_SLAB_EXP & 4) return slab_u4{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
        return __builtin_amdgcn_raw_buffer_load_b128(rsT, ok ? (uint32_t)(rowb4 + ns * 4) : OOB, 0, 16);
    };
    // store of a run: element by element, only nodes of the column (its neighbours in memory are other columns' nodes)
    auto run_store = [&](int rowb4, int ns, bool exists, slab_u4 v) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = exists && (unsigned)(ns + q) < (unsigned)NF;
            if (!(FSM_SLAB_EXP & 8)) __builtin_amdgcn_raw_buffer_store_b32(v[q], rsT, ok ? (uint32_t)(rowb4 + (ns + q) * 4) : OOB, 0, 16);
        }
    };
    // J halos in / J edge out.  Pair p < PKR: row p, upwind halo (column j0 - 1, levels X-1 .. X+6: the i' of column j0 at X .. X+7);
    // PKR <= p < 2 PKR: row p - PKR, downwind halo (column j0 + 64, levels X+1 .. X+8: the i' of column j0 + 63 at X .. X+7).
    // The J edge a neighbour patch reads is column j0 + 63 of every row, levels X .. X+7 (pair p < PKR).
    bool hj_ex = false, ej_ex = false;
    int hj_rowb = 0, hj_ring = 0, hj_csum = 0, ej_rowb = 0, ej_ring = 0, ej_csum = 0;
    {
        const int p = lane >> 1;
        if (p < 2 * PKR) {
            const bool up = p < PKR;
            const int r = up ? p : p - PKR, kq = kw0 + r, jq = up ? j0 - 1 : j0 + PJ;
            if ((up ? jup_ex : jdn_ex) && kq < NK) {
                hj_ex = true;
                hj_rowb = (FSM_SLAB_GUARD + nat_rowb(jq, kq)) * 4;
                hj_ring = row_base(w, r) + (up ? 0 : 65) * FSM_SLAB_COLB;
                hj_csum = (up ? j0 : j0 + PJ - 1) + kq;   // the own column beside it: same i' at the levels concerned
            }
            if (up && jdn_ex && kq < NK) {
                ej_ex = true;
                ej_rowb = (FSM_SLAB_GUARD + nat_rowb(j0 + PJ - 1, kq)) * 4;
                ej_ring = row_base(w, r) + 64 * FSM_SLAB_COLB;
                ej_csum = j0 + PJ - 1 + kq;
            }
        }
    }
    // K-upwind halo row in (wavefront 0: row k0 - 1, levels X-1 .. X+6 = the i' of row k0 at X .. X+7) and K edge row out (last
    // wavefront: row k0 + KW - 1, levels X .. X+7): access hk, pair p <-> column j0 + hk * 32 + p
    bool hk_ex[NHK], ek_ex[NHK];
    int hk_rowb[NHK], hk_ring[NHK], hk_csum[NHK], ek_rowb[NHK], ek_ring[NHK], ek_csum[NHK];
#pragma unroll
    for (int hk = 0; hk < NHK; ++hk) {
        const int cj = hk * 32 + (lane >> 1), jq = j0 + cj, kq = k0 + KW - 1;
        hk_ex[hk] = w == 0 && kup_ex && jq < NJ;
        hk_rowb[hk] = hk_ex[hk] ? (FSM_SLAB_GUARD + nat_rowb(jq, k0 - 1)) * 4 : 0;
        hk_ring[hk] = (cj + 1) * FSM_SLAB_COLB;   // row 0
        hk_csum[hk] = jq + k0;
        ek_ex[hk] = w == NW - 1 && kdn_ex && jq < NJ;
        ek_rowb[hk] = ek_ex[hk] ? (FSM_SLAB_GUARD + nat_rowb(jq, kq)) * 4 : 0;
        ek_ring[hk] = row_base(w, PKR - 1) + (cj + 1) * FSM_SLAB_COLB;
        ek_csum[hk] = jq + kq;
    }
    slab_u4 hjv = {0u, 0u, 0u, 0u}, hkv[NHK];
#pragma unroll
    for (int hk = 0; hk < NHK; ++hk) hkv[hk] = hjv;
    int halo_for = -(1 << 30);   // chunk whose halo runs are in hjv / hkv
    auto issue_halo = [&](int X) {
        hjv = run_load(hj_rowb, run_ns(X - hj_csum), hj_ex);
        if (w == 0 && kup_ex) {
#pragma unroll
            for (int hk = 0; hk < NHK; ++hk) hkv[hk] = run_load(hk_rowb[hk], run_ns(X - hk_csum[hk]), hk_ex[hk]);
        }
        halo_for = X;
    };
    auto land_halo = [&](int X) {
        if (hj_ex) ring_put4(hj_ring, run_ns(X - hj_csum), hjv);
        if (w == 0 && kup_ex) {
#pragma unroll
            for (int hk = 0; hk < NHK; ++hk)
                if (hk_ex[hk]) ring_put4(hk_ring[hk], run_ns(X - hk_csum[hk]), hkv[hk]);
        }
    };
    auto store_edges = [&](int X) {
        if (jdn_ex) {
            const int ns = run_ns(X - ej_csum);
            run_store(ej_rowb, ns, ej_ex, ring_get4(ej_ring, ns));
        }
        if (w == NW - 1 && kdn_ex) {
#pragma unroll
            for (int hk = 0; hk < NHK; ++hk) {
                const int ns = run_ns(X - ek_csum[hk]);
                run_store(ek_rowb[hk], ns, ek_ex[hk], ring_get4(ek_ring[hk], ns));
            }
        }
    };

    // ---- slowness of a chunk: sheared copy, position (k', x = (i' + j') mod M, j'); x is the same for all lanes of a row at a level
    const int M = a.g.M, SR = a.g.SR;
    auto shear_soff = [&](int L, int r) -> uint32_t {   // scalar part (bytes) of the position of row r at level L
        int x = L - (kw0 + r);
        x = rev ? NF + NJ - 2 - x : x;
        x = x < 0 ? x + M : x;
        x = x >= M ? x - M : x;
        x = x < 0 ? 0 : (x >= M ? M - 1 : x);   // (levels at which the row has no node)
        return (uint32_t)(((x >> 1) * SR + (x & 1) * 16) * 4);
    };
    float sv[PKR][C], svn[PKR][C];
    auto issue_slowness = [&](int X) {
#pragma unroll
        for (int r = 0; r < PKR; ++r)
#pragma unroll
            for (int q = 0; q < C; ++q)
                svn[r][q] = (FSM_SLAB_EXP & 16) ? 0.05f : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsS, (uint32_t)soff[r], shear_soff(X + q, r), 0));
    };

    float dec = 0.f;
    unsigned long long nevals = 0;
    unsigned long long pacc[6] = {0, 0, 0, 0, 0, 0}, prof_t = (FSM_SLAB_PROF && a.prof) ? wall_clock64() : 0ull;
    unsigned pchunks = 0;
#define SLAB_MARK(slot_)                                                  \
    if (FSM_SLAB_PROF && a.prof) {                                        \
        const unsigned long long now_ = wall_clock64();                   \
        pacc[slot_] += now_ - prof_t;                                     \
        prof_t = now_;                                                    \
    }
    int smp = 0;   // lanes 0 / 1: progress words of the J / K upwind slab as sampled during the previous chunk
    const int* smp_ptr = lane == 0 ? up_j : (lane == 1 ? up_k : nullptr);
    auto wait_upwind = [&](int need) {
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
    };

    // ---- prologue: the ring as the top of the first chunk wants it (the pieces that hold oriented i' <= Lc + 9 - csum)
    issue_pieces(Lc - 2 * C);
    land_pieces();
    issue_pieces(Lc - C);
    issue_slowness(Lc);

    // march state: np = result at level L-1 (the F-upwind neighbour), cc = old value of the node at level L, cn = old value of the node
    // at level L+1 (the F-downwind neighbour; its J / K neighbours' cn are the J / K-downwind values), q4 = ring position of the node
    float np[PKR], cc[PKR], cn[PKR];
    int q4[PKR];
#pragma unroll
    for (int r = 0; r < PKR; ++r) {
        np[r] = INF;
        cc[r] = INF;
        cn[r] = INF;
        q4[r] = (int)(((unsigned)nat_i(Lc - csum[r]) << 2) & 124u);
    }
    int updone = Lc - 1;       // last level the slab above is known to have finished (the level counters start there)
    int pending = 0;           // progress value to publish once the stores of the previous chunk have drained
    unsigned long long chg_prev = 0ull;
    bool first = true;

    SLAB_MARK(5)
    for (; Lc <= Le; Lc += C) {
        const int X = Lc;
        // (1) both upwind slabs (other patches) have published every level <= X + C - 2
        wait_upwind(X + C - 1);
        if (halo_for != X) issue_halo(X);
        SLAB_MARK(0)
        // (2) the slab below must be done with the ring entries the new pieces replace (up to level X - 16 of our last row, which it
        // reads during its level X - 15)
        if (w < NW - 1) (void)lds_wait(s_wlev + w + 1, X - 2 * C + 1);
        asm volatile("" ::: "memory");
        // (3) pieces and halos into the ring
        land_pieces();
        land_halo(X);
        SLAB_MARK(1)
#pragma unroll
        for (int r = 0; r < PKR; ++r)
#pragma unroll
            for (int q = 0; q < C; ++q) sv[r][q] = svn[r][q];
        const bool more = X + C <= Le;
        if (more) {
            issue_pieces(X);
            issue_slowness(X + C);
        }
        if (first) {   // old values of the nodes at the first two levels
#pragma unroll
            for (int r = 0; r < PKR; ++r) {
                cc[r] = ldsf(ring_own[r] + q4[r]);
                cn[r] = ldsf(ring_own[r] + ((q4[r] + step4) & 124));
            }
            first = false;
        }
        // nodes outside the grid / frozen nodes in this chunk?
        bool inside = true;
#pragma unroll
        for (int r = 0; r < PKR; ++r) inside = inside && (unsigned)(X - csum[r] - 1) <= (unsigned)(NF - 10);   // i' - 1 .. i' + 8 in the column at every level
        const bool near = X >= near_lo && X <= near_hi;
        const bool plain = __builtin_amdgcn_ballot_w64(!inside) == 0ull && !near;
#pragma unroll
        for (int r = 0; r < PKR; ++r) {
            int lo = X - csum[r], hi = lo + C;
            lo = lo < 0 ? 0 : lo;
            hi = hi > NF ? NF : hi;
            nevals += hi > lo ? (unsigned)(hi - lo) : 0u;
        }
        unsigned long long chg = 0ull;
        // halo values of the first level (the later ones come in a level ahead)
        float hv[PKR], kd;
#pragma unroll
        for (int r = 0; r < PKR; ++r) hv[r] = ldsf(ring_own[r] + hoff + q4[r]);
        kd = ldsf(ring_below + q4[PKR - 1]);
        SLAB_MARK(2)

        auto march = [&](auto masked_tag, auto half_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            constexpr int E0 = decltype(half_tag)::value * (C / 2);
#pragma unroll
            for (int e = E0; e < E0 + C / 2; ++e) {
                const int L = X + e;
                int q4n[PKR];
                float cnn[PKR], hvn[PKR], kdn = 0.f;
#pragma unroll
                for (int r = 0; r < PKR; ++r) q4n[r] = (q4[r] + step4) & 124;
                // old values of the next level, off the chain
#pragma unroll
                for (int r = 0; r < PKR; ++r) {
                    cnn[r] = ldsf(ring_own[r] + ((q4n[r] + step4) & 124));
                    hvn[r] = e < C - 1 ? ldsf(ring_own[r] + hoff + q4n[r]) : 0.f;
                }
                if (e < C - 1) kdn = ldsf(ring_below + q4n[PKR - 1]);
                // the slab above has finished level L - 1: its last row's results are in its ring
                float ku;
                if (w > 0) {
                    if (updone < L - 1) updone = lds_wait(s_wlev + w - 1, L - 1);
                    asm volatile("" ::: "memory");   // (nothing of the slab above is read before its counter)
                    ku = ldsf(ring_above + q4[0]);
                } else {
                    ku = ldsf(ring_above + q4[0]);
                }
                bool valid[PKR];
                float sl[PKR];
#pragma unroll
                for (int r = 0; r < PKR; ++r) {
                    sl[r] = sv[r][e];
                    valid[r] = true;
                    if (MASKED) {
                        const int ip = L - csum[r];
                        valid[r] = (unsigned)ip < (unsigned)NF;
                        const bool validn = (unsigned)(ip + 1) < (unsigned)NF;
                        cn[r] = validn ? cn[r] : INF;
                        bool frz = false;
                        if (near && valid[r]) {
                            const uint32_t n = frz_base[r] + (uint32_t)nat_i(ip);
                            frz = (Fz[n >> 5] >> (n & 31)) & 1u;
                        }
                        sl[r] = (valid[r] && !frz) ? sl[r] : INF;   // never accepted: the update is +inf / NaN
                    }
                }
                float ak[PKR], aj[PKR], af[PKR], nt[PKR];
#pragma unroll
                for (int r = 0; r < PKR; ++r) {
                    af[r] = vmin(np[r], cn[r]);
                    aj[r] = vmin(lane_below(np[r], hv[r]), lane_above(cn[r], hv[r]));
                    const float km = r > 0 ? np[r - 1] : ku;
                    const float kp = r < PKR - 1 ? cn[r + 1] : kd;
                    ak[r] = vmin(km, kp);
                }
                update3_multi<PKR>(ak, aj, af, sl, a.dx, nt);
#pragma unroll
                for (int r = 0; r < PKR; ++r) {
                    const float c = cc[r];
                    const unsigned long long accm = lanes_gt(c, nt[r]);
                    const bool acc = c > nt[r];
                    const float nv = acc ? nt[r] : c;
                    dec += acc ? c - nt[r] : 0.f;
                    chg |= accm;
                    if (!MASKED || valid[r]) ldsf(ring_own[r] + q4[r]) = nv;
                    np[r] = (!MASKED || valid[r]) ? nv : INF;
                    cc[r] = cn[r];
                    cn[r] = cnn[r];
                    hv[r] = hvn[r];
                    q4[r] = q4n[r];
                }
                kd = kdn;
                asm volatile("" ::: "memory");   // (the results above are issued before the level counter moves)
                if (lane == 0) s_wlev[w] = L;
            }
        };
        if (plain) march(std::false_type{}, std::integral_constant<int, 0>{}); else march(std::true_type{}, std::integral_constant<int, 0>{});
        SLAB_MARK(3)
        // the write-back of the chunk before has had half a march to drain: its progress goes out.  Vector memory operations of a
        // wavefront complete in order on this target (loads and stores count down the one vmcnt alike), so waiting until no more are
        // outstanding than were issued AFTER those stores -- this chunk's pieces and slowness -- is waiting for the stores.
        if (pending) {
            if (!(FSM_SLAB_EXP & 32)) {
                if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIT + C * PKR) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (lane == 0) st_prog(my_prog, pending);
            pending = 0;
        }
        if (more) {   // sample the upwind words for the next chunk; its halo runs too when they are there already
            if (smp_ptr) smp = ld_raw(smp_ptr);
        }
        SLAB_MARK(4)
        if (plain) march(std::false_type{}, std::integral_constant<int, 1>{}); else march(std::true_type{}, std::integral_constant<int, 1>{});
        SLAB_MARK(3)
        // (5) write back: the columns the neighbour patches read, level-aligned, and the aligned pieces this chunk completed
        if (chg != 0ull) store_edges(X);
        if ((chg | chg_prev) != 0ull) store_pieces(X);
        chg_prev = chg;
        pending = X + C > Le ? 0 : X + C;
        if (FSM_SLAB_PUB == 1 && pending) {
            if (!(FSM_SLAB_EXP & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this chunk's pieces and slowness came in long ago)
            if (lane == 0) st_prog(my_prog, pending);
            pending = 0;
        }
        if (more) {   // the next chunk's upwind halo runs, when the upwind slabs are known to be far enough
            const bool cov = !smp_ptr || dec_prog(smp) >= X + 2 * C - 1;
            if (__builtin_amdgcn_ballot_w64(!cov) == 0ull) issue_halo(X + C);
        }
        ++pchunks;
        SLAB_MARK(4)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) st_prog(my_prog, FSM_SLAB_DONE);

    // L1 decrease and evaluated node updates of the slab
    double accd = (double)dec;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        accd += __shfl_down(accd, off, 64);
        nevals += __shfl_down(nevals, off, 64);
    }
    if (lane == 0) {
        if (accd != 0.0) atomicAdd(a.change + slot, accd);
        if (nevals) atomicAdd(a.evals + slot, nevals);
        if (FSM_SLAB_PROF && a.prof) {
            SLAB_MARK(5)
            for (int q = 0; q < 6; ++q) atomicAdd(a.prof + q, pacc[q]);
            atomicAdd(a.prof + 6, 1ull);
            atomicAdd(a.prof + 7, (unsigned long long)pchunks);
        }
    }
#undef SLAB_MARK
}

// One work unit.  Returns false when the tickets of the launch have run out (or a unit timed out).
template <int PKR, int NW>
__device__ __forceinline__ bool fsm_slab_unit(const SlabArgs& a, slab_lds_char* lds) {
    constexpr int PJ = 64, KW = NW * PKR;
    __shared__ int s_ticket, s_abort;
    __shared__ int s_wlev[NW];
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // (nothing derived from it is to be kept across units)
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned epoch = (unsigned)a.iter_ptr[1];
    const int e2 = (int)(epoch % 3u) + 1;
    auto ld_raw = [&](const int* p_) -> int { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto dec_prog = [&](int raw_) -> int { return (int)((unsigned)raw_ >> 30) == e2 ? (raw_ & 0x3fffffff) : 0; };
    auto st_prog = [&](int* p_, int v_) {
        __hip_atomic_store(p_, (int)(((unsigned)e2 << 30) | (unsigned)v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    __syncthreads();   // every wavefront is done with the unit before (ring, level counters, ticket word)
    if (tid == 0) {
        const int t_ = atomicAdd(a.sync + (epoch & 3u), 1);
        if (t_ == 0) __hip_atomic_store(a.sync + ((epoch + 2u) & 3u), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ticket = t_;
    }
    if (tid == 1) s_abort = ld_raw(a.sync + 4);
    __syncthreads();
    if (s_abort) return false;   // a unit timed out: the solve fails on the host
    const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);
    const int oidx = ticket / a.batch, z = ticket - oidx * a.batch;
    if (oidx >= a.n_patches * 8) return false;
    const uint32_t unit = a.order[oidx];
    const int dir = (int)(unit >> 28), TJ = (int)(unit & 0x3fffu), TK = (int)((unit >> 14) & 0x3fffu);
    int* fin = a.sync + 8 + (((size_t)dir * a.batch + z) * a.n_patches + (size_t)(TK * a.npj + TJ)) * (NW + 1) + NW;
    const int slot = a.slots[z];
    if (slot < 0) {   // converged source: nothing to do, but never leave a waiter hanging
        if (tid <= NW) st_prog(fin - NW + tid, FSM_SLAB_DONE);
        return true;
    }
    // ---- previous sweep: the patches (of ITS oriented partition) that own a column within 2 of ours have finished
    if (!(FSM_SLAB_EXP & 1) && dir > 0 && tid < 64) {
        const int NJ = a.g.NJ, NK = a.g.NK;
        const int pd = dir - 1;
        const int rj = (dir >> 1) & 1, rk = (dir >> 2) & 1, prj = (pd >> 1) & 1, prk = (pd >> 2) & 1;
        const int j0 = TJ * PJ, k0 = TK * KW;
        const int jmaxp = (j0 + PJ < NJ ? j0 + PJ : NJ) - 1, kmaxp = (k0 + KW < NK ? k0 + KW : NK) - 1;
        int ja = j0 - 2, jb = jmaxp + 2, ka = k0 - 2, kb = kmaxp + 2;
        ja = ja < 0 ? 0 : ja; jb = jb > NJ - 1 ? NJ - 1 : jb;
        ka = ka < 0 ? 0 : ka; kb = kb > NK - 1 ? NK - 1 : kb;
        const int ja2 = (rj != prj) ? NJ - 1 - jb : ja, jb2 = (rj != prj) ? NJ - 1 - ja : jb;
        const int ka2 = (rk != prk) ? NK - 1 - kb : ka, kb2 = (rk != prk) ? NK - 1 - ka : kb;
        const int tja = ja2 / PJ, ntj = jb2 / PJ - tja + 1, tka = ka2 / KW, ntk = kb2 / KW - tka + 1;
        const int ia = lane & 3, ib = (lane >> 2) & 3;
        const int* pp = (lane < 16 && ia < ntj && ib < ntk)
                            ? a.sync + 8 + (((size_t)pd * a.batch + z) * a.n_patches + (size_t)((tka + ib) * a.npj + tja + ia)) * (NW + 1) + NW : nullptr;
        bool ok = !pp;
        unsigned long long t0 = 0;
        int spins = 0;
        for (;;) {
            if (!ok) ok = dec_prog(ld_raw(pp)) >= FSM_SLAB_DONE;
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
    // first chunk start; the level counters of the slabs start one level before it
    const int Ls_ = TJ * PJ + TK * KW, Lc = Ls_ - (((Ls_ - (TJ + TK)) % FSM_SLAB_C + FSM_SLAB_C) % FSM_SLAB_C);
    if (tid < NW) s_wlev[tid] = Lc - 1;
    __syncthreads();
    fsm_slab_body<PKR, NW>(a, lds, (slab_lds_flag*)s_wlev, lane, w, e2, dir, TJ, TK, z, slot, Lc);
    __syncthreads();   // every slab has drained its stores and published its word
    if (tid == 0) st_prog(fin, FSM_SLAB_DONE);
    return true;
}

template <int PKR, int NW>
__global__ __launch_bounds__(64 * NW) void fsm_sweep_slab(const SlabArgs a) {
    extern __shared__ __attribute__((aligned(16))) char slab_lds[];
    (void)a;
    for (;;) {
        auto kp = __builtin_amdgcn_kernarg_segment_ptr();   // (arguments re-read per unit: nothing is carried, fsm_sweep_persistent)
        asm volatile("" : "+s"(kp));
        if (!fsm_slab_unit<PKR, NW>(*(const SlabArgs*)kp, (slab_lds_char*)slab_lds)) break;
    }
}

}  // namespace ttcr_amd
