// ttcr_amd/csrc/fsm_piped_kernels.h -- first-order 3-D sweeps of fp32 grids (one field per slot) with the staging OFF the march:
// a workgroup is four MARCH wavefronts (the 16 x 16 columns of a patch, one column per lane, exactly the level march of
// fsm_sweep_persistent) and two STAGING wavefronts that move everything between HBM and the LDS tile -- the not-yet-swept
// values of the next chunk, its upwind halo, the write-back of the previous chunk, the progress word, the polls.
//
// What it computes: Grid3Drn::sweep + update_node (ttcr/Grid3Drn.h:2816-2959) -- the same partial order (level L = i' + j' + k'
// of the oriented indices, chunks of C = 8 levels, 16 x 16-column patches), the same arithmetic (update3), the same
// synchronisation words (tickets, progress words with launch epochs, ticket order by expected start time, wait for the 3 x 3
// patches of the previous sweep) as fsm_sweep_persistent<float,16,16,8,true,false,1,1,true,...>: any linear extension of the
// sweep's partial order gives the serial Gauss-Seidel result bit for bit.
//
// Why (profiles/r05/lone_source_chunk_ramp.txt): on the critical path of a lone source a chunk of the four-wave
// kernel takes 4.5 us of which 2.9 us are the march: between two marches every wavefront waits for the upwind progress word
// (poll round trip 0.5-0.8 us), stages (LDS writes of the prefetched values, upwind halo loads and their latency, barrier: 0.6-0.8 us)
// and writes back (0.3 us), and the progress of a chunk only goes out behind the NEXT chunk's wait and staging (1.2 us after its
// march).  A hop of the patch wavefront is three such chunks: 13.7-15 us where 16 levels of march are 5.8 us.  Here two tiles
// alternate: while the march wavefronts work on tile b, the staging wavefronts write back tile b^1 (the chunk before), publish
// it, poll for the chunk after, load it and fill tile b^1 -- between two marches only the carry of the own columns (two LDS
// writes), one barrier and the reload of the column registers are left.
//
// Synchronisation inside the workgroup: s_barrier counts every wavefront of the workgroup, so the staging wavefronts take part in
// the barrier of every level: per chunk all six wavefronts pass exactly nine barriers, and the staging work is cut into the phases
// between them (none of which should wait for memory that was requested in the same or the previous phases, except where the
// march would have to wait anyway: the upwind patch is behind).
//
// Where it stands (profiles/r05/piped_kernel.txt): bit-identical, 7.9-8.0 ms per sweep-iteration for the lone 512^3 source against
// 7.2 ms of the four-wave kernel -- opt-in (option "piped").  What keeps it there: a staging wavefront is ONE instruction stream
// (6-8 cycles per instruction), the compiler waits for ALL outstanding accesses (vmcnt(0)) wherever a loaded value is used behind a
// branch, and the write-through stores of the write-back take ~1 us to be acknowledged: a poll of the upwind progress two levels
// behind the stores sits out their acknowledgement, and the march wavefronts sit at the level's barrier with it.
#pragma once
#include "fsm_kernels.h"

namespace ttcr_amd {

typedef unsigned int piped_u4 __attribute__((ext_vector_type(4)));
constexpr int FSM_PIPED_GUARD = 16;   // elements the host keeps allocated in front of / behind the fields (16-byte accesses of column ends)

#ifndef FSM_PIPED_PROF
#define FSM_PIPED_PROF 0   // 1: thread 0 sums the time it spends AT each barrier of a chunk (B0, levels 0 .. 7) and in the rest of the chunk;
                           // TTCR_FSM_PROF=1 TTCR_FSM_PROF_TRACE=file: words 0 .. 9 of the file, 10: chunks
#endif
#ifndef FSM_PIPED_WAVES
#define FSM_PIPED_WAVES 4   // resident wavefronts per SIMD asked of the compiler (128 registers: two workgroups of five wavefronts fit a CU however their wavefronts fall on the SIMDs)
#endif
#ifndef FSM_PIPED_EXP
#define FSM_PIPED_EXP 0   // TIMING experiments (wrong results): 1: no unit waits for another unit; 2: no write-back; 4: no loads of the next chunk (tiles keep
                          // what they hold); 8: no tile fill
#endif

// One work unit.  Returns false when the tickets of the launch have run out.
__device__ __forceinline__ bool fsm_piped_unit(const PersistArgs<float>& pa) {
    using T = float;
    constexpr int PJ = 16, PK = 16, C = 8, H = 1;
    constexpr int NM = PJ * PK;                 // march threads
    constexpr int RJ = PJ + 2, NROWS = RJ * (PK + 2), NQ = C + 2, RS = NQ | 1;
    constexpr int NSTAT = NM + PJ + PK;         // columns with not-yet-swept values: own + downwind halo
    constexpr int NPIECE = 2 * NSTAT;           // pieces of four levels
    constexpr int NSW = 2;                      // staging wavefronts: lanes 0 .. 127, half of the pieces each
    constexpr int NOWNP = 2 * NM / (64 * NSW);  // passes of a staging lane over the pieces of the own columns (4)
    constexpr int NPASS = NOWNP + 1;            // ... + one over the downwind halo (first staging wavefront) (5)
    constexpr int NWB = NOWNP;                  // write-back passes (4): the pieces of the own columns one level lower
    static_assert(NPIECE <= 64 * (NSW * NOWNP + 1), "pieces fit the passes");
    const SweepArgs<T>& a = pa.s;
    const unsigned epoch = (unsigned)pa.iter_ptr[1];
    const int e2 = (int)(epoch % 3u) + 1;
    auto dec_prog = [&](int raw_) -> int { return (int)((unsigned)raw_ >> 30) == e2 ? (raw_ & 0x3fffffff) : 0; };
    auto ld_prog = [&](const int* p_) -> int { return dec_prog(__hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
    auto st_prog = [&](int* p_, int v_) {
        __hip_atomic_store(p_, (int)(((unsigned)e2 << 30) | (unsigned)v_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    __shared__ float Tt[2][NROWS * RS];
    __shared__ int s_ticket, s_abort;
    __shared__ int s_chg[2];    // tile b: some node of its chunk was accepted (set by the march wavefronts, read and cleared by the staging ones)
    __shared__ int s_pubcnt;    // staging wavefronts whose write-back stores of the chunk to publish have drained (the last one publishes)

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // (nothing derived from it is carried from unit to unit, see fsm_sweep_persistent)
    const bool stager = tid >= NM;
    const int sl = tid - NM;        // lane of the staging wavefront
    const int NF = a.g.NF, NJ = a.g.NJ, NK = a.g.NK, npj = a.g.npj;

    __syncthreads();   // (every read of the previous unit's shared state is over)
    if (tid == NM) {
        const int t_ = atomicAdd(pa.sync + (epoch & 3u), 1);
        if (t_ == 0) __hip_atomic_store(pa.sync + ((epoch + 2u) & 3u), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ticket = t_;
        s_chg[0] = 0; s_chg[1] = 0; s_pubcnt = 0;
    }
    if (tid == NM + 1) s_abort = __hip_atomic_load(pa.sync + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_abort) return false;
    const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);
    const int oidx = ticket / pa.batch, z = ticket - oidx * pa.batch;
    if (oidx >= pa.n_patches * pa.ndir) return false;
    const uint32_t tile = pa.order[oidx];
    const int dir = (int)(tile >> 28), TJ = (int)(tile & 0x3fffu), TK = (int)((tile >> 14) & 0x3fffu);
    int* prog = pa.sync + 8 + ((size_t)dir * pa.batch + z) * pa.n_patches;
    int* my_prog = prog + (TK * npj + TJ);
    const int rf = dir & 1, rj = (dir >> 1) & 1, rk = (dir >> 2) & 1;   // ttcr/Grid3Drn.h:2816-2899
    const int rev = rk, fam = (rf ^ rk) | ((rj ^ rk) << 1);
    const T* __restrict__ Sg = pa.ssh + (size_t)fam * pa.ssh_stride;
    const int grp = __builtin_amdgcn_readfirstlane(a.slots[z]);   // (uniform: the buffer descriptor of the field lives in scalar registers)
    if (grp < 0) {   // converged source: nothing to do, but never leave a waiter hanging
        if (tid == NM) st_prog(my_prog, 0x3fffffff);
        return true;
    }
    const int* up_j = (!(FSM_PIPED_EXP & 1) && TJ > 0) ? prog + (TK * npj + TJ - 1) : nullptr;
    const int* up_k = (!(FSM_PIPED_EXP & 1) && TK > 0) ? prog + ((TK - 1) * npj + TJ) : nullptr;
    const int j0 = TJ * PJ, k0 = TK * PK;
    const int jmaxp = (j0 + PJ < NJ ? j0 + PJ : NJ) - 1, kmaxp = (k0 + PK < NK ? k0 + PK : NK) - 1;
    const int Ls = j0 + k0, Le = jmaxp + kmaxp + NF - 1;
    const int m = TJ + TK;
    const int Lc0 = Ls - (((Ls - m) % C + C) % C);   // first chunk start, congruent to m modulo C (fsm_sweep_persistent)
    T* __restrict__ Tg = a.tt + (size_t)grp * a.g.n_nodes;
    const T INF = real_traits<T>::inf();
    auto lds_row = [&](int cj, int ck) { return ((ck + 1) * RJ + cj + 1) * RS; };
    auto nat_row = [&](int jq, int kq) { return ((uint32_t)(rk ? NK - 1 - kq : kq) * NJ + (rj ? NJ - 1 - jq : jq)) * NF; };

    // ---- the previous sweep of this iteration: the <= 3 x 3 patches (of ITS oriented partition) that own a column within 2 of ours
    if (!(FSM_PIPED_EXP & 1) && dir > 0 && stager && sl < 16) {
        const int pd = dir - 1;
        const int prj = (pd >> 1) & 1, prk = (pd >> 2) & 1;
        int ja = j0 - 2, jb = jmaxp + 2, ka = k0 - 2, kb = kmaxp + 2;
        ja = ja < 0 ? 0 : ja; jb = jb > NJ - 1 ? NJ - 1 : jb;
        ka = ka < 0 ? 0 : ka; kb = kb > NK - 1 ? NK - 1 : kb;
        const int ja2 = (rj != prj) ? NJ - 1 - jb : ja, jb2 = (rj != prj) ? NJ - 1 - ja : jb;
        const int ka2 = (rk != prk) ? NK - 1 - kb : ka, kb2 = (rk != prk) ? NK - 1 - ka : kb;
        const int tja = ja2 / PJ, ntj = jb2 / PJ - tja + 1, tka = ka2 / PK, ntk = kb2 / PK - tka + 1;
        const int ia = sl & 3, ib = sl >> 2;
        if (ia < ntj && ib < ntk) {
            const int* pp = pa.sync + 8 + ((size_t)pd * pa.batch + z) * pa.n_patches + ((tka + ib) * npj + tja + ia);
            const unsigned long long t0 = wall_clock64();
            int spins = 0;
            while (ld_prog(pp) < 0x3fffffff) {
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

    // (the wavefronts that did not poll: nothing of this unit is read from the field before the previous sweep is done with its columns)
    if (!(FSM_PIPED_EXP & 1) && dir > 0) __syncthreads();

    // =====================================================================================================================
    if (stager) {
        // ---- the staging wavefront ----------------------------------------------------------------------------------------
        // buffer descriptor of the field: FSM_PIPED_GUARD elements in front of and behind it are allocated (a piece at the end of a
        // column reaches up to three elements beyond the field; a negative byte offset would fail the bounds check for the whole access)
        const uint32_t nbytes = (uint32_t)(((size_t)a.g.n_nodes + 2 * FSM_PIPED_GUARD) * sizeof(T));
        __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(Tg - FSM_PIPED_GUARD, 0, nbytes, 0x00020000);
        // piece p = lane + 64 it of a chunk that starts at level L0: column c = p / 2 (own columns first, then the downwind halo), levels
        // L0 + 1 + 4 (p % 2) + t, t = 0 .. 3  <->  tile q = 2 + 4 (p % 2) + t;  i' = level - j' - k'.  Own columns (it < 8): the lane
        // keeps its j' and walks k' in steps of two, so everything about a piece is affine in `it` (three registers, not 3 x 9)
        constexpr int KS = 2 * NSW;   // k' step of a lane from pass to pass (the 128 lanes cover 4 rows of 16 columns x 2 halves)
        const int sw = sl >> 6, sll = sl & 63;   // staging wavefront, lane in it
        const int ph = sl & 1, pcj = (sl >> 1) & 15, pck0 = sl >> 5;
        const int pjq = j0 + pcj, pkq0 = k0 + pck0;
        const int pb0 = 1 + 4 * ph - pjq - pkq0;                     // i' of t = 0 minus L0 at it = 0; - KS per it
        const uint32_t prow0 = nat_row(pjq < NJ ? pjq : NJ - 1, pkq0 < NK ? pkq0 : NK - 1);
        const uint32_t prstep = (uint32_t)((rk ? -KS : KS) * NJ) * (uint32_t)NF;   // natural index of node i = 0 of the column: + prstep per it
        const int plds0 = lds_row(pcj, pck0) + 2 + 4 * ph;           // tile index of t = 0; + KS RJ RS per it
        auto pc_valid = [&](int it) { return pjq < NJ && pkq0 + KS * it < NK; };
        // ... and the 32 downwind halo columns (the last pass of the FIRST staging wavefront): j' = j0 + 16 (16 rows), then k' = k0 + 16 (16 columns)
        int ph_b; uint32_t ph_row; int ph_lds;
        {
            const int c = sll >> 1;
            const int cj = c < PK ? PJ : c - PK, ck = c < PK ? c : PK;
            const int jq = j0 + cj, kq = k0 + ck;
            const bool ok = sw == 0 && jq < NJ && kq < NK;
            ph_b = ok ? 1 + 4 * ph - jq - kq : -(1 << 29);
            ph_row = ok ? nat_row(jq, kq) : 0u;
            ph_lds = lds_row(cj, ck) + 2 + 4 * ph;
        }
        auto pc_b = [&](int it) { return it < NOWNP ? (pc_valid(it) ? pb0 - KS * it : -(1 << 29)) : ph_b; };
        auto pc_row = [&](int it) { return it < NOWNP ? prow0 + (uint32_t)it * prstep : ph_row; };
        auto pc_lds = [&](int it) { return it < NOWNP ? plds0 + it * (KS * RJ * RS) : ph_lds; };
        // upwind halo: 32 columns (j' = j0 - 1: 16 rows; k' = k0 - 1: 16 columns) x levels L0 - 1 + t  <->  tile q = t, t = 0 .. 7: a piece
        // of four per lane
        int hu_b; uint32_t hu_row; int hu_lds;   // (the SECOND staging wavefront's: it is the one that follows the upwind progress)
        {
            const int c = sll >> 1, h = sll & 1;
            int cj, ck;
            if (c < PK) { cj = -1; ck = c; } else { cj = c - PK; ck = -1; }
            const int jq = j0 + cj, kq = k0 + ck;
            const bool ok = sw == NSW - 1 && jq >= 0 && jq < NJ && kq >= 0 && kq < NK;
            hu_b = ok ? -1 + 4 * h - jq - kq : -(1 << 29);
            hu_row = ok ? nat_row(jq, kq) : 0u;
            hu_lds = lds_row(cj, ck) + 4 * h;
        }
        // the four levels i', i' + 1 .. of a piece lie at ascending (rf = 0) or descending (rf = 1) addresses: one 16-byte access from
        // the lowest one, components in level order or reversed
        auto piece_off = [&](uint32_t rowbase, int ip0) -> uint32_t {   // byte offset in the descriptor of the lowest address of the piece
            const uint32_t first = rf ? rowbase + (uint32_t)(NF - 1 - ip0 - 3) : rowbase + (uint32_t)ip0;
            return (first + FSM_PIPED_GUARD) * (uint32_t)sizeof(T);
        };
        // The loads of a chunk are issued back to back, without a branch and without touching what they return (a use would make the
        // compiler wait for the load on the spot: ten round trips in a row instead of one); what they return is taken apart when the
        // tile is filled, levels later.  A piece with no level inside its column is asked for at an offset beyond the descriptor:
        // the bounds check answers zeros, and the piece is INF in the tile (the minimum of a neighbour pair ignores it).
        auto piece_issue = [&](uint32_t rowbase, int ip0) -> piped_u4 {
            const bool none = ip0 + 3 < 0 || ip0 >= NF;
            return __builtin_amdgcn_raw_buffer_load_b128(rsT, none ? 0xfffffff0u : piece_off(rowbase, ip0), 0, 16);   // sc1: another XCD may have written it in this launch
        };
        auto piece_take = [&](const piped_u4& w, int ip0, float (&v)[4]) {
            const float f0 = __uint_as_float(rf ? w.w : w.x), f1 = __uint_as_float(rf ? w.z : w.y), f2 = __uint_as_float(rf ? w.y : w.z),
                        f3 = __uint_as_float(rf ? w.x : w.w);
            v[0] = (unsigned)ip0 < (unsigned)NF ? f0 : INF;
            v[1] = (unsigned)(ip0 + 1) < (unsigned)NF ? f1 : INF;
            v[2] = (unsigned)(ip0 + 2) < (unsigned)NF ? f2 : INF;
            v[3] = (unsigned)(ip0 + 3) < (unsigned)NF ? f3 : INF;
        };
        // what chunk L0 needs from HBM, into registers (36 of them for three levels; a landing area in LDS filled by buffer_load ... lds
        // was tried instead: such a load keeps the wavefront for ~150 ns, nine of them for four levels' time, profiles/r05/piped_kernel.txt):
        // the not-yet-swept values of the own and the downwind halo columns (final since the previous sweep: they wait for nobody) ...
        piped_u4 raw[NPASS], hraw;
        auto issue_statics = [&](int L0, int it0, int it1) {
            if (FSM_PIPED_EXP & 4) return;
#pragma unroll
            for (int it = 0; it < NPASS; ++it)
                if (it >= it0 && it < it1 && (it < NOWNP || sw == 0)) raw[it] = piece_issue(pc_row(it), L0 + pc_b(it));
        };
        // ... and the upwind halo (once both upwind patches have published it)
        auto issue_halo = [&](int L0) {
            if ((FSM_PIPED_EXP & 4) || sw != NSW - 1) return;
            hraw = piece_issue(hu_row, L0 + hu_b);
        };
        // ... and from there into tile b
        auto fill_statics = [&](int b, int L0, int it0, int it1) {
            if (FSM_PIPED_EXP & 12) return;
#pragma unroll
            for (int it = 0; it < NPASS; ++it) {
                if (it < it0 || it >= it1 || (it == NOWNP && sw != 0)) continue;
                float v[4];
                piece_take(raw[it], L0 + pc_b(it), v);
#pragma unroll
                for (int t = 0; t < 4; ++t) Tt[b][pc_lds(it) + t] = v[t];
            }
        };
        auto fill_halo = [&](int b, int L0) {
            if ((FSM_PIPED_EXP & 12) || sw != NSW - 1) return;
            float v[4];
            piece_take(hraw, L0 + hu_b, v);
#pragma unroll
            for (int t = 0; t < 4; ++t) Tt[b][hu_lds + t] = v[t];
        };
        // upwind progress for the chunk that starts at L0: both patches have published every level <= L0 + C - 2
        // (lanes 0 / 1 of the storer; the words stay raw in their register until they are looked at -- a decode where they are loaded would be a
        // wait for the load where it is issued)
        int smp = 0, seen = ((sll == 0 && up_j) || (sll == 1 && up_k)) ? 0 : 0x3fffffff;   // newest value seen (progress only grows)
        const int* smp_ptr = sll == 0 ? up_j : (sll == 1 ? up_k : nullptr);
        auto sample = [&]() {
            if (sw == NSW - 1 && smp_ptr) smp = __hip_atomic_load(smp_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto covered = [&](int L0) -> bool {
            if (sw != NSW - 1) return true;   // (the loader has nothing to wait for)
            const int need = L0 + C - 1;
            if (smp_ptr) { const int v = dec_prog(smp); seen = v > seen ? v : seen; }
            return __builtin_amdgcn_ballot_w64(seen < need) == 0ull;
        };
        auto wait_upwind = [&](int L0) {   // blocking
            unsigned long long t0 = 0;
            int spins = 0;
            for (;;) {
                if (covered(L0)) break;
                sample();
                if (covered(L0)) break;
                if (spins == 0) t0 = wall_clock64();
                if ((++spins & 63) == 0) {
                    if (__hip_atomic_load(pa.sync + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    if (wall_clock64() - t0 > pa.timeout_ticks) {
                        if (sll == 0) __hip_atomic_store(pa.sync + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(FSM_POLL_SLEEP);
            }
        };
        // write-back of the chunk that started at L0 out of tile b: levels L0 + 4 h + t of the own columns  <->  tile q = 1 + 4 h + t
        auto write_back = [&](int b, int L0, int it0, int it1) {
#pragma unroll
            for (int it = 0; it < NWB; ++it) {
                if (it < it0 || it >= it1 || !pc_valid(it)) continue;
                const int ip0 = L0 + pb0 - 1 - KS * it;               // level L0 + 4 h  (one level below the staging piece of the same lane)
                if (ip0 + 3 < 0 || ip0 >= NF) continue;
                const int lo = plds0 - 1 + it * (KS * RJ * RS);
                const float f0 = Tt[b][lo], f1 = Tt[b][lo + 1], f2 = Tt[b][lo + 2], f3 = Tt[b][lo + 3];
                const uint32_t rowbase = prow0 + (uint32_t)it * prstep;
                if (ip0 >= 0 && ip0 + 3 < NF) {
                    piped_u4 w;
                    w.x = __float_as_uint(rf ? f3 : f0); w.y = __float_as_uint(rf ? f2 : f1); w.z = __float_as_uint(rf ? f1 : f2); w.w = __float_as_uint(rf ? f0 : f3);
                    __builtin_amdgcn_raw_buffer_store_b128(w, rsT, piece_off(rowbase, ip0), 0, 16);
                } else {
                    const float f[4] = {f0, f1, f2, f3};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if ((unsigned)(ip0 + t) < (unsigned)NF) st_sc1(Tg + (rowbase + (uint32_t)(rf ? NF - 1 - (ip0 + t) : ip0 + t)), f[t]);
                }
            }
        };

        // ---- chunks whose pieces all lie inside their columns, of patches with all their halo columns in the grid (nearly all chunks of a
        // large grid): no range test, no INF, no select; the eight own pieces of a lane are ONE register offset plus a scalar step
        // (the soffset operand of the buffer instruction), the tile indices immediates
        const int s_jk = j0 + k0;
        const bool patch_full = TJ > 0 && TK > 0 && j0 + PJ < NJ && k0 + PK < NK;
#ifndef FSM_PIPED_DBG
#define FSM_PIPED_DBG 0   // bisecting builds: 1: no fast path; 2: every load waited for at once; 4: write-back never fast; 8: staging never fast
#endif
        auto is_fast = [&](int Lc_) { return !(FSM_PIPED_DBG & 1) && patch_full && Lc_ + 1 - (s_jk + PJ + PK - 1) >= 0 && Lc_ + 8 - s_jk <= NF - 1; };
        const int fstep = (rk ? -KS : KS) * NJ * NF + (rf ? KS : -KS);   // elements from the piece of pass `it` to that of it + 1
        const uint32_t fstepB = (uint32_t)(fstep < 0 ? -fstep : fstep) * (uint32_t)sizeof(T);
        // element of the lowest address of the lane's own piece at pass 0 (fstep > 0) or the last pass (fstep < 0), for the chunk that starts at 0
        const int fit0 = fstep < 0 ? NOWNP - 1 : 0;
        const uint32_t fbase = rf ? prow0 + (uint32_t)fit0 * prstep + (uint32_t)(NF - 4 - (pb0 - KS * fit0)) : prow0 + (uint32_t)fit0 * prstep + (uint32_t)(pb0 - KS * fit0);
        auto fsoff = [&](int it) -> uint32_t { return (uint32_t)(fstep < 0 ? NOWNP - 1 - it : it) * fstepB; };   // (uniform)
        auto fvoff = [&](uint32_t base_, int Lc_) -> uint32_t { return (rf ? base_ - (uint32_t)Lc_ : base_ + (uint32_t)Lc_) * (uint32_t)sizeof(T) + FSM_PIPED_GUARD * (uint32_t)sizeof(T); };
        const uint32_t fbase_hd = rf ? ph_row + (uint32_t)(NF - 4 - ph_b) : ph_row + (uint32_t)ph_b;   // downwind halo piece (pass 8)
        const uint32_t fbase_hu = rf ? hu_row + (uint32_t)(NF - 4 - hu_b) : hu_row + (uint32_t)hu_b;   // upwind halo piece
        auto issue_statics_fast = [&](int Lc_, int it0, int it1) {
            if (FSM_PIPED_EXP & 4) return;
            const uint32_t vo = fvoff(fbase, Lc_);
#pragma unroll
            for (int it = 0; it < NOWNP; ++it)
                if (it >= it0 && it < it1) raw[it] = __builtin_amdgcn_raw_buffer_load_b128(rsT, vo, fsoff(it), 16);
            if (it1 > NOWNP && sw == 0) raw[NOWNP] = __builtin_amdgcn_raw_buffer_load_b128(rsT, fvoff(fbase_hd, Lc_), 0, 16);
        };
        auto issue_halo_fast = [&](int Lc_) {
            if ((FSM_PIPED_EXP & 4) || sw != NSW - 1) return;
            hraw = __builtin_amdgcn_raw_buffer_load_b128(rsT, fvoff(fbase_hu, Lc_), 0, 16);
        };
        auto put4 = [&](int b, int idx, const piped_u4& w) {   // four consecutive levels of a column into the tile
            Tt[b][idx] = __uint_as_float(rf ? w.w : w.x);
            Tt[b][idx + 1] = __uint_as_float(rf ? w.z : w.y);
            Tt[b][idx + 2] = __uint_as_float(rf ? w.y : w.z);
            Tt[b][idx + 3] = __uint_as_float(rf ? w.x : w.w);
        };
        auto fill_statics_fast = [&](int b, int it0, int it1) {
            if (FSM_PIPED_EXP & 12) return;
#pragma unroll
            for (int it = 0; it < NPASS; ++it) {
                if (it < it0 || it >= it1 || (it == NOWNP && sw != 0)) continue;
                put4(b, it < NOWNP ? plds0 + it * (KS * RJ * RS) : ph_lds, raw[it]);
            }
        };
        auto fill_halo_fast = [&](int b) {
            if ((FSM_PIPED_EXP & 12) || sw != NSW - 1) return;
            put4(b, hu_lds, hraw);
        };
        auto write_back_fast = [&](int b, int Lc_, int it0, int it1) {   // (the pieces of the own columns one level lower)
            const uint32_t vo = fvoff(rf ? fbase + 1u : fbase - 1u, Lc_);
#pragma unroll
            for (int it = 0; it < NWB; ++it) {
                if (it < it0 || it >= it1) continue;
                const int lo = plds0 - 1 + it * (KS * RJ * RS);
                const float f0 = Tt[b][lo], f1 = Tt[b][lo + 1], f2 = Tt[b][lo + 2], f3 = Tt[b][lo + 3];
                piped_u4 w;
                w.x = __float_as_uint(rf ? f3 : f0); w.y = __float_as_uint(rf ? f2 : f1); w.z = __float_as_uint(rf ? f1 : f2); w.w = __float_as_uint(rf ? f0 : f3);
                __builtin_amdgcn_raw_buffer_store_b128(w, rsT, vo, fsoff(it), 16);
                // gfx950: a 16-byte buffer store with a REGISTER soffset still reads its data registers when the next instruction is issued
                // (the compiler models that hazard only for stores without a register soffset and reuses the registers at once: every second
                // piece of a write-back went out with the next piece's values -- found with 41-node columns, profiles/r05/piped_kernel.txt)
                asm volatile("s_nop 1" ::: "memory");
            }
        };

        // ---- first chunk: staged before the march starts.  Its own columns' levels L0 - 1, L0 (tile q = 0, 1) come from HBM too
        int L0 = Lc0;
        issue_statics(L0, 0, NPASS);
        wait_upwind(L0);
        issue_halo(L0);
        {
            // (addresses clamped into the column instead of branches around the loads, for the same reason as above)
            float v0[NM / (64 * NSW)], v1[NM / (64 * NSW)];
#pragma unroll
            for (int it = 0; it < NM / (64 * NSW); ++it) {
                const int c = sl + 64 * NSW * it, cj = c % PJ, ck = c / PJ;
                const int jq = j0 + cj < NJ ? j0 + cj : NJ - 1, kq = k0 + ck < NK ? k0 + ck : NK - 1;
                const int ip = L0 - 1 - jq - kq;
                const int ia = ip < 0 ? 0 : (ip > NF - 1 ? NF - 1 : ip), ib = ip + 1 < 0 ? 0 : (ip + 1 > NF - 1 ? NF - 1 : ip + 1);
                const uint32_t rb = nat_row(jq, kq);
                v0[it] = ld_sc1(Tg + (rb + (uint32_t)(rf ? NF - 1 - ia : ia)));
                v1[it] = ld_sc1(Tg + (rb + (uint32_t)(rf ? NF - 1 - ib : ib)));
            }
#pragma unroll
            for (int it = 0; it < NM / (64 * NSW); ++it) {
                const int c = sl + 64 * NSW * it, cj = c % PJ, ck = c / PJ, jq = j0 + cj, kq = k0 + ck;
                const int ip = L0 - 1 - jq - kq;
                const bool col = jq < NJ && kq < NK;
                Tt[0][lds_row(cj, ck)] = col && (unsigned)ip < (unsigned)NF ? v0[it] : INF;
                Tt[0][lds_row(cj, ck) + 1] = col && (unsigned)(ip + 1) < (unsigned)NF ? v1[it] : INF;
            }
        }
        fill_statics(0, L0, 0, NPASS);
        fill_halo(0, L0);
        int b = 0;
        int pub_pending = 0;   // progress value of the chunk written back last, not yet published
        // The phases of a chunk.  A staging wavefront passes a barrier every level (0.36 us), issues an instruction every 6-8 cycles (60 ns
        // to get a scattered 16-byte load out) and a memory round trip is 1-2 us: the work is cut into pieces of < 100 instructions,
        // one per level and wavefront, none of which waits for an access of the same or the previous two phases -- except where the
        // march has to wait anyway:
        //   0         sample the upwind progress (second wavefront); write back the chunk before this one (tile b^1), first half
        //   1         second half
        //   2, 3      the loads of the next chunk's not-yet-swept values go out; upwind progress there: its halo too (else look again)
        //   4         upwind progress, again
        //   5         the stores of 0, 1 are done once at most the loads issued behind them are outstanding (accesses complete in the order of
        //             their issue): the wavefront that gets there last publishes the chunk before; first pieces into tile b^1
        //   6, 7      the other pieces; the halo (if its loads went out by 4)
        //   behind 7  (else; second wavefront) wait, load, fill
        auto publish = [&](int value, bool drain_all) {
            if (drain_all) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (sw == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // (five / at least four loads behind the stores)
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (sll == 0) {
                if (atomicAdd(&s_pubcnt, 1) == NSW - 1) {
                    s_pubcnt = 0;
                    st_prog(my_prog, value);
                }
            }
        };
        for (;; L0 += C, b ^= 1) {
            const bool last = L0 + C > Le;
            const bool fast_n = !(FSM_PIPED_DBG & 8) && !last && is_fast(L0 + C), fast_p = !(FSM_PIPED_DBG & 4) && L0 > Lc0 && is_fast(L0 - C);
            int halo_at = sw == NSW - 1 ? -1 : 0;   // phase in which the halo loads of the next chunk went out (first wavefront: nothing to do)
            auto try_halo = [&](int phase) {
                if (last || halo_at >= 0) return;
                if (covered(L0 + C)) { if (fast_n) issue_halo_fast(L0 + C); else issue_halo(L0 + C); halo_at = phase; }
                else sample();
            };
            auto statics = [&](int it0, int it1) { if (!last) { if (fast_n) issue_statics_fast(L0 + C, it0, it1); else issue_statics(L0 + C, it0, it1); } };
            auto fills = [&](int it0, int it1) { if (!last) { if (fast_n) fill_statics_fast(b ^ 1, it0, it1); else fill_statics(b ^ 1, L0 + C, it0, it1); } };
            __syncthreads();                                   // B0: tile b complete (march: carry written)
            const bool wb = L0 > Lc0 && !(FSM_PIPED_EXP & 2) && s_chg[b ^ 1] != 0;
            if (!last) sample();
            if (wb) { if (fast_p) write_back_fast(b ^ 1, L0 - C, 0, 2); else write_back(b ^ 1, L0 - C, 0, 2); }
            if (L0 > Lc0) pub_pending = L0;
            __syncthreads();                                   // level 0 done
            if (sl == 0) s_chg[b ^ 1] = 0;
            if (wb) { if (fast_p) write_back_fast(b ^ 1, L0 - C, 2, 4); else write_back(b ^ 1, L0 - C, 2, 4); }
            __syncthreads();                                   // level 1
            statics(0, 3);
            try_halo(2);
            __syncthreads();                                   // level 2
            statics(3, NPASS);
            try_halo(3);
            __syncthreads();                                   // level 3
            try_halo(4);
            __syncthreads();                                   // level 4
            if (pub_pending) { publish(pub_pending, last); pub_pending = 0; }
            fills(0, 2);
            __syncthreads();                                   // level 5
            fills(2, 4);
            __syncthreads();                                   // level 6
            fills(4, NPASS);
            bool filled = sw != NSW - 1;
            if (!last && !filled && halo_at >= 0) { if (fast_n) fill_halo_fast(b ^ 1); else fill_halo(b ^ 1, L0 + C); filled = true; }
            __syncthreads();                                   // level 7: the march of this chunk is over
            if (last) break;
            if (!filled) {   // the upwind patches were not there in time: the march waits (at B0) like the four-wave kernel does
                wait_upwind(L0 + C);
                issue_halo(L0 + C);
                fill_halo(b ^ 1, L0 + C);
            }
        }
        // the last chunk: its flag is final behind one more barrier
        __syncthreads();                                       // BF
        if (s_chg[b]) write_back(b, L0, 0, NWB);
        publish(0x3fffffff, true);
        return true;
    }

    // =====================================================================================================================
    // ---- the march wavefronts: one column per thread, exactly the level march of fsm_sweep_persistent (H = 1, one source) -------
    const int tj = tid % PJ, tk = tid / PJ;
    const int jp = j0 + tj, kp = k0 + tk;
    const bool col_ok = jp < NJ && kp < NK;
    const int row = (tk + 1) * RJ + tj + 1;
    const int jn = rj ? NJ - 1 - jp : jp, kn = rk ? NK - 1 - kp : kp;
    const uint32_t colbase = ((uint32_t)kn * NJ + jn) * NF;
    const T dx = a.dx;
    const uint32_t* __restrict__ Fz = a.frozen + (size_t)grp * a.mask_words;
    const int* bb = a.bbox + 6 * grp;
    const int M = a.g.M;
    // sheared slowness: see issue_static of fsm_sweep_persistent (scalar row pointer + this thread's 32-bit offset)
    const int skx_min = rev ? NK - 1 - kmaxp : k0;
    uint32_t stoff;
    {
        const int kc = kp < NK ? kp : NK - 1, jc = jp < NJ ? jp : NJ - 1;
        const int kx = rev ? NK - 1 - kc : kc, jx = rev ? NJ - 1 - jc : jc;
        stoff = (uint32_t)(((size_t)(kx - skx_min) * shear_plane(a.g) + (size_t)((jx >> 4) * 32 + (jx & 15))) * sizeof(T));
    }
    T sv[C];
    int xs_next = 0, xs_level = -(1 << 30);
    auto load_slowness = [&](int L) {
        int x;
        if (L == xs_level) {
            x = xs_next;
        } else {
            x = (rev ? NF + NJ + NK - 3 - L : L) % M;
            x = x < 0 ? x + M : x;
        }
        x = __builtin_amdgcn_readfirstlane(x);
        int xlo = rev ? x - (C - 1) : x;
        if (xlo < 0) { xlo %= M; xlo = xlo < 0 ? xlo + M : xlo; }
        const uint32_t SRB = (uint32_t)a.g.SR * (uint32_t)sizeof(T);
        const char* rowp = reinterpret_cast<const char*>(Sg + (size_t)skx_min * shear_plane(a.g)) + (size_t)((uint32_t)xlo >> 1) * SRB;
        auto ld_row = [&](int half) -> T {
            asm volatile("" : "+s"(rowp));
            return *reinterpret_cast<const T*>(rowp + half * 16 * (int)sizeof(T) + stoff);
        };
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
        x = rev ? x - C : x + C;
        if (x < 0 || x >= M) { x %= M; x = x < 0 ? x + M : x; }
        xs_next = x;
        xs_level = L + C;
    };
    // chunks that may hold frozen nodes of the source (fsm_sweep_persistent)
    int near_lo, near_hi;
    {
        const int jlo = rj ? NJ - 1 - jmaxp : j0, jhi = rj ? NJ - 1 - j0 : jmaxp;
        const int klo = rk ? NK - 1 - kmaxp : k0, khi = rk ? NK - 1 - k0 : kmaxp;
        const int b0 = rf ? NF - 1 - bb[1] : bb[0], b1 = rf ? NF - 1 - bb[0] : bb[1];
        const bool jk = !(jhi < bb[2] || jlo > bb[3] || khi < bb[4] || klo > bb[5]) && b0 <= NF - 1 && b1 >= 0;
        near_lo = jk ? b0 - (C - 1) + j0 + k0 : 1;
        near_hi = jk ? b1 + jmaxp + kmaxp : 0;
    }
    T dec = 0;
    unsigned long long nevals = 0;
    T carry0 = INF, carry1 = INF;
    unsigned long long pacc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, plast = (FSM_PIPED_PROF && a.prof && tid == 0) ? wall_clock64() : 0ull;
    load_slowness(Lc0);
    int b = 0;
    int L0 = Lc0;
    for (;; L0 += C, b ^= 1) {
        const bool last = L0 + C > Le;
        if (L0 > Lc0) {   // own column: the last result and the next old value of the chunk before (tile q = 0, 1)
            Tt[b][row * RS] = carry0;
            Tt[b][row * RS + 1] = carry1;
        }
        unsigned long long pt0 = 0, pt1 = 0;
        if (FSM_PIPED_PROF && a.prof && tid == 0) pt0 = wall_clock64();
        __syncthreads();                                       // B0
        if (FSM_PIPED_PROF && a.prof && tid == 0) { pt1 = wall_clock64(); pacc[0] += pt1 - pt0; pacc[9] += pt0 - plast; plast = pt1; }
        const int eoff = jp + kp - L0;
        const int ea = col_ok ? (eoff > 0 ? eoff : 0) : C;
        const int eb = eoff + NF - 1 < C - 1 ? eoff + NF - 1 : C - 1;
        nevals += (eb >= ea) ? (unsigned)(eb - ea + 1) : 0u;
        T own[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) own[q] = Tt[b][row * RS + q];
        T sc[C];
#pragma unroll
        for (int q = 0; q < C; ++q) sc[q] = rev ? sv[C - 1 - q] : sv[q];
        if (!last) load_slowness(L0 + C);
        const bool near_src = L0 >= near_lo && L0 <= near_hi;
        bool changed = false;
#pragma unroll
        for (int ee = 0; ee < C; ++ee) {
            const int q = ee + H;
            const bool in_grid = (ee >= ea) & (ee <= eb);
            const unsigned long long grid_lanes = __builtin_amdgcn_sicmp(ea, ee, 41) & __builtin_amdgcn_sicmp(eb, ee, 39);   // SLE, SGE
            const T c = own[q];
            const T jm1 = Tt[b][(row - 1) * RS + q - 1], jp1 = Tt[b][(row + 1) * RS + q + 1];
            const T km1 = Tt[b][(row - RJ) * RS + q - 1], kp1 = Tt[b][(row + RJ) * RS + q + 1];
            bool active = in_grid;
            unsigned long long live_lanes = grid_lanes;
            if (near_src) {   // block-uniform and rare
                int ipn = L0 + ee - jp - kp;
                asm volatile("" : "+v"(ipn));
                if (active) {
                    const uint32_t n = colbase + (rf ? NF - 1 - ipn : ipn);
                    active = !((Fz[n >> 5] >> (n & 31)) & 1u);
                }
                live_lanes = __builtin_amdgcn_ballot_w64(active);
            }
            const T af = vmin(own[q - 1], own[q + 1]);
            const T aj = vmin(jm1, jp1);
            const T ak = vmin(km1, kp1);
            const T t = update3(ak, aj, af, sc[ee], dx, live_lanes);
            const bool acc = active & (t < c);
            const T nv = acc ? t : c;
            dec += acc ? c - t : (T)0;
            changed |= acc;
            own[q] = nv;
            Tt[b][row * RS + q] = nv;
            if (FSM_PIPED_PROF && a.prof && tid == 0) pt0 = wall_clock64();
            __syncthreads();                                   // level ee
            if (FSM_PIPED_PROF && a.prof && tid == 0) { pt1 = wall_clock64(); pacc[1 + ee] += pt1 - pt0; pacc[9] += pt0 - plast; plast = pt1; }
        }
        carry0 = own[C];
        carry1 = own[C + 1];
        if (wave_any(changed) && (tid & 63) == 0) s_chg[b] = 1;
        if (FSM_PIPED_PROF) ++pacc[10];
        if (last) break;
    }
    __syncthreads();                                           // BF (the last chunk's flag)
    if (FSM_PIPED_PROF && a.prof && tid == 0)
        for (int q = 0; q < 11; ++q) atomicAdd(a.prof + 8 + q, pacc[q]);
    // L1 decrease and evaluated updates of the unit
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nevals += __shfl_down(nevals, off, 64);
    double accd = (double)dec;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) accd += __shfl_down(accd, off, 64);
    if ((tid & 63) == 0 && accd != 0.0) atomicAdd(a.change + grp, accd);
    if ((tid & 63) == 0 && nevals) atomicAdd(pa.evals + grp, nevals);
    return true;
}

__global__ __launch_bounds__(384, FSM_PIPED_WAVES) void fsm_sweep_piped(const PersistArgs<float> pa) {
    (void)pa;
    for (;;) {
        auto kp = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        if (!fsm_piped_unit(*(const PersistArgs<float>*)kp)) break;
    }
}

}  // namespace ttcr_amd
