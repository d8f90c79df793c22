/* oracle/fsm_oracle.h -- TEST INFRASTRUCTURE ONLY (see fsm_oracle_impl.h).
 * C interface of the CPU restatement of the reference FSM solver.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it. */
#ifndef FSM_ORACLE_H
#define FSM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSM_ORACLE_DECL(REAL, S)                                                                     \
    typedef struct {                                                                                 \
        size_t nnx, nny, nnz; /* node counts */                                                      \
        REAL dx, xmin, ymin, zmin, xmax, ymax, zmax, ox, oy, oz;                                     \
    } fsm_grid3d_##S;                                                                                \
    typedef struct {                                                                                 \
        size_t nnx, nnz;                                                                             \
        REAL dx, dz, xmin, zmin, xmax, zmax;                                                         \
    } fsm_grid2d_##S;                                                                                \
    void fsm_grid3d_init_##S(fsm_grid3d_##S* g, uint32_t ncx, uint32_t ncy, uint32_t ncz, REAL dx,   \
                             REAL xmin, REAL ymin, REAL zmin, int translate);                        \
    int fsm_outside3d_##S(const fsm_grid3d_##S* g, int n, const REAL* p);                            \
    void fsm_cells_to_nodes3d_##S(size_t ncx, size_t ncy, size_t ncz, const REAL* sc, REAL* sn);     \
    int fsm_solve3d_##S(const fsm_grid3d_##S* g, const REAL* s, int n_src, const REAL* src,          \
                        const REAL* t0, REAL eps, int maxit, int weno, REAL* T, REAL* change_hist,   \
                        int* niterw_out);                                                            \
    REAL fsm_interp3d_##S(const fsm_grid3d_##S* g, const REAL* T, REAL px, REAL py, REAL pz);        \
    REAL fsm_compute_slowness3d_##S(const fsm_grid3d_##S* g, const REAL* sn, REAL px, REAL py,       \
                                    REAL pz, int iv);                                                \
    REAL fsm_compute_slowness2d_##S(const fsm_grid2d_##S* g, const REAL* sn, REAL px, REAL pz);      \
    int fsm_tt_from_raypath3d_##S(const fsm_grid3d_##S* g, const REAL* sn, const REAL* T, int n_src, \
                                  const REAL* src, const REAL* t0, const REAL rx[3], int iv,         \
                                  long max_steps, REAL* tt_out);                                     \
    int fsm_raypath3d_##S(const fsm_grid3d_##S* g, const REAL* sn, const REAL* T, int n_src,         \
                          const REAL* src, const REAL* t0, const REAL rx[3], int iv, long max_steps, \
                          REAL* tt_out, REAL* pts, long cap, long* npts);                            \
    int fsm_raypath3d_m_##S(const fsm_grid3d_##S* g, const REAL* sn, const REAL* T, int n_src,       \
                            const REAL* src, const REAL* t0, const REAL rx[3], int iv, long max_steps, \
                            REAL* tt_out, long long* mj, REAL* mv, long cap, long* nm);              \
    int fsm_raypath3d_rm_##S(const fsm_grid3d_##S* g, const REAL* sn, const REAL* T, int n_src,      \
                             const REAL* src, const REAL* t0, const REAL rx[3], int iv, long max_steps, \
                             REAL* tt_out, REAL* pts, long cap_pts, long* npts, long long* mj, REAL* mv, \
                             long cap, long* nm);                                                    \
    void fsm_grid2d_init_##S(fsm_grid2d_##S* g, uint32_t ncx, uint32_t ncz, REAL dx, REAL dz,        \
                             REAL xmin, REAL zmin);                                                  \
    int fsm_outside2d_##S(const fsm_grid2d_##S* g, int n, const REAL* p);                            \
    void fsm_cells_to_nodes2d_##S(size_t ncx, size_t ncz, const REAL* sc, REAL* sn);                 \
    int fsm_solve2d_##S(const fsm_grid2d_##S* g, const REAL* s, int n_src, const REAL* src,          \
                        const REAL* t0, REAL eps, int maxit, int weno, REAL* T, REAL* change_hist,   \
                        int* niterw_out);                                                            \
    REAL fsm_interp2d_##S(const fsm_grid2d_##S* g, const REAL* T, REAL px, REAL pz);                 \
    int fsm_raypath2d_##S(const fsm_grid2d_##S* g, const REAL* sn, const REAL* sc, const REAL* T,    \
                          int n_src, const REAL* src, const REAL* t0, const REAL rx[2], int record,  \
                          long max_steps, REAL* tt_out, REAL* pts, long cap, long* npts);                   \
    int fsm_raypath2d_l_##S(const fsm_grid2d_##S* g, const REAL* sn, const REAL* sc, const REAL* T,  \
                            int n_src, const REAL* src, const REAL* t0, const REAL rx[2],            \
                            int with_rays, long max_steps, REAL* tt_out, REAL* pts, long cap,        \
                            long* npts, long long* lcell, REAL* lval, long lcap, long* nlen);

FSM_ORACLE_DECL(float, f32)
FSM_ORACLE_DECL(double, f64)

#ifdef __cplusplus
}
#endif
#endif
