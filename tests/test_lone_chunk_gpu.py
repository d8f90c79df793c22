"""3-D grids that keep one field per workgroup run on chunks of 16 levels (option "lone_chunk", default 16) instead of 8 where the
longer chunk is faster -- fp32 first-order sweeps at any batch size below the pairing threshold, fp64 first-order sweeps of up to four
sources, the WENO stage of one or two sources: the partial order of the node updates is the same (Grid3Drn::sweep /
update_node, ttcr/Grid3Drn.h:2816-2959), so fields, iteration counts and the change history are those of the 8-level kernels bit
for bit (which test_parity_gpu.py pins to the oracle) -- with and without exact skipping, for one source and for a batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(shape, s, src, chunk, weno=0, skip=0, dtype=np.float32):
    import ttcr_amd
    dx = 0.37
    n = src.shape[0]
    g = ttcr_amd.Grid3d(*(np.arange(m) * dx for m in shape), n_threads=n, cell_slowness=0, method="FSM", tt_from_rp=0, weno=weno,
                        dtype=dtype)
    g.set_option("lone_chunk", chunk)
    g.set_option("skip", skip)
    g.set_slowness(s.astype(dtype))
    g.raytrace(src, np.zeros((n, 3)))
    return [(g.get_grid_traveltimes(i).copy(), g.get_niter(i), tuple(g.get_changes(i)[0])) for i in range(n)], g.last_kernel()


def _chunk_of(kernel):   # fsm_sweep_persistent<T,PJ,PK,C,...>
    return int(kernel[kernel.index("<") + 1:].split(",")[3])


def _model(rng, shape, kind):
    nx, ny, nz = shape
    if kind == 0:
        return rng.uniform(0.25, 1.0, shape).astype(np.float32)
    if kind == 1:
        return np.broadcast_to((1.0 / (1.0 + 0.1 * np.arange(nz) * 0.37)).astype(np.float32), shape).copy()
    b = rng.uniform(0.25, 1.0, tuple((v + 7) // 8 for v in shape)).astype(np.float32)
    return np.repeat(np.repeat(np.repeat(b, 8, 0), 8, 1), 8, 2)[:nx, :ny, :nz].copy()


@pytest.mark.parametrize("seed", [3, 17])
def test_chunks_of_16_levels_match_chunks_of_8(seed):
    rng = np.random.default_rng(seed)
    ran = 0
    for c in range(12):
        nx = int(rng.choice([2, 5, 16, 17, 33, 40, 71, 130]))
        ny, nz = (int(v) for v in rng.integers(2, 120, 2)) if c % 3 else (int(v) for v in rng.choice([2, 15, 16, 17, 32, 33, 65], 2))
        weno = 1 if c % 5 == 4 else 0
        shape = tuple(max(v, 5) for v in (nx, ny, nz)) if weno else (nx, ny, nz)   # (the WENO stage needs three cells per axis)
        s = _model(rng, shape, c % 3)
        nsrc = (1, 1, 3, 5)[c % 4]
        src = rng.uniform(0, 1, (nsrc, 3)) * (np.array(shape) - 1) * 0.37
        if c % 4 == 0:
            src[0] = np.round(src[0] / 0.37) * 0.37   # on a node
        skip = (c // 2) % 2
        dtype = np.float64 if c % 4 in (1, 2) else np.float32   # (fp64: one source and three)
        r8, k8 = _solve(shape, s, src, 8, weno, skip, dtype)
        r16, k16 = _solve(shape, s, src, 16, weno, skip, dtype)
        for (t8, n8, c8), (t16, n16, c16) in zip(r8, r16):
            assert n8 == n16 and np.array_equal(t8, t16), (shape, c, k8, k16)
            assert len(c8) == len(c16) and np.allclose(c8, c16, rtol=1e-6), (shape, c8, c16)
        if weno and nsrc == 1:   # (the WENO stage of a lone source: chunks of 16 levels as well)
            assert _chunk_of(k8) == 8 and _chunk_of(k16) == 16 and ",2,1," in k16, (k8, k16)
        if not weno:   # (weno: the last kernel launched is the WENO stage's)
            assert _chunk_of(k8) == 8 and _chunk_of(k16) == 16, (k8, k16)
            assert ("true,true,1,1" in k16) == bool(skip), k16
            ran += 1
    assert ran >= 8


def test_paired_grids_keep_chunks_of_8():
    import os
    import ttcr_amd
    if os.environ.get("TTCR_FSM_PAIR") == "0":
        pytest.skip("pair layout forced off")
    n = 40
    x = np.arange(n) * 0.37
    os.environ["TTCR_FSM_PAIR"] = "1"
    try:
        g = ttcr_amd.Grid3d(x, x, x, n_threads=4, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32)
    finally:
        del os.environ["TTCR_FSM_PAIR"]
    g.set_slowness(np.full((n, n, n), 0.5, np.float32))
    src = np.random.default_rng(1).uniform(1.0, 13.0, (4, 3))
    g.raytrace(src, np.zeros((4, 3)))
    assert ",16,16,8," in g.last_kernel() and ",1,2," in g.last_kernel(), g.last_kernel()
