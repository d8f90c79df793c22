"""Multi-process (world_size 2, gloo, CPU) test of the source-sharding path used for N > 1
GPUs: block distribution == get_blk_size (ttcr/Grid3D.h:451-465), local solves, gather to
rank 0.  The local solver injected here is the CPU oracle (tests may use it); on the GPU box
the same code runs with the HIP path and the nccl (RCCL) backend."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blk_sizes_match_reference_round_robin():
    from ttcr_amd.dist import blk_sizes, shard_bounds

    def ref(n_tx, n_threads):  # literal get_blk_size
        n_blk = min(n_threads, n_tx)
        blk = [0] * n_blk
        nj = n_tx
        while nj > 0:
            for n in range(n_blk):
                blk[n] += 1
                nj -= 1
                if nj == 0:
                    break
        return blk

    for n_tx in (1, 2, 7, 8, 9, 64, 65):
        for nt in (1, 2, 4, 8):
            assert blk_sizes(n_tx, nt) == ref(n_tx, nt)
            covered = []
            for r in range(nt):
                lo, hi = shard_bounds(n_tx, nt, r)
                covered += list(range(lo, hi))
            assert covered == list(range(n_tx))


def test_unique_sources_first_occurrence_order():
    from ttcr_amd.dist import unique_sources

    src = np.array([[5., 5, 5], [1, 1, 1], [5, 5, 5], [3, 3, 3], [1, 1, 1]])
    u, which = unique_sources(src)
    np.testing.assert_array_equal(u, [[5, 5, 5], [1, 1, 1], [3, 3, 3]])
    np.testing.assert_array_equal(which, [0, 1, 0, 2, 1])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import cases
    from oracle import oracle as O
    from ttcr_amd.dist import broadcast_slowness, raytrace_sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 17
    dx = 20.0 / (n - 1)
    # slowness lives on rank 0 and is broadcast (RCCL broadcast on the GPU box)
    s = torch.zeros(n ** 3, dtype=torch.float64)
    if rank == 0:
        s.copy_(torch.from_numpy(cases.random3d((n, n, n), seed=5)))
    broadcast_slowness(s)
    srcs = cases.mt_sources(5)
    rcv1 = cases.rcv_lattice3d(n=5)
    source = np.repeat(srcs, rcv1.shape[0], axis=0)
    rcv = np.tile(rcv1, (srcs.shape[0], 1))
    solved = []

    def solve_fn(src_rows, rcv_rows):
        out = np.zeros(src_rows.shape[0])
        for p in np.unique(src_rows, axis=0):
            m = np.all(src_rows == p, axis=1)
            solved.append(tuple(p))
            out[m] = O.solve3d(np.float64, (n - 1,) * 3, dx, (0, 0, 0), s.numpy(), [p], rcv=rcv_rows[m])["tt_rcv"]
        return out

    tt = raytrace_sharded(source, rcv, solve_fn)
    q.put((rank, solved, None if tt is None else tt.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_raytrace_world2_gloo():
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from oracle import oracle as O

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, solved, tt = q.get(timeout=240)
        res[r] = (solved, tt)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # block distribution of 5 sources over 2 ranks: 3 + 2, in order
    srcs = cases.mt_sources(5)
    assert [tuple(p) for p in srcs[:3]] == sorted(res[0][0], key=lambda t: [tuple(x) for x in srcs].index(t))
    assert len(res[1][0]) == 2 and res[1][1] is None
    # rank 0 holds every receiver traveltime, equal to a single-process solve
    n = 17
    dx = 20.0 / (n - 1)
    s = cases.random3d((n, n, n), seed=5)
    rcv1 = cases.rcv_lattice3d(n=5)
    want = np.concatenate([O.solve3d(np.float64, (n - 1,) * 3, dx, (0, 0, 0), s, [p], rcv=rcv1)["tt_rcv"] for p in srcs])
    np.testing.assert_array_equal(np.array(res[0][1]), want)
