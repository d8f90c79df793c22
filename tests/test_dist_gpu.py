"""The source-sharded path with the HIP solver under world_size 2 (-m gpu): two processes, one per rank, launched
the way the driver launches bench.py.  With two visible devices the backend is nccl (RCCL over xGMI, one GPU per rank);
on a one-GPU box both ranks share device 0 and the collectives run over gloo on host tensors -- the sharding, the local
HIP solves and the gather are the same code (ttcr_amd/dist.py).  Checked row for row against the single-process solve,
and bench.py itself is run under two ranks (its JSON line must say n_gpus = 2 and count every source once)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import torch, torch.distributed as dist
import cases, ttcr_amd
from ttcr_amd.dist import broadcast_slowness, raytrace_sharded, shard_bounds
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
ndev = torch.cuda.device_count()
backend = 'nccl' if ndev >= world else 'gloo'
dev_id = rank if backend == 'nccl' else 0
torch.cuda.set_device(dev_id)
dev = torch.device('cuda', dev_id)
if backend == 'nccl':
    dist.init_process_group('nccl', device_id=dev)
else:
    dist.init_process_group('gloo')
cdev = dev if backend == 'nccl' else torch.device('cpu')
n = 48
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
# the model lives on rank 0 and is broadcast; handed to the solver as a device pointer
s = torch.zeros(n ** 3, dtype=torch.float32, device=cdev)
if rank == 0:
    s.copy_(torch.from_numpy(cases.random3d((n, n, n), seed=5).astype(np.float32)))
broadcast_slowness(s)
s_dev = s.to(dev)
srcs = cases.mt_sources(7)
rcv1 = cases.rcv_lattice3d(n=5)
source = np.repeat(srcs, rcv1.shape[0], axis=0)
rcv = np.tile(rcv1, (srcs.shape[0], 1))
lo, hi = shard_bounds(7, world, rank)
g = ttcr_amd.Grid3d(x, x, x, n_threads=max(hi - lo, 1), cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32, device=dev_id)
g.set_slowness_device(s_dev.data_ptr(), s_dev.numel())
calls = []
def solve_fn(src_rows, rcv_rows):
    calls.append(int(src_rows.shape[0]))
    return g.raytrace(src_rows, rcv_rows)          # the HIP path
tt = raytrace_sharded(source, rcv, solve_fn, device=cdev, dtype=np.float32)
out = dict(rank=rank, backend=backend, ndev=ndev, rows=calls, lib=ttcr_amd._lib.LIB_PATH, tt=None if tt is None else [float(v) for v in tt],
           niter=[g.get_niter(i) for i in range(hi - lo)])
with open(os.path.join(%(out)r, 'rank%%d.json' %% rank), 'w') as f:
    json.dump(out, f)
dist.barrier()
dist.destroy_process_group()
"""


def test_sharded_raytrace_world2_hip(tmp_path, oracle):
    import cases

    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    # 7 sources over 2 ranks: 4 + 3 (get_blk_size), 25 receivers each; every rank solved with the HIP library
    assert res[0]["rows"] == [100] and res[1]["rows"] == [75]
    assert all(x["lib"].endswith("libttcr_amd.so") for x in res)
    # which collective backend ran: RCCL (nccl) whenever the box shows one GPU per rank, gloo only on a one-GPU box
    import torch
    want_backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    assert [x["backend"] for x in res] == [want_backend] * 2, (res[0]["backend"], res[0]["ndev"])
    print("collective backend of the two ranks:", want_backend, "(devices visible: %d)" % res[0]["ndev"])
    assert res[1]["tt"] is None
    n = 48
    dx = 20.0 / (n - 1)
    s = cases.random3d((n, n, n), seed=5).astype(np.float32)
    srcs = cases.mt_sources(7)
    rcv1 = cases.rcv_lattice3d(n=5)
    o = [oracle.solve3d(np.float32, (n - 1,) * 3, dx, (0, 0, 0), s, [p], rcv=rcv1) for p in srcs]
    np.testing.assert_array_equal(np.array(res[0]["tt"], dtype=np.float32), np.concatenate([q["tt_rcv"] for q in o]))
    assert res[0]["niter"] + res[1]["niter"] == [q["niter"] for q in o]


def test_bench_runs_under_two_ranks():
    """bench.py launched as the driver launches it for N = 2 (small grid, few sources): one JSON line from rank 0,
    n_gpus = 2, every source counted once, the per-rank split reported."""
    import torch

    port = _free_port()
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "128",
           "--sources", "6", "--backend", backend, "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["config"]["sources_total"] == 6 and d["config"]["sources_per_rank"] == [3, 3]
    assert d["value"] > 0 and d["sources_per_s"] > 0
    # the line says which collective backend ran and on which device every rank sat: nccl (= RCCL) whenever the box has a GPU per rank --
    # gloo is only ever accepted on a one-GPU box, where both ranks share device 0
    assert d["config"]["collective_backend"] == backend and d["config"]["world_size"] == 2 and len(d["config"]["devices"]) == 2
    if torch.cuda.device_count() >= 2:
        assert d["config"]["collective_backend"] == "nccl" and d["config"]["devices"][0].split("uuid")[-1] != d["config"]["devices"][1].split("uuid")[-1]
    # whole-job value = nodes x iterations of ALL sources / max-over-ranks time
    it = d["config"]["sweep_iterations_per_source"]
    assert len(it) == 1
    want = 128 ** 3 * it[0] * 6 * 2 / (d["ms_per_step"] * 1e-3 * 2) / 1e6
    assert abs(d["value"] - want) / want < 0.02


def test_get_s0_matches_oracle(oracle):
    """Grid3d.get_s0 / Grid2d.get_s0 (rgrid.pyx:758-826, :3735-3802): slowness at the first row of every event"""
    import ttcr_amd

    rng = np.random.default_rng(41)
    n = 16
    x = 100.0 + np.arange(n) * 0.75
    s = rng.uniform(0.3, 1.0, (n - 1, n - 1, n - 1))
    for tr in (0, 1):
        g = ttcr_amd.Grid3d(x, x, x, cell_slowness=1, method="FSM", translate_grid=tr)
        hypo = np.zeros((6, 5))
        hypo[:, 0] = [3, 1, 3, 2, 1, 3]
        hypo[:, 2:] = rng.uniform(x[0], x[-1], (6, 3))
        hypo[3, 2:] = [x[2], x[3], x[1]]   # an event on a node
        s0 = g.get_s0(hypo, slowness=s)
        first = {1: 1, 2: 3, 3: 0}
        want = oracle.compute_slowness3d(np.float64, (n - 1,) * 3, 0.75, (100.0,) * 3, s.flatten("F"), hypo[[first[1], first[2], first[3]], 2:],
                                         cell_slowness=True, translate=bool(tr))
        for k, e in enumerate((1, 2, 3)):
            assert np.all(s0[hypo[:, 0] == e] == want[k])
    x2, z2 = np.arange(12) * 0.5, np.arange(9) * 0.25
    s2 = rng.uniform(0.3, 1.0, (12, 9))
    g2 = ttcr_amd.Grid2d(x2, z2, cell_slowness=0, method="FSM", dtype=np.float32)
    hypo2 = np.zeros((4, 4))
    hypo2[:, 0] = [7, 7, 9, 8]
    hypo2[:, 2:] = [[1.3, 0.9], [2.0, 1.0], [0.0, 0.0], [5.5, 2.0]]
    s02 = g2.get_s0(hypo2, slowness=s2)
    want2 = oracle.compute_slowness2d(np.float32, (11, 8), 0.5, 0.25, (0, 0), s2.ravel(), hypo2[[0, 3, 2], 2:])
    assert s02[0] == want2[0] and s02[1] == want2[0] and s02[3] == want2[1] and s02[2] == want2[2]
    with pytest.raises(ValueError, match="hypo should be"):
        g2.get_s0(np.zeros((2, 5)))


RCCL_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import torch, torch.distributed as dist
import cases, ttcr_amd
from ttcr_amd.dist import broadcast_slowness, raytrace_sharded
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)          # backend "nccl" IS RCCL on ROCm
n = 48
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
s = torch.from_numpy(cases.random3d((n, n, n), seed=5).astype(np.float32)).to(dev)
broadcast_slowness(s, always=True)                                             # an RCCL broadcast on the device buffer
srcs = cases.mt_sources(3)
rcv1 = cases.rcv_lattice3d(n=5)
source = np.repeat(srcs, rcv1.shape[0], axis=0)
rcv = np.tile(rcv1, (srcs.shape[0], 1))
g = ttcr_amd.Grid3d(x, x, x, n_threads=3, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32, device=0)
g.set_slowness_device(s.data_ptr(), s.numel())
tt = raytrace_sharded(source, rcv, g.raytrace, device=dev, dtype=np.float32, always_gather=True)   # an RCCL all_gather of the receiver rows
plain = g.raytrace(source, rcv)
maps = open('/proc/self/maps').read()
out = dict(backend=dist.get_backend(), world=dist.get_world_size(), same=bool(np.array_equal(tt, plain)), finite=bool(np.all(np.isfinite(tt))),
           rccl_mapped=sorted({l.split('/')[-1] for l in maps.splitlines() if 'librccl' in l}), lib=ttcr_amd._lib.LIB_PATH)
dist.barrier()
dist.destroy_process_group()
print('RCCL_WORKER ' + json.dumps(out))
"""


def test_rccl_backend_runs_on_one_gpu(tmp_path, capsys):
    """torch.distributed with backend "nccl" (= RCCL) in a group of ONE rank on this box's GPU: the library is loaded, a communicator is
    created, and the two collectives of the source-sharded path -- the broadcast of the model, the all_gather of the receiver
    traveltimes (ttcr_amd/dist.py) -- run through it on device buffers around the HIP solves.  An 8-GPU node is not ours to lease; this is
    the part of the RCCL path a one-GPU box can execute (the world_size-2 tests above say which backend THEY ran over)."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_WORKER ")][-1]
    out = json.loads(line[len("RCCL_WORKER "):])
    with capsys.disabled():
        print(f"\n[dist] collectives of the sharded path over backend {out['backend']!r}, world {out['world']}; mapped: {out['rccl_mapped']}")
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["same"] and out["finite"]
    assert out["rccl_mapped"], "librccl is not mapped into the process that ran the collectives"
