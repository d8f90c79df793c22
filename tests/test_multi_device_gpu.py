"""One grid on several devices behind the C ABI (ttcr_fsm3d_create_multi / ttcr_fsm2d_create_multi, include/ttcr_amd.h):
one replica of the grid per listed device, the slots divided, the sources of a call block-distributed over all slots like
get_blk_size (ttcr/Grid3D.h:451-465, :810-853).  The device list [0, 0] puts two replicas on the one GPU of the test box
(two host threads, two streams, concurrent launches); with more GPUs visible the list names distinct devices as well.
Bar: bit-equal to the single-device grid and to the oracle."""
import threading

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _device_lists():
    import torch

    n = torch.cuda.device_count()
    lists = [[0, 0], [0, 0, 0]]
    if n >= 2:
        lists.append(list(range(min(n, 4))))
    return lists


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_multi_device_3d_equals_single_device(oracle, dt):
    import ttcr_amd

    n = 40
    dx = 0.5
    x = np.arange(n) * dx
    s = np.random.default_rng(11).uniform(0.25, 1.0, (n, n, n)).astype(dt)   # (nx, ny, nz), as ttcrpy takes it
    srcs = cases.mt_sources(9) * (x[-1] / 20.0)
    rcv1 = np.random.default_rng(3).uniform(0.5, x[-1] - 0.5, (6, 3))
    source = np.repeat(srcs, rcv1.shape[0], axis=0)
    rcv = np.tile(rcv1, (srcs.shape[0], 1))
    g1 = ttcr_amd.Grid3d(x, x, x, n_threads=5, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=dt)
    t1 = g1.raytrace(source, rcv, slowness=s)
    want = np.concatenate([oracle.solve3d(dt, (n - 1,) * 3, dx, (0, 0, 0), s.flatten("F"), [p], rcv=rcv1)["tt_rcv"] for p in srcs])
    np.testing.assert_array_equal(t1, want)
    for devs in _device_lists():
        g = ttcr_amd.Grid3d(x, x, x, n_threads=5, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=dt, device=devs)
        assert g.n_devices == min(len(devs), 5) and g1.n_devices == 1
        t = g.raytrace(source, rcv, slowness=s)
        np.testing.assert_array_equal(t, t1)
        # slots are global: field and iteration count of every slot equal the single-device grid's
        for slot in range(5):
            np.testing.assert_array_equal(g.get_grid_traveltimes(slot), g1.get_grid_traveltimes(slot))
            assert g.get_niter(slot) == g1.get_niter(slot)
        # a single source in a named slot (thread_no), routed to the replica that owns the slot
        ts = g.raytrace(srcs[:1], rcv1, thread_no=4)
        np.testing.assert_array_equal(ts, t1[:rcv1.shape[0]])
        np.testing.assert_array_equal(g.get_grid_traveltimes(4), oracle.solve3d(dt, (n - 1,) * 3, dx, (0, 0, 0), s.flatten("F"), [srcs[0]])["tt"].reshape((n, n, n), order="F"))
        tm = g.timing()
        assert tm["n_sources"] == 1 and tm["node_updates"] > 0


def test_multi_device_rays_weno_and_threads():
    """raypaths in call order across replicas, the default ttcrpy configuration (cells, WENO, traveltimes from raypaths), and
    host threads calling single-source raytrace on a multi-device handle (the request combiner hands the batch to the
    replicas by slot)"""
    import ttcr_amd

    n = 25
    x = np.arange(n) * 1.0
    rng = np.random.default_rng(8)
    s = rng.uniform(0.4, 1.0, (n - 1, n - 1, n - 1))
    srcs = rng.uniform(3.0, 20.0, (5, 3))
    rcv1 = rng.uniform(2.0, 21.0, (4, 3))
    source = np.repeat(srcs, rcv1.shape[0], axis=0)
    rcv = np.tile(rcv1, (srcs.shape[0], 1))
    g1 = ttcr_amd.Grid3d(x, x, x, n_threads=4, cell_slowness=1, method="FSM")
    gm = ttcr_amd.Grid3d(x, x, x, n_threads=4, cell_slowness=1, method="FSM", device=[0, 0])
    t1, r1 = g1.raytrace(source, rcv, slowness=s, return_rays=True)
    tm, rm = gm.raytrace(source, rcv, slowness=s, return_rays=True)
    np.testing.assert_array_equal(tm, t1)
    assert len(rm) == len(r1)
    for a, b in zip(rm, r1):
        np.testing.assert_array_equal(a, b)
    # host threads, one slot each
    out = [None] * 4
    def work(k):
        out[k] = gm.raytrace(srcs[k:k + 1], rcv1, thread_no=k)
    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(4):
        np.testing.assert_array_equal(out[k], t1[k * 4:(k + 1) * 4])


def test_multi_device_2d_and_env(monkeypatch):
    import ttcr_amd

    nx, nz = 70, 90
    x, z = np.arange(nx) * 0.5, np.arange(nz) * 0.5
    rng = np.random.default_rng(2)
    s = rng.uniform(0.3, 1.0, (nx, nz))
    srcs = np.column_stack([rng.uniform(1, 33, 6), rng.uniform(1, 43, 6)])
    rcv1 = np.column_stack([rng.uniform(1, 33, 5), rng.uniform(1, 43, 5)])
    source = np.repeat(srcs, 5, axis=0)
    rcv = np.tile(rcv1, (6, 1))
    g1 = ttcr_amd.Grid2d(x, z, n_threads=3, cell_slowness=0, method="FSM", dtype=np.float32)
    t1 = g1.raytrace(source, rcv, slowness=s)
    gm = ttcr_amd.Grid2d(x, z, n_threads=3, cell_slowness=0, method="FSM", dtype=np.float32, device=[0, 0])
    assert gm.n_devices == 2
    np.testing.assert_array_equal(gm.raytrace(source, rcv, slowness=s), t1)
    # an unmodified caller (device = -1) is spread over the devices TTCR_AMD_DEVICES lists
    monkeypatch.setenv("TTCR_AMD_DEVICES", "0,0")
    ge = ttcr_amd.Grid2d(x, z, n_threads=3, cell_slowness=0, method="FSM", dtype=np.float32)
    assert ge.n_devices == 2
    np.testing.assert_array_equal(ge.raytrace(source, rcv, slowness=s), t1)
    monkeypatch.setenv("TTCR_AMD_DEVICES", "0,x")
    with pytest.raises(ValueError):
        ttcr_amd.Grid2d(x, z, n_threads=3, cell_slowness=0, method="FSM", dtype=np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [1, 0])
def test_paired_by_distance_sources_are_found_under_their_thread_numbers(oracle, pair, monkeypatch):
    """The batch driver pairs the sources of a call by distance (first-order 3-D grids): whatever slot storage a source ends
    up in, its field, iteration count and change history are found under the thread number the block distribution gives it."""
    import ttcr_amd
    monkeypatch.setenv("TTCR_FSM_PAIR", "1")   # (pairs are the layout of big batches: asked for on this small grid)
    n = 33
    x = np.arange(n) * 0.5
    rng = np.random.default_rng(5)
    s = rng.uniform(0.4, 1.0, (n, n, n)).astype(np.float32)
    S = 7   # odd: one source stays without a partner
    src = np.column_stack([np.zeros(S), rng.uniform(1.0, 15.0, (S, 3))])
    src[3, 1:] = src[0, 1:] + 0.3   # 0 and 3 are each other's nearest neighbours: they share a pair when pairing is on
    rcv = rng.uniform(0.5, 15.5, (S, 3))
    g = ttcr_amd.Grid3d(x, x, x, n_threads=S, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
    g.set_slowness(s)
    g.set_option('pair_sources', pair)
    tt = g.raytrace(src, rcv)
    for k in range(S):
        o = oracle.solve3d(np.float32, (n - 1,) * 3, 0.5, (0, 0, 0), np.asfortranarray(s).ravel(order='F'), src[k:k + 1, 1:], rcv=rcv[k:k + 1])
        np.testing.assert_array_equal(g.get_grid_traveltimes(k).ravel(order='F'), o['tt'])
        assert g.get_niter(k) == o['niter']
        assert tt[k] == o['tt_rcv'][0]
    # a single-source call on a thread number afterwards lands in that thread's storage, wherever it is now
    tt1 = g.raytrace(src[2:3], rcv[2:3], thread_no=5)
    o = oracle.solve3d(np.float32, (n - 1,) * 3, 0.5, (0, 0, 0), np.asfortranarray(s).ravel(order='F'), src[2:3, 1:], rcv=rcv[2:3])
    np.testing.assert_array_equal(g.get_grid_traveltimes(5).ravel(order='F'), o['tt'])
    assert tt1[0] == o['tt_rcv'][0]
    # ... and the other threads still hold what they held
    o4 = oracle.solve3d(np.float32, (n - 1,) * 3, 0.5, (0, 0, 0), np.asfortranarray(s).ravel(order='F'), src[4:5, 1:])
    np.testing.assert_array_equal(g.get_grid_traveltimes(4).ravel(order='F'), o4['tt'])


def test_slot_map_fuzz_short():
    """scripts/fuzz_pairing.py for a few seconds: random grids, slot and source counts (several rounds), max_batch, single-source
    calls on thread numbers in between -- pairing by distance on against off"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_pairing.py"), "8", "17"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz_pairing:" in r.stdout and " 0 random" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
