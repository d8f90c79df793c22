"""The C ABI driven from plain C (tests/capi_smoke.c): no Python between the caller and libttcr_amd.so.
The program checks the status codes and the reference's error texts itself; the traveltimes it prints
(hex floats) are compared with the CPU oracle here."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
EXE = os.path.join(BUILD, "capi_smoke")
LIBDIR = os.path.join(ROOT, "ttcr_amd")


def build_smoke():
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "tests", "capi_smoke.c")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", src, "-I", os.path.join(ROOT, "include"), "-L", LIBDIR,
           "-lttcr_amd", "-Wl,-rpath," + LIBDIR, "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def slow(n):
    """tests/capi_smoke.c:slow() in uint32 arithmetic"""
    n = np.asarray(n, dtype=np.uint64)
    h = (n * 2654435761) & 0xffffffff
    h ^= h >> 15
    h = (h * 2246822519) & 0xffffffff
    h ^= h >> 13
    return (np.float32(0.3) + np.float32(0.7) * (h & 0xffff).astype(np.float32) / np.float32(65535.0)).astype(np.float32)


def smooth3(nnx, nny, nnz):
    """tests/capi_smoke.c:smooth3() for every node, flat x-fastest, float32 arithmetic in the same order"""
    n = np.arange(nnx * nny * nnz)
    i, j, k = n % nnx, (n // nnx) % nny, n // (nnx * nny)
    f = np.float32
    v = f(0.4) + f(0.02) * k.astype(f)
    v = v + f(0.01) * j.astype(f)
    v = v + f(0.005) * i.astype(f)
    return v.astype(f)


def test_capi_smoke_compiles_as_plain_c():
    """-m "not gpu": the header is valid C99 and every entry point the program uses links against the library"""
    if not os.path.exists(os.path.join(LIBDIR, "libttcr_amd.so")):
        pytest.skip("libttcr_amd.so not built")
    assert os.path.exists(build_smoke())


@pytest.mark.gpu
def test_capi_smoke_runs_and_matches_the_oracle(oracle):
    exe = build_smoke()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    out = r.stdout
    assert r.returncode == 0 and "failures 0" in out, out + r.stderr
    assert "FAIL" not in out
    vals = {}
    for line in out.splitlines():
        k, *rest = line.split()
        if k in ("tt3d", "ttmulti", "ttcells", "tt2d", "field3d_sum", "field3d_probe", "ttrays", "ttm", "ttrm", "change3d"):
            vals[k] = [float.fromhex(v) for v in rest]
        elif k in ("niter3d", "rays_npts", "m_shape"):
            vals[k] = [int(v) for v in rest]
        elif k == "m_sums":
            vals[k] = [float.fromhex(rest[0]), int(rest[1])]
    # 3-D node grid, fp32
    nc, dx, org = (18, 14, 11), 0.5, (1.0, -2.0, 0.0)
    nn = 19 * 15 * 12
    s = smooth3(19, 15, 12)
    rx = np.array([[1.0, -2.0, 0.0], [10.0, 5.0, 5.5], [4.4, 0.3, 1.9]])
    o = oracle.solve3d(np.float32, nc, dx, org, s, [[3.3, 1.1, 2.7]], t0=[0.25], rcv=rx)
    assert vals["niter3d"] == [o["niter"]]
    np.testing.assert_array_equal(np.array(vals["tt3d"], dtype=np.float32), o["tt_rcv"])
    assert vals["field3d_sum"][0] == float(np.sum(o["tt"].astype(np.float64)))
    np.testing.assert_array_equal(np.array(vals["field3d_probe"], dtype=np.float32), o["tt"][[0, nn // 2, nn - 1]])
    mtx = np.array([[3.3, 1.1, 2.7], [8.0, 2.0, 4.0], [2.0, 0.0, 1.5]])
    mt0 = [0.25, 0.0, 1.0]
    mrx = np.array([[1.0, -2.0, 0.0], [10.0, 5.0, 5.5], [4.4, 0.3, 1.9], [2.0, 2.0, 2.0], [9.5, 4.5, 5.0], [3.0, 0.0, 1.0]])
    off = [0, 3, 4, 6]
    want = np.concatenate([oracle.solve3d(np.float32, nc, dx, org, s, mtx[n:n + 1], t0=mt0[n:n + 1],
                                          rcv=mrx[off[n]:off[n + 1]])["tt_rcv"] for n in range(3)])
    np.testing.assert_array_equal(np.array(vals["ttmulti"], dtype=np.float32), want)
    # the r_data and m_data overloads (rays per slot, matrix M per slot)
    ry = rx[1:]
    orr = oracle.solve3d(np.float32, nc, dx, org, s, [[3.3, 1.1, 2.7]], t0=[0.25], rcv=ry, return_rays=True)
    np.testing.assert_array_equal(np.array(vals["ttrays"], dtype=np.float32), orr["tt_rcv"])
    assert vals["rays_npts"] == [len(r) for r in orr["rays"]]
    om = oracle.solve3d(np.float32, nc, dx, org, s, [[3.3, 1.1, 2.7]], t0=[0.25], rcv=ry, compute_m=True)
    np.testing.assert_array_equal(np.array(vals["ttm"], dtype=np.float32), om["tt_rcv"])
    assert vals["m_shape"] == [len(j) for j, _ in om["m"]]
    assert vals["m_sums"][1] == int(sum(int(np.sum(j)) for j, _ in om["m"]))
    assert vals["m_sums"][0] == float(sum(np.sum(v.astype(np.float64)) for _, v in om["m"]))
    orm = oracle.solve3d(np.float32, nc, dx, org, s, [[3.3, 1.1, 2.7]], t0=[0.25], rcv=ry, compute_m=True, return_rays=True)
    np.testing.assert_array_equal(np.array(vals["ttrm"], dtype=np.float32), orm["tt_rcv"])
    # the stopping rule's quantity: fp64 sum of the decreases vs the reference's sequential fp32 sum
    # (iteration 1 lowers every node from the initial "infinity": huge in both)
    assert vals["change3d"][0] > 1e30 and o["change"][0] > 1e30
    np.testing.assert_allclose(vals["change3d"][1], o["change"][-1], rtol=1e-3)
    # 3-D cell grid, fp64
    sc = slow(1000 + np.arange(6 * 5 * 4)).astype(np.float64)
    oc = oracle.solve3d(np.float64, (6, 5, 4), 1.0, (0, 0, 0), sc, [[2.5, 2.5, 1.0]], rcv=[[0, 0, 0], [6, 5, 4]], cell_slowness=True)
    np.testing.assert_array_equal(np.array(vals["ttcells"]), oc["tt_rcv"])
    # 2-D, dx != dz
    s2 = slow(5000 + np.arange(21 * 13))
    o2 = oracle.solve2d(np.float32, (20, 12), 0.5, 0.25, (0, 0), s2, [[3.3, 1.1]], rcv=[[0, 0], [10.0, 3.0]])
    np.testing.assert_array_equal(np.array(vals["tt2d"], dtype=np.float32), o2["tt_rcv"])
