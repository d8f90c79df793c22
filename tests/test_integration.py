"""The reference-side binding, compiled: integration/Grid3Drnfs_amd.h / Grid2Drnfs_amd.h (adapters deriving from the
reference's Grid3D / Grid2D) and integration/ttcr_amd.pxd (Cython declarations of the C ABI).

  not gpu: the adapters compile against the unmodified reference headers where they lie and override every virtual
           the Cython layer calls (static_asserts in integration/adapter_check.cpp) -- build container only; the Cython
           stub is cythonized + compiled into an extension module that talks to libttcr_amd.so.
  gpu:     integration/_build/adapter_check (built here, travels like oracle/_ref) drives the backend through
           Grid3D<T,uint32_t>* / Grid2D<T,uint32_t,sxz<T>>* on the GPU; its printed values are compared with the oracle."""
import importlib.util
import os
import subprocess
import sys
import sysconfig

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEG = os.path.join(ROOT, "integration")
EXE = os.path.join(INTEG, "_build", "adapter_check")
HAVE_REF = os.path.isdir("/root/reference/ttcr")
HAVE_LIB = os.path.exists(os.path.join(ROOT, "ttcr_amd", "libttcr_amd.so"))


@pytest.mark.skipif(not HAVE_REF, reason="reference headers absent (GPU box)")
def test_adapters_compile_against_the_reference_and_override_what_cython_calls():
    subprocess.check_call(["make", "-s", "-C", INTEG, "syntax"])


@pytest.mark.skipif(not (HAVE_REF and HAVE_LIB), reason="needs the reference headers and libttcr_amd.so")
def test_adapter_check_program_links():
    subprocess.check_call(["make", "-s", "-C", INTEG])
    assert os.path.exists(EXE)


@pytest.mark.skipif(not HAVE_LIB, reason="libttcr_amd.so not built")
def test_cython_stub_compiles_and_calls_the_library(tmp_path):
    cpp = tmp_path / "abi_probe.cpp"
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-3", "-I", INTEG, os.path.join(INTEG, "abi_probe.pyx"), "-o", str(cpp)])
    so = tmp_path / ("abi_probe" + sysconfig.get_config_var("EXT_SUFFIX"))
    libdir = os.path.join(ROOT, "ttcr_amd")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-std=c++17", str(cpp), "-I", sysconfig.get_paths()["include"],
                           "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lttcr_amd", "-Wl,-rpath," + libdir, "-o", str(so)])
    spec = importlib.util.spec_from_file_location("abi_probe", str(so))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n_dev = mod.device_count()
    st, msg, nn = mod.create_and_destroy(4)
    if n_dev == 0:
        assert st == 3 and "no HIP device" in msg and nn == 0   # TTCR_ERR_DEVICE: no fallback
    else:
        assert st == 0 and nn == 125


@pytest.mark.skipif(not (HAVE_REF and HAVE_LIB), reason="needs the reference headers and libttcr_amd.so")
def test_cython_adapter_module_compiles_like_rgrid_pyx(tmp_path):
    """integration/adapter_probe.pyx: `new Grid3Drnfs_amd[double,uint32_t](...)` + setSlowness + raytrace with `except +`,
    compiled against the reference headers; without a GPU the constructor's failure arrives as RuntimeError."""
    cpp = tmp_path / "adapter_probe.cpp"
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-3", "-I", INTEG, os.path.join(INTEG, "adapter_probe.pyx"), "-o", str(cpp)])
    so = tmp_path / ("adapter_probe" + sysconfig.get_config_var("EXT_SUFFIX"))
    libdir = os.path.join(ROOT, "ttcr_amd")
    ref = "/root/reference"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-std=c++17", "-pthread", "-w", str(cpp), "-I", sysconfig.get_paths()["include"],
                           "-I", INTEG, "-I", os.path.join(ROOT, "include"), "-I", ref + "/ttcr", "-I", ref + "/boost_1_91_0",
                           "-I", ref + "/eigen-5.0.0", "-DTTCR_ADAPTER_PROBE", "-L", libdir, "-lttcr_amd", "-Wl,-rpath," + libdir, "-o", str(so)])
    spec = importlib.util.spec_from_file_location("adapter_probe", str(so))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import ttcr_amd._lib as L
    if L.load().ttcr_fsm_device_count() == 0:
        with pytest.raises(RuntimeError, match="no HIP device"):
            mod.solve(4, 1.0, [1.0] * 125, [0.0, 0.0, 0.0], [[4.0, 4.0, 4.0]])


def _slow(n):
    from test_capi_smoke_gpu import slow
    return slow(n)


@pytest.mark.gpu
def test_adapter_drives_the_gpu_through_the_reference_base_classes(oracle):
    if not os.path.exists(EXE):
        pytest.skip("integration/_build/adapter_check was not built (needs the reference headers at build time)")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    out = r.stdout
    assert r.returncode == 0 and "failures 0" in out and "FAIL" not in out, out + r.stderr
    vals = {}
    for line in out.splitlines():
        k, *rest = line.split()
        if k.startswith("a3") or k.startswith("a2"):
            vals[k] = [int(v) if "niter" in k else float.fromhex(v) for v in rest]
    # 3-D node grid fp32 (the model of tests/capi_smoke.c)
    nc, dx, org = (18, 14, 11), 0.5, (1.0, -2.0, 0.0)
    nn = 19 * 15 * 12
    from test_capi_smoke_gpu import smooth3
    s = smooth3(19, 15, 12)
    rx = np.array([[1.0, -2.0, 0.0], [10.0, 5.0, 5.5], [4.4, 0.3, 1.9]])
    src = [[3.3, 1.1, 2.7]]
    o = oracle.solve3d(np.float32, nc, dx, org, s, src, t0=[0.25], rcv=rx)
    np.testing.assert_array_equal(np.array(vals["a3_tt"], dtype=np.float32), o["tt_rcv"])
    assert vals["a3_niter"] == [o["niter"]]
    assert vals["a3_field_sum"][0] == float(np.sum(o["tt"].astype(np.float64)))
    np.testing.assert_array_equal(np.array(vals["a3_s0"], dtype=np.float32), oracle.compute_slowness3d(np.float32, nc, dx, org, s, src))
    orp = oracle.solve3d(np.float32, nc, dx, org, s, src, t0=[0.25], rcv=rx, return_rays=True)
    np.testing.assert_array_equal(np.array(vals["a3_tt_rays"], dtype=np.float32), orp["tt_rcv"])
    msrc = [[3.3, 1.1, 2.7], [8.0, 2.0, 4.0], [1.0, -2.0, 0.0], [5.5, 3.3, 1.1]]
    mt0 = [0.25, 0.0, 1.0, 0.0]
    mrx = [rx, [[2.0, 2.0, 2.0]], [[9.5, 4.5, 5.0], [3.0, 0.0, 1.0]], rx]
    want = np.concatenate([oracle.solve3d(np.float32, nc, dx, org, s, [msrc[n]], t0=[mt0[n]], rcv=mrx[n])["tt_rcv"] for n in range(4)])
    np.testing.assert_array_equal(np.array(vals["a3_multi"], dtype=np.float32), want)
    # 3-D cell grid fp64, translated origin, WENO, traveltimes from raypaths (the ttcrpy defaults)
    sc = _slow(1000 + np.arange(6 * 5 * 4)).astype(np.float64)
    orgc = (500000.0, 4000000.0, -1000.0)
    srcc = [[500002.5, 4000002.5, -999.0]]
    rxc = [[500001.0, 4000001.5, -998.5], [500005.0, 4000004.0, -997.0]]
    oc = oracle.solve3d(np.float64, (6, 5, 4), 1.0, orgc, sc, srcc, rcv=rxc, cell_slowness=True, translate=True, weno=True, tt_from_rp=True)
    np.testing.assert_array_equal(np.array(vals["a3c_tt"]), oc["tt_rcv"])
    assert vals["a3c_niter"] == [oc["niter"], oc["niterw"]]
    np.testing.assert_array_equal(np.array(vals["a3c_s0"]), oracle.compute_slowness3d(np.float64, (6, 5, 4), 1.0, orgc, sc, srcc, True, True))
    # 2-D
    s2 = _slow(5000 + np.arange(21 * 13))
    rx2 = [[0.0, 0.0], [10.0, 3.0]]
    o2 = oracle.solve2d(np.float32, (20, 12), 0.5, 0.25, (0, 0), s2, [[3.3, 1.1]], rcv=rx2)
    np.testing.assert_array_equal(np.array(vals["a2_tt"], dtype=np.float32), o2["tt_rcv"])
    np.testing.assert_array_equal(np.array(vals["a2_s0"], dtype=np.float32),
                                  oracle.compute_slowness2d(np.float32, (20, 12), 0.5, 0.25, (0, 0), s2, [[3.3, 1.1]]))
    m2 = [([[3.3, 1.1]], 0.0, rx2), ([[7.0, 2.0]], 0.5, [[5.0, 1.0]]), ([[0.0, 0.0]], 0.0, rx2)]
    want2 = np.concatenate([oracle.solve2d(np.float32, (20, 12), 0.5, 0.25, (0, 0), s2, a, t0=[b], rcv=c)["tt_rcv"] for a, b, c in m2])
    np.testing.assert_array_equal(np.array(vals["a2_multi"], dtype=np.float32), want2)


def test_reference_cython_wrapper_builds_against_the_backend(tmp_path):
    """INTEGRATION.md section 1 carried out: integration/patch_ttcrpy.sh applies the three edits to a scratch copy of the
    reference's own src/ttcrpy/rgrid.pyx / rgrid.pxd (sed commands only -- nothing of the reference lives in this
    repository), cythonizes it and compiles it against the adapters and libttcr_amd.so.  The module cannot be imported here
    (it imports vtk at module scope, which this image lacks -- SURVEY section 8c), so its symbol table is read instead: the
    init function is there, the backend's ABI is what it links against, and no OpenCL entry point is left."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "src", "ttcrpy")):
        pytest.skip("the reference tree is not present (GPU box)")
    from ttcr_amd import build as B

    if not os.path.exists(B.LIB):
        pytest.skip("libttcr_amd.so has not been built")
    out = tmp_path / "ttcrpy_patched"
    r = subprocess.run([os.path.join(ROOT, "integration", "patch_ttcrpy.sh"), ref, str(out), "-O0"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    so = r.stdout.strip().splitlines()[-1]
    assert os.path.exists(so) and str(out) in so
    # the edited sources name the adapter where the reference named its OpenCL classes
    pyx = open(out / "ttcrpy" / "rgrid.pyx").read()
    pxd = open(out / "ttcrpy" / "rgrid.pxd").read()
    assert pyx.count("new Grid3Drnfs_amd[double,uint32_t](") == 2 and pyx.count("new Grid2Drnfs_amd[") == 4
    assert "_OpenCL[" not in pyx and "_OpenCL[" not in pxd and 'cdef extern from "Grid3Drnfs_amd.h"' in pxd
    nm = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    undefined = {ln.split()[-1] for ln in nm.splitlines() if " U " in ln}
    assert "PyInit_rgrid" in defined
    for sym in ("ttcr_fsm3d_create", "ttcr_fsm2d_create", "ttcr_fsm_destroy", "ttcr_fsm_set_slowness", "ttcr_fsm_raytrace",
                "ttcr_fsm_raytrace_rays", "ttcr_fsm_get_slot_rays", "ttcr_fsm_get_tt",
                "ttcr_fsm_compute_slowness", "ttcr_fsm_get_niter", "ttcr_fsm_set_option", "ttcr_fsm_last_error"):
        assert sym in undefined, sym
    assert not [s for s in undefined if s.startswith("cl") and s[2:3].isupper()], "an OpenCL entry point is still linked"
    ldd = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "libttcr_amd.so" in ldd and "OpenCL" not in ldd
