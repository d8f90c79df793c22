"""The C-ABI library loads without a GPU and exports every symbol include/ttcr_amd.h declares;
calls that need a device fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ttcr_amd import build, _lib

    build.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ttcr_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ttcr_fsm\w*)\s*\(", text)))


def test_header_and_binding_agree(lib):
    from ttcr_amd import _lib

    names = declared_symbols()
    assert len(names) >= 19
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    for n in names:
        assert getattr(lib, n) is not None


def test_no_device_no_fallback(lib):
    """On a box without a GPU the create call must fail with a device error, not compute on CPU."""
    if lib.ttcr_fsm_device_count() > 0:
        pytest.skip("a GPU is present")
    from ttcr_amd import _lib

    h = C.c_void_p()
    st = lib.ttcr_fsm3d_create(C.byref(h), 0, 0, 4, 4, 4, 1.0, 0.0, 0.0, 0.0, 1e-5, 50, 0, 1, 0, -1)
    assert st == _lib.ERR_DEVICE
    assert "no HIP device" in _lib.last_error()
    assert not h.value
    import ttcr_amd

    x = np.arange(5.0)
    with pytest.raises(_lib.DeviceError):
        ttcr_amd.Grid3d(x, x, x, method="FSM", tt_from_rp=0, weno=0)


def test_argument_validation_happens_before_the_device(lib):
    from ttcr_amd import _lib

    h = C.c_void_p()
    assert lib.ttcr_fsm3d_create(C.byref(h), 7, 0, 4, 4, 4, 1.0, 0., 0., 0., 1e-5, 50, 0, 1, 0, -1) == _lib.ERR_VALUE
    assert lib.ttcr_fsm3d_create(C.byref(h), 0, 0, 0, 4, 4, 1.0, 0., 0., 0., 1e-5, 50, 0, 1, 0, -1) == _lib.ERR_VALUE
    # the WENO stencil needs >= 3 cells per axis
    assert lib.ttcr_fsm3d_create(C.byref(h), 0, 0, 4, 2, 4, 1.0, 0., 0., 0., 1e-5, 50, 1, 1, 0, -1) == _lib.ERR_VALUE
    assert "weno" in _lib.last_error()
    assert lib.ttcr_fsm2d_create(C.byref(h), 0, 0, 4, 2, 1.0, 1.0, 0., 0., 1e-5, 50, 1, 0, 1, -1) == _lib.ERR_VALUE
    assert lib.ttcr_fsm2d_create(C.byref(h), 0, 0, 4, 4, 1.0, -1.0, 0., 0., 1e-5, 50, 0, 1, 1, -1) == _lib.ERR_VALUE


def test_product_does_not_import_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(ROOT, "ttcr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
                assert "fsm_oracle" not in src, f
