"""Shared by tests/test_accuracy_study_gpu.py and tests/golden/make_accuracy_study.py: the models, sources, receivers and
error measures of the reference's accuracy study (tests/accuracy_grid3d.cpp: study 1 :203-252 with get_rel_error of
tests/test_grid3d.cpp:67-96, study 2 :258-345), as data and formulas -- a = 1, b = 0.1, V0 = 3
(tests/files/mk_models3d.py:14-17,75-149, mk_constant_models.py)."""
import numpy as np

A, B, V0 = 1.0, 0.1, 3.0
# precision, model, resolution -> error column of tests/accuracy_grid3d.csv, method FAST_SWEEPING (six digits as printed)
PUBLISHED = {
    ("double", "layers", "medium"): 0.00669965, ("double", "layers", "fine"): 0.00373589,
    ("double", "gradient", "medium"): 0.00228619, ("double", "gradient", "fine"): 0.00279997,
    ("float", "layers", "medium"): 0.00670199, ("float", "layers", "fine"): 0.003749,
    ("float", "gradient", "medium"): 0.00228538, ("float", "gradient", "fine"): 0.00281546,
    ("double", "constant", "medium"): 0.00152022, ("double", "constant", "fine"): 0.0017866,
    ("float", "constant", "medium"): 0.00151924, ("float", "constant", "fine"): 0.00177034,
}
FRAC = {"medium": 2, "fine": 8}
DTYPE = {"double": np.float64, "float": np.float32}


def model(name, resolution):
    """(node coordinates of an axis, slowness as an (n,n,n) array of nodes or cells, cell_slowness)"""
    frac = FRAC[resolution]
    n = 20 * frac
    x = np.arange(n + 1) / frac                       # n * dx with dx = 1 / frac, exact
    if name == "gradient":                            # node slowness 1 / (a + b z)
        s = np.broadcast_to((1.0 / (A + B * x))[None, None, :], (n + 1,) * 3)
        return x, np.ascontiguousarray(s), 0
    if name == "layers":                              # cell slowness 1 / (a + b (floor(z_lo) + 0.5))
        s = np.broadcast_to((1.0 / (A + B * (np.floor(x[:-1]) + 0.5)))[None, None, :], (n,) * 3)
        return x, np.ascontiguousarray(s), 1
    return x, np.full((n + 1,) * 3, 1.0 / V0), 0


def six_digits(value, published):
    """the CSV prints six significant digits"""
    return abs(value - published) <= 0.51 * 10.0 ** (np.floor(np.log10(published)) - 5)


def constant_error(dt, srcs, rcv, tt):
    """mean relative misfit against s0 * distance over all source-receiver pairs (accuracy_grid3d.cpp:308-324): the
    receivers as T, the sources as the doubles they were drawn as, s0 = the slowness as the grid holds it"""
    tt = np.asarray(tt, dtype=np.float64).reshape(len(srcs), rcv.shape[0])
    s0 = float(dt(1.0 / V0))
    rx = rcv.astype(dt).astype(np.float64)
    ref = s0 * np.sqrt(((rx[None, :, :] - srcs[:, None, :]) ** 2).sum(-1))
    ok = ref != 0.0
    return float(np.mean(np.abs((ref[ok] - tt[ok]) / ref[ok])))
