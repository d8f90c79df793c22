"""The reference's stopping rule AS IT STANDS (ttcr/Grid3Drnfs.h:141-152: change = sequential T1 sum of abs(times[n] - T[n]),
continue while change >= eps * N): a fixture in which that sum and the fp64 sum of decreases fall on different sides of the
threshold -- the threshold is placed between the two sums of one iteration -- and the HIP path must report the iteration count
of the reference (restatement, pinned to the compiled reference by tests/test_oracle_vs_reference.py), not the fp64 rule's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(n, seed):
    rng = np.random.default_rng(seed)
    nb = (n + 7) // 8
    b = rng.uniform(0.25, 1.0, (nb, nb, nb))
    return np.repeat(np.repeat(np.repeat(b, 8, 0), 8, 1), 8, 2)[:n, :n, :n].copy()


@pytest.mark.parametrize("n,seed", [(128, 3), (112, 8)])
def test_borderline_iteration_follows_the_reference(oracle, n, seed):
    import ttcr_amd

    dt = np.float32
    dx = 0.25
    x = np.arange(n) * dx
    s = _model(n, seed)
    src = np.array([[7.3, 11.1, 4.9]]) * (n / 128.0)
    rcv = np.array([[0.0, 0.0, 0.0]])
    sF = s.flatten("F")

    def hip(eps, rule, fixed=0):
        g = ttcr_amd.Grid3d(x, x, x, n_threads=1, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, eps=eps, dtype=dt)
        g.set_slowness(s)
        g.set_option("stopping_rule", rule)
        if fixed:
            g.set_option("fixed_iters", fixed)
        g.raytrace(src, rcv)
        return g.get_niter(), g.get_changes()[0], g.get_grid_traveltimes().flatten("F")

    o = oracle.solve3d(dt, (n - 1,) * 3, dx, (0, 0, 0), sF, src, eps=1e-5)
    c_ref = np.asarray(o["change"], dtype=np.float64)             # the reference's sums, iteration by iteration
    _, c64, _ = hip(1e-5, 0, fixed=o["niter"])                     # the fp64 sums of the same iterations
    N = float(n ** 3)
    # an iteration whose two sums differ: put eps * N between them.  (Not the first, whose change is infinite, and not the second:
    # the snapshot the reference's sum needs is only taken once an iteration has announced that the next may be the last,
    # include/ttcr_amd.h "stopping_rule".)
    cand = [(abs(np.log(c_ref[k] / c64[k])), k) for k in range(2, o["niter"]) if np.isfinite(c_ref[k]) and c_ref[k] > 0 and c64[k] > 0
            and abs(c_ref[k] / c64[k] - 1.0) > 2e-5]
    assert cand, (c_ref, c64)
    _, k = max(cand)
    eps = float(np.sqrt(c_ref[k] * c64[k]) / N)
    thr = np.float32(eps) * np.float32(N)                          # epsilon *= N in T1 (ttcr/Grid3Drnfs.h:49)
    assert min(c_ref[k], c64[k]) < thr < max(c_ref[k], c64[k]), (c_ref[k], c64[k], thr)
    o2 = oracle.solve3d(dt, (n - 1,) * 3, dx, (0, 0, 0), sF, src, eps=eps)
    n_rule_ref, _, f_ref = hip(eps, 1)
    n_rule_64, _, _ = hip(eps, 0)
    print(f"n={n}: iteration {k + 1}: reference sum {c_ref[k]:.6f}, fp64 sum {c64[k]:.6f}, eps*N {float(thr):.6f}; niter reference {o2['niter']}, "
          f"HIP stopping_rule=reference {n_rule_ref}, stopping_rule=fp64 {n_rule_64}")
    assert n_rule_ref == o2["niter"]
    assert np.array_equal(f_ref, o2["tt"])
    assert n_rule_64 != o2["niter"], "the fixture is not borderline: both rules agree"
