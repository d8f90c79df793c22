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


def _seq_sum(x):
    s = x.dtype.type(0)
    for v in x:
        s = x.dtype.type(s + v)
    return s


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_parallel_form_of_the_sequential_sum(dt):
    """fsm_refsum_* (the reference's `change` in parallel) against the one-chain kernel and, on a small field, a Python loop: fields
    with ties at every scale (powers of two, multiples of half an ulp of the running sum), sparse changes, changes that take the
    sum through many binades -- the two device kernels must agree to the bit."""
    import ttcr_amd

    rng = np.random.default_rng(11)
    for n, reps in ((24, 6), (160, 5)):
        x = np.arange(n) * 0.5
        g = ttcr_amd.Grid3d(x, x, x, n_threads=1, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=dt)
        N = n ** 3
        for rep in range(reps):
            kind = rep % 6
            if kind == 0: d = rng.uniform(0, 1e-3, N)
            elif kind == 1: d = rng.uniform(0, 1, N) * (rng.uniform(0, 1, N) < 0.01)
            elif kind == 2: d = 2.0 ** rng.integers(-40, 3, N).astype(np.float64)
            elif kind == 3: d = rng.integers(0, 4, N) * 2.0 ** (-24 if dt == np.float32 else -53) + (rng.uniform(0, 1, N) < 1e-4) * 1.0
            elif kind == 4: d = np.abs(rng.normal(0, 1, N)) * 10.0 ** rng.integers(-12, 3, N)
            else: d = np.full(N, 2.0 ** -20)
            d = d.astype(dt)
            base = rng.uniform(1.0, 2.0, N).astype(dt)
            old = (base + d).astype(dt)           # times[n]; the kernels take abs(times - field) in T1 themselves
            a = g.reference_change(old, base, parallel=True)
            b = g.reference_change(old, base, parallel=False)
            assert a == b, (n, rep, kind, a, b)
            if n <= 24:
                dd = np.abs(old - base).astype(dt)
                assert a == _seq_sum(dd), (n, rep, kind, a, _seq_sum(dd))
        print(n, dt.__name__, g.stopping_stats())


def _solve_all(n, n_src, eps, opts, s, dx=0.25):
    import ttcr_amd

    x = np.arange(n) * dx
    g = ttcr_amd.Grid3d(x, x, x, n_threads=n_src, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, eps=eps, dtype=np.float32)
    g.set_slowness(s)
    for k, v in opts.items():
        g.set_option(k, v)
    rng = np.random.default_rng(4)
    src = rng.uniform(0.1, 0.9, (n_src, 3)) * (n - 1) * dx
    rcv = np.array([[0.0, 0.0, 0.0]])
    g.raytrace(np.repeat(src, 1, axis=0), np.tile(rcv, (n_src, 1)))
    out = []
    for i in range(n_src):
        out.append((g.get_niter(i), g.get_reference_changes(i)[0], g.get_grid_traveltimes(i).flatten("F")))
    return out, g.stopping_stats(), src, g.last_kernel()


def _same_decisions(got, ref, thr):
    """Sums handed out by get_reference_changes against the full sums of the same iterations: the NaN pattern is the same, a sum below
    the threshold is the whole sum bit for bit, one that reached it may have been cut short there (the sum only grows)."""
    m = ~np.isnan(got)
    if not np.array_equal(m, ~np.isnan(ref)):
        return False
    below = m & (ref < thr)
    return bool(np.array_equal(got[below], ref[below]) and np.all(got[m & ~below] >= thr))


@pytest.mark.parametrize("layout", [1, 0])
def test_compacted_terms_and_brick_passes_give_the_reference_sum(oracle, layout, monkeypatch):
    """The default form of the rule -- non-zero terms compacted in node order, the passes over field and snapshot restricted to the
    bricks the skipping sweep kernels stamped, snapshots brought up to date brick by brick -- against (i) the one-chain sum over whole
    strided fields with whole-field snapshots (stopping_rule = 2, stopping_shortcuts = 0) and (ii) the reference's own `change` of the
    same iterations (restatement pinned to the compiled reference): the sums that decided are equal bit for bit, so are the
    iteration counts.  Eight sources, both field layouts, exact skipping forced on (it is what keeps the stamps)."""
    n, n_src, eps = 96, 8, 1e-5
    thr = float(np.float32(eps) * np.float32(n ** 3))            # epsilon *= N in T1 (ttcr/Grid3Drnfs.h:49)
    s = _model(n, 21)
    monkeypatch.setenv("TTCR_FSM_PAIR", str(layout))             # (read when a grid is made: two fields per workgroup / one)
    a, st_a, src, kern = _solve_all(n, n_src, eps, {"skip": 1, "stopping_shortcuts": 1}, s)
    b, st_b, _, _ = _solve_all(n, n_src, eps, {"skip": 1, "stopping_rule": 2, "stopping_shortcuts": 0}, s)
    d, st_d, _, _ = _solve_all(n, n_src, eps, {"skip": 1}, s)    # the default: sums that bounds already decide are not computed
    print(kern, st_a, st_b, st_d, [q[0] for q in a])
    assert kern.split(",")[7] == ("2" if layout else "1") and kern.split(",")[5] == "true", kern
    assert st_d["reference_sums"] == st_b["reference_sums"] and st_d["reference_sums_missed"] == 0 and st_d["rounds"] <= st_a["rounds"]
    for i in range(n_src):
        assert d[i][0] == b[i][0] and np.array_equal(d[i][2], b[i][2])
        m = ~np.isnan(d[i][1])
        assert np.all(~np.isnan(b[i][1][m])) and _same_decisions(d[i][1][m], b[i][1][m], thr), (i, d[i][1], b[i][1])
    print("sums computed in full by default:", sum(int(np.sum(~np.isnan(q[1]))) for q in d), "of", st_d["reference_sums"])
    assert st_a["reference_sums"] == st_b["reference_sums"] > 0 and st_a["reference_sums_missed"] == st_b["reference_sums_missed"] == 0
    asked = 0
    for i in range(n_src):
        assert a[i][0] == b[i][0]
        assert _same_decisions(a[i][1], b[i][1], thr), (i, a[i][1], b[i][1])
        assert np.array_equal(a[i][2], b[i][2])
        asked += int(np.sum(~np.isnan(a[i][1])))
    assert asked == st_a["reference_sums"]
    sF = s.flatten("F")
    whole = 0
    for i in (0, 5):
        o = oracle.solve3d(np.float32, (n - 1,) * 3, 0.25, (0, 0, 0), sF, src[i:i + 1], eps=eps)
        assert a[i][0] == o["niter"]
        ref = np.asarray(o["change"], dtype=np.float64)
        got = a[i][1]
        m = ~np.isnan(got)
        full = np.where(m, ref[:got.size], np.nan)
        assert m.any() and _same_decisions(got, full, thr), (i, got, ref)
        whole += int(np.sum(m & (full < thr)))
        assert np.array_equal(a[i][2], o["tt"])
    assert whole > 0, "none of the sums compared with the reference's was a whole one"


def test_brick_passes_beyond_the_always_snapshot_size(monkeypatch):
    """The same comparison (i) on a grid above 2^24 values per slot group, where snapshots are taken on prediction only: the first one of
    a group is a whole copy, the later ones and the passes of the sums go by the stamps."""
    n, n_src, eps = 208, 2, 1e-5
    thr = float(np.float32(eps) * np.float32(n ** 3))
    s = _model(n, 22)
    monkeypatch.setenv("TTCR_FSM_PAIR", "1")
    a, st_a, _, kern = _solve_all(n, n_src, eps, {"skip": 1, "stopping_shortcuts": 1}, s)
    b, st_b, _, _ = _solve_all(n, n_src, eps, {"skip": 1, "stopping_rule": 2, "stopping_shortcuts": 0}, s)
    d, st_d, _, _ = _solve_all(n, n_src, eps, {"skip": 1}, s)
    print(kern, st_a, st_b, st_d, [q[0] for q in a])
    assert kern.split(",")[7] == "2" and kern.split(",")[5] == "true", kern
    assert [q[0] for q in d] == [q[0] for q in b] and all(np.array_equal(d[i][2], b[i][2]) for i in range(n_src))
    assert st_a["reference_sums"] == st_b["reference_sums"] > 0 and st_a["reference_sums_missed"] == st_b["reference_sums_missed"] == 0
    for i in range(n_src):
        assert a[i][0] == b[i][0]
        assert _same_decisions(a[i][1], b[i][1], thr), (i, a[i][1], b[i][1])
        assert np.array_equal(a[i][2], b[i][2])


@pytest.mark.parametrize("fields", [0, 2])
def test_small_batches_and_the_strided_fallback_decide_alike(fields, monkeypatch):
    """Where the device has no room for the compact term arrays of 16 fields the batches shrink, and with no room for a pair the sums go by
    the strided fields (TTCR_FSM_RS_FIELDS, read when a grid is made, forces either): same sums, same iteration counts, same fields."""
    n, n_src, eps = 96, 8, 1e-5
    thr = float(np.float32(eps) * np.float32(n ** 3))
    s = _model(n, 23)
    monkeypatch.setenv("TTCR_FSM_PAIR", "1")
    b, st_b, _, _ = _solve_all(n, n_src, eps, {"skip": 1, "stopping_shortcuts": 1}, s)
    monkeypatch.setenv("TTCR_FSM_RS_FIELDS", str(fields))
    a, st_a, _, kern = _solve_all(n, n_src, eps, {"skip": 1, "stopping_shortcuts": 1}, s)
    print(kern, st_a, st_b)
    assert st_a["reference_sums"] == st_b["reference_sums"] > 0 and st_a["reference_sums_missed"] == 0
    for i in range(n_src):
        assert a[i][0] == b[i][0] and np.array_equal(a[i][2], b[i][2])
        m = ~np.isnan(b[i][1])
        assert np.array_equal(m, ~np.isnan(a[i][1]))
        # (cut-short sums may stop at different places: whole sums are equal, the others lie on the same side)
        whole = m & (b[i][1] < thr)
        assert np.array_equal(a[i][1][whole], b[i][1][whole]) and np.all(a[i][1][m & ~whole] >= thr)
