"""The tolerance-grade local solvers (update3_fast / update2_fast, ttcr_amd/csrc/fsm_kernels.h, option "arith" = 1) restated in numpy float32,
operation by operation, against the reference's quadratics evaluated in float64 (ttcr/Grid3Drn.h:2936-2956, ttcr/Grid2Drn.h:945-950): the
error of ONE update is a few ulp of the increment t - a1 plus the final rounding -- what tests/test_arith_mode_gpu.py then sees accumulate to
~1e-6 s RMS over a field.  No device needed: this pins the formulas, the GPU tests pin the kernels."""
import numpy as np

f32 = np.float32


def update3_fast(ax, ay, az, s, dx):
    a = np.sort(np.stack([ax, ay, az]), axis=0)
    a1, a2, a3 = a[0], a[1], a[2]
    fh = (s * dx).astype(f32)
    rfh = (f32(1) / fh).astype(f32)
    p2 = ((a2 - a1) * rfh).astype(f32); p3 = ((a3 - a1) * rfh).astype(f32)
    e = (p3 - p2).astype(f32)
    q = (p3.astype(np.float64) * p3 + (e * e).astype(f32)).astype(f32)               # fma(p3, p3, e*e)
    n2 = (f32(2) - p2.astype(np.float64) * p2).astype(f32)                          # fma(-p2, p2, 2)
    s3 = q < 1
    disc = np.where(s3, ((n2 + f32(1)).astype(f32) - q).astype(f32), n2)
    root = np.sqrt(np.maximum(disc, 0).astype(f32)).astype(f32)
    psum = np.where(s3, (p2 + p3).astype(f32), p2)
    w = (fh * np.where(s3, f32(1.0 / 3.0), f32(0.5))).astype(f32)
    t = (w.astype(np.float64) * (psum + root).astype(f32) + a1).astype(f32)           # fma(w, psum + root, a1)
    return np.where(p2 < 1, t, (a1 + fh).astype(f32))


def update3_ref64(ax, ay, az, s, dx):
    a = np.sort(np.stack([ax, ay, az]).astype(np.float64), axis=0)
    a1, a2, a3 = a
    fh = s.astype(np.float64) * dx
    t1 = a1 + fh
    t2 = 0.5 * (a1 + a2 + np.sqrt(np.maximum(2 * fh * fh - (a1 - a2) ** 2, 0)))
    t3 = (a1 + a2 + a3 + np.sqrt(np.maximum(3 * fh * fh - (a1 - a2) ** 2 - (a1 - a3) ** 2 - (a2 - a3) ** 2, 0))) / 3
    return np.where(t1 > a2, np.where(t2 > a3, t3, t2), t1), fh


def update2_fast(a, b, s, dx):
    fh = (s * dx).astype(f32)
    d = (a - b).astype(f32)
    t1 = (np.minimum(a, b) + fh).astype(f32)
    disc = ((f32(2) * fh).astype(np.float64) * fh - (d * d).astype(f32)).astype(f32)   # fma(2 fh, fh, -(d*d))
    root = np.sqrt(np.maximum(disc, 0)).astype(f32)
    t2 = (f32(0.5) * ((a + b).astype(f32) + root).astype(f32)).astype(f32)
    return np.where(np.abs(d) >= fh, t1, t2)


def test_update3_fast_is_a_few_ulp_of_the_increment_off():
    rng = np.random.default_rng(1)
    n = 400000
    base = rng.uniform(0.0, 30.0, n).astype(f32)
    s = rng.uniform(0.2, 1.0, n).astype(f32)
    dx = f32(0.0391)
    fh = s * dx
    # neighbours within a few fh of each other (the 1-D, 2-D and 3-D branches all occur), some far apart / unreached
    ax = base
    ay = (base + rng.uniform(0, 2.5, n).astype(f32) * fh).astype(f32)
    az = (base + rng.uniform(0, 2.5, n).astype(f32) * fh).astype(f32)
    az[::97] = np.finfo(f32).max
    with np.errstate(over="ignore", invalid="ignore"):     # the unreached neighbour: p3 = inf, the 2-D branch is taken as on the device
        got = update3_fast(ax, ay, az, s, dx).astype(np.float64)
        ref, fh64 = update3_ref64(ax, ay, az, s, dx)
    err = np.abs(got - ref)
    ulp_t = np.spacing(np.abs(ref).astype(f32)).astype(np.float64)
    # one rounding of the result (half an ulp of t) + the increment's error (a few 1e-7 of fh)
    assert np.all(err <= 0.5 * ulp_t + 8e-7 * fh64 + 1e-30), float(np.max((err - 0.5 * ulp_t) / fh64))
    # the three branches are all exercised
    a = np.sort(np.stack([ax, ay, az]).astype(np.float64), axis=0)
    t1 = a[0] + fh64
    assert np.mean(t1 <= a[1]) > 0.05 and np.mean(ref > a[2]) > 0.05 and np.mean((t1 > a[1]) & (ref <= a[2])) > 0.05


def test_update3_fast_is_scale_invariant_and_handles_fh_zero():
    rng = np.random.default_rng(2)
    n = 1000
    base = rng.uniform(1.0, 2.0, n).astype(f32)
    s = rng.uniform(0.2, 1.0, n).astype(f32)
    for scale in (f32(2.0 ** -40), f32(2.0 ** 30)):          # powers of two: every operation scales exactly
        dx = f32(0.125)
        fh = s * dx
        args = [base, (base + f32(0.3) * fh).astype(f32), (base + f32(0.6) * fh).astype(f32)]
        t = update3_fast(*args, s, dx)
        ts = update3_fast(*[(a * scale).astype(f32) for a in args], (s * scale).astype(f32), dx)
        assert np.array_equal((t * scale).astype(f32), ts)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = update3_fast(base, base, base, np.zeros(n, dtype=f32), f32(0.125))
    assert np.array_equal(t0, base)                           # zero slowness: the 1-D value a1 + 0, as in the reference


def test_update2_fast_matches_the_reference_up_to_the_root():
    rng = np.random.default_rng(3)
    n = 400000
    a = rng.uniform(0.0, 30.0, n).astype(f32)
    s = rng.uniform(0.2, 1.0, n).astype(f32)
    dx = f32(0.0049)
    fh = (s * dx).astype(f32)
    b = (a + rng.uniform(-1.5, 1.5, n).astype(f32) * fh).astype(f32)
    got = update2_fast(a, b, s, dx)
    # the reference's float evaluation (update2_fh in fsm_kernels.h): float d, float d*d, float a+b, double discriminant and root, one rounding
    d = (a - b).astype(f32)
    d2 = (d * d).astype(f32)
    disc = 2.0 * fh.astype(np.float64) ** 2 - d2.astype(np.float64)
    t2 = (0.5 * ((a + b).astype(f32).astype(np.float64) + np.sqrt(np.maximum(disc, 0)))).astype(f32)
    ref = np.where(np.abs(d) >= fh, (np.minimum(a, b) + fh).astype(f32), t2)
    same = np.mean(got == ref)
    assert same > 0.995, same                                  # identical except where the fp32 root moves the final rounding
    assert np.max(np.abs(got.astype(np.float64) - ref)) <= np.max(np.spacing(np.abs(ref)))
