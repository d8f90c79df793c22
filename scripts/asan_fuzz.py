"""Out-of-range audit of the oracle's index arithmetic under AddressSanitizer (CPU, build container).

  make -C oracle liboracle_asan.so
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 TTCR_ORACLE_LIB=oracle/liboracle_asan.so \
      python scripts/asan_fuzz.py <seconds> [seed]

Random small grids (2-D / 3-D, node / cell slowness, dx above 1 and non-dyadic, translated origins, WENO on/off) with
receivers and sources biased towards the faces, edges and corners of the grid and towards a few ulps inside the last
plane of an axis; every receiver goes through getTraveltime, getTraveltimeFromRaypath / getRaypath and computeSlowness.
ASan aborts the process at the first read or write outside an array; the configuration is printed before every solve."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n = walked = failed = 0


def edge_points(axes, dt, k):
    pts = []
    for _ in range(k):
        p = []
        for a in axes:
            r = rng.random()
            if r < 0.25:
                v = a[-1]
            elif r < 0.4:
                v = a[0]
            elif r < 0.7:   # a few ulps inside the last / first plane
                v = dt(a[-1] if rng.random() < 0.7 else a[0])
                for _ in range(int(rng.integers(1, 300))):
                    v = np.nextafter(v, dt(a[a.size // 2]))
                v = float(v)
            elif r < 0.8:
                v = a[int(rng.integers(0, a.size))]
            else:
                v = rng.uniform(a[0], a[-1])
            p.append(v)
        pts.append(p)
    return np.array(pts)


def smooth(axes):
    g = np.meshgrid(*axes, indexing='ij')
    return 1.0 / (1.0 + rng.uniform(0.02, 0.1) * (g[-1] - axes[-1][0]))


while time.time() < t_end:
    dim = 3 if rng.random() < 0.5 else 2
    dt = np.float32 if rng.random() < 0.5 else np.float64
    cell = bool(rng.random() < 0.4)
    weno = bool(rng.random() < 0.3)
    nc = tuple(int(v) for v in rng.integers(4 if weno else 1, 14 if dim == 3 else 30, dim))
    dx = float(rng.choice([0.25, 1.0, 2.3, 0.7, 17.0, 1e-3]))
    dz = dx if dim == 3 or rng.random() < 0.5 else float(rng.choice([0.125, 3.1, 0.75]))
    steps = (dx,) * 3 if dim == 3 else (dx, dz)
    org = tuple(float(rng.choice([0.0, -3.5, 1000.25, 500000.0])) for _ in range(dim))
    translate = dim == 3 and rng.random() < 0.3
    axes = [o + np.arange(m + 1) * h for o, m, h in zip(org, nc, steps)]
    caxes = [0.5 * (a[1:] + a[:-1]) for a in axes]
    s = smooth(caxes if cell else axes)
    src = edge_points(axes, dt, 1) if rng.random() < 0.5 else np.array([[rng.uniform(a[0], a[-1]) for a in axes]])
    if rng.random() < 0.4:   # one or two more points of the same source within a cell of the first: the end game runs once per point
        more = [np.clip(src[0] + rng.uniform(-0.7, 0.7, dim) * np.array(steps), [a[0] for a in axes], [a[-1] for a in axes])
                for _ in range(int(rng.integers(1, 3)))]
        src = np.vstack([src] + more)
    rcv = edge_points(axes, dt, 6)
    mode = rng.integers(0, 5)   # 0 interpolation, 1 tt_from_rp, 2 return_rays, 3 matrix M / L, 4 the same with the rays
    iv = bool(dim == 3 and rng.random() < 0.3)
    print(f"#{n} dim={dim} {np.dtype(dt).name} nc={nc} dx={dx} dz={dz} org={org} cell={cell} weno={weno} translate={translate} mode={mode} iv={iv}", flush=True)
    try:
        if dim == 3:
            O.solve3d(dt, nc, dx, org, s.flatten('F'), src, rcv=rcv, cell_slowness=cell, translate=translate, weno=weno,
                      tt_from_rp=mode == 1, return_rays=mode in (2, 4), compute_m=mode >= 3 and not cell, interp_vel=iv)
            O.compute_slowness3d(dt, nc, dx, org, s.flatten('F'), np.vstack([rcv, src]), cell, translate, iv)
        else:
            O.solve2d(dt, nc, dx, dz, org, s.ravel(), src, rcv=rcv, cell_slowness=cell, weno=weno, tt_from_rp=mode == 1, return_rays=mode in (2, 4),
                      compute_L=mode >= 3 and cell)
            O.compute_slowness2d(dt, nc, dx, dz, org, s.ravel(), np.vstack([rcv, src]), cell)
        walked += mode > 0
    except RuntimeError as e:   # a point the grid refuses, a ray that leaves the grid or does not end: the reference throws / hangs too
        failed += 1
    n += 1
print(f"done: {n} configurations, {walked} with raypath walks, {failed} refused (outside / ray left the grid); no ASan report")
