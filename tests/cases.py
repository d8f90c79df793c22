"""Synthetic models and the parity-case matrix shared by the golden generator, the
oracle tests and the GPU parity tests.

Models follow the reference's own generators (paths relative to /root/reference):
  gradient  V(z) = a + b z, a = 1, b = 0.1 on [0,20]^3   tests/files/mk_models3d.py:16-18,217
  layers    cell slowness 1/(a + b (floor(z_lo)+0.5))     tests/files/mk_models3d.py:139-140
  constant  V0 = 3                                         tests/files/mk_constant_models.py:20
plus two heterogeneous models (uniform random slowness, slow-sphere lens) that need
several Gauss-Seidel iterations, so the sweep ordering and stopping rule are exercised.
Random sources: mt19937_64(12345), uniform in [0.5,19.5] (tests/accuracy_grid3d.cpp:352-360).
"""
import numpy as np

A, B = 1.0, 0.1


def node_coords(n, length=20.0, cmin=0.0):
    """n nodes spanning [cmin, cmin+length]."""
    return cmin + np.arange(n, dtype=np.float64) * (length / (n - 1))


def gradient3d(nn, dx, zmin=0.0):
    """node slowness, flat x-fastest; nn = (nnx, nny, nnz)"""
    nnx, nny, nnz = nn
    z = zmin + np.arange(nnz) * dx
    s = 1.0 / (A + B * z)
    return np.repeat(s, nnx * nny)


def constant3d(nn, v0=3.0):
    return np.full(nn[0] * nn[1] * nn[2], 1.0 / v0)


def random3d(nn, seed=7):
    rng = np.random.default_rng(seed)
    return rng.uniform(0.25, 1.0, nn[0] * nn[1] * nn[2])


def lens3d(nn, dx):
    """background 0.4 s/km with a slow sphere (1.0 s/km) in the middle"""
    nnx, nny, nnz = nn
    x = np.arange(nnx) * dx
    y = np.arange(nny) * dx
    z = np.arange(nnz) * dx
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    c = np.array([x[-1], y[-1], z[-1]]) / 2
    r = np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2)
    s = np.where(r < 0.3 * min(x[-1], y[-1], z[-1]), 1.0, 0.4)
    return s.ravel()


def layers3d_cells(nc, dx):
    """cell slowness, flat x-fastest; nc = (ncx, ncy, ncz)"""
    ncx, ncy, ncz = nc
    zlo = np.arange(ncz) * dx
    s = 1.0 / (A + B * (np.floor(zlo) + 0.5))
    return np.repeat(s, ncx * ncy)


def gradient2d(nn, dz):
    """node slowness, flat z-fastest; nn = (nnx, nnz)"""
    nnx, nnz = nn
    z = np.arange(nnz) * dz
    return np.tile(1.0 / (A + B * z), nnx)


def random2d(nn, seed=11):
    rng = np.random.default_rng(seed)
    return rng.uniform(0.25, 1.0, nn[0] * nn[1])


def layers2d_cells(nc, dz):
    ncx, ncz = nc
    zlo = np.arange(ncz) * dz
    return np.tile(1.0 / (A + B * (np.floor(zlo) + 0.5)), ncx)


def mt_sources(n, ndim=3, lo=0.5, hi=19.5, seed=12345):
    """Sources as drawn by the reference's constant-model study
    (std::mt19937_64(12345) + uniform_real_distribution<double>(0.5,19.5)).
    numpy's MT19937 is the 32-bit generator, so the 64-bit one is implemented here."""
    w, nn, m, r = 64, 312, 156, 31
    a = 0xB5026F5AA96619E9
    mask = (1 << 64) - 1
    mt = [0] * nn
    mt[0] = seed
    for i in range(1, nn):
        mt[i] = (6364136223846793005 * (mt[i - 1] ^ (mt[i - 1] >> 62)) + i) & mask
    idx = nn
    out = []

    def nxt():
        nonlocal idx
        if idx >= nn:
            for i in range(nn):
                x = (mt[i] & 0xFFFFFFFF80000000) | (mt[(i + 1) % nn] & 0x7FFFFFFF)
                xa = x >> 1
                if x & 1:
                    xa ^= a
                mt[i] = mt[(i + m) % nn] ^ xa
            idx = 0
        y = mt[idx]
        idx += 1
        y ^= (y >> 29) & 0x5555555555555555
        y ^= (y << 17) & 0x71D67FFFEDA60000
        y ^= (y << 37) & 0xFFF7EEE000000000
        y ^= y >> 43
        return y & mask

    for _ in range(n * ndim):
        # libstdc++ generate_canonical<double,53> with a 64-bit engine: one draw / 2^64
        u = nxt() / 18446744073709551616.0
        if u >= 1.0:
            u = np.nextafter(1.0, 0.0)
        out.append(lo + (hi - lo) * u)
    return np.array(out).reshape(n, ndim)


def rcv_lattice3d(length=20.0, n=21):
    """441 receivers on the x = 0 plane integer lattice (== tests/files/rcv.dat scaled)."""
    v = np.linspace(0.0, length, n)
    Y, Z = np.meshgrid(v, v, indexing="ij")
    return np.stack([np.zeros(Y.size), Y.ravel(), Z.ravel()], axis=1)


def weno_ok(c):
    """Cases used for the WENO3 stage: every axis needs >= 3 cells (the reference's stencil reads
    idx+2 at idx == 1 unconditionally, ttcr/Grid3Drn.h:3086-3092), kept small for the fixture file."""
    return min(c["ncells"]) >= 3 and int(np.prod(np.array(c["ncells"]) + 1)) <= 20000


def rp_ok(c):
    """Cases used for traveltime-from-raypath: smooth models only (on rough random media the
    reference's steepest-descent walk does not terminate)."""
    return c["dim"] == 3 and weno_ok(c) and "random" not in c["name"]


# ------------------------------------------------------------------ case matrix
# Every case: dict(name, dim, ncells, dx[,dz], origin, slowness (float64), cell_slowness,
#                  src (n,dim), t0 (n,), rcv (m,dim), translate)
# The arrays are float64; each test casts them to the dtype under test.


def _rcv3(nc, dx, origin, rng):
    lo = np.array(origin)
    hi = lo + np.array(nc) * dx
    on = lo + rng.integers(0, np.array(nc) + 1, size=(6, 3)) * dx
    off = rng.uniform(lo, hi, size=(10, 3))
    mixed = on.copy()[:3]
    mixed[:, 0] = rng.uniform(lo[0], hi[0], size=3)  # on an edge / face
    return np.vstack([on, off, mixed, hi[None, :], lo[None, :]])


def cases3d():
    rng = np.random.default_rng(2024)
    out = []

    def add(name, nc, dx, origin, s, src, t0=None, cell=False, translate=False):
        src = np.atleast_2d(np.asarray(src, dtype=np.float64))
        out.append(dict(name=name, dim=3, ncells=tuple(nc), dx=dx, origin=tuple(origin),
                        slowness=np.asarray(s, dtype=np.float64), cell_slowness=cell,
                        src=src, t0=np.zeros(len(src)) if t0 is None else np.asarray(t0, float),
                        rcv=_rcv3(nc, dx, origin, rng), translate=translate))

    # (1) constant, source on a node (centre) and off-node: pins initFSM incl. the skipped corner
    for n in (17, 33):
        dx = 20.0 / (n - 1)
        nn = (n, n, n)
        add(f"const{n}_node", (n - 1,) * 3, dx, (0, 0, 0), constant3d(nn), [(n // 2) * dx] * 3)
        add(f"const{n}_off", (n - 1,) * 3, dx, (0, 0, 0), constant3d(nn), np.array([3.3, 4.1, 5.7]) * dx)
    # (2) gradient, corner source (== tests/files/src.dat) and interior off-node source
    for n in (21, 41):
        dx = 20.0 / (n - 1)
        nn = (n, n, n)
        add(f"grad{n}_corner", (n - 1,) * 3, dx, (0, 0, 0), gradient3d(nn, dx), [0.0, 0.0, 0.0])
        add(f"grad{n}_off", (n - 1,) * 3, dx, (0, 0, 0), gradient3d(nn, dx), [7.3, 11.2, 5.9])
    # (3) layers, cell slowness through Grid3Drcfs
    for nc in (20, 40):
        dx = 20.0 / nc
        add(f"layers{nc}_cells", (nc,) * 3, dx, (0, 0, 0), layers3d_cells((nc,) * 3, dx), [10.2, 9.7, 0.4],
            cell=True)
    add("randomcells_12x10x14", (12, 10, 14), 0.5, (0, 0, 0),
        np.random.default_rng(21).uniform(0.25, 1.0, 12 * 10 * 14), [2.2, 3.1, 4.4], cell=True)
    # (4) heterogeneous models on a non-cubic-count grid, non-zero origin
    nn = (24, 20, 28)
    nc = tuple(v - 1 for v in nn)
    add("random_24x20x28", nc, 0.5, (-2.0, 3.0, 10.0), random3d(nn), [-2.0 + 5.2, 3.0 + 4.4, 10.0 + 6.1])
    add("random_24x20x28_node", nc, 0.5, (-2.0, 3.0, 10.0), random3d(nn, seed=8),
        [-2.0 + 11 * 0.5, 3.0 + 0 * 0.5, 10.0 + 27 * 0.5])
    add("lens_28x24x20", (27, 23, 19), 0.25, (0, 0, 0), lens3d((28, 24, 20), 0.25), [0.3, 0.2, 0.1])
    # (5) multi-point source with distinct t0 (aggregate_src); later points overwrite earlier ones
    add("random_multisrc", nc, 0.5, (0, 0, 0), random3d(nn, seed=9),
        [[1.0, 1.0, 1.0], [1.2, 1.3, 1.1], [9.0, 7.5, 11.0]], t0=[0.0, 0.05, 0.3])
    # (6) translate_grid with large (UTM-like) coordinates
    add("grad21_translate", (20,) * 3, 1.0, (500000.0, 4000000.0, -1000.0), gradient3d((21,) * 3, 1.0),
        [500007.3, 4000011.2, -994.1], translate=True)
    # (7) source in the far corner cell / on the max faces (getCellNo's xmax clamp)
    add("const17_maxcorner", (16,) * 3, 1.25, (0, 0, 0), constant3d((17,) * 3), [20.0, 20.0 - 0.3, 20.0])
    # (8) thin grids (2 nodes along an axis)
    add("random_thin", (1, 15, 9), 0.5, (0, 0, 0), random3d((2, 16, 10), seed=10), [0.2, 3.3, 1.1])
    return out


def _rcv2(nc, dx, dz, origin, rng):
    lo = np.array(origin)
    hi = lo + np.array(nc) * np.array([dx, dz])
    on = lo + rng.integers(0, np.array(nc) + 1, size=(5, 2)) * np.array([dx, dz])
    off = rng.uniform(lo, hi, size=(8, 2))
    mixed = on.copy()[:2]
    mixed[:, 1] = rng.uniform(lo[1], hi[1], size=2)
    return np.vstack([on, off, mixed, hi[None, :], lo[None, :]])


def rp2_ok(c):
    """2-D cases used for the raypath family (tt_from_rp / return_rays): smooth media and the cell models; on the
    rough node-slowness random media the reference's steepest-descent walk does not terminate"""
    return c["dim"] == 2 and not c["name"].startswith("random2d")


def rot_ok(c):
    """Cases run with rotated_template=True as well (sweep45 only exists for 2-D square cells)"""
    return c["dim"] == 2 and c["dx"] == c["dz"]


def cases2d():
    rng = np.random.default_rng(4048)
    out = []

    def add(name, nc, dx, dz, origin, s, src, t0=None, cell=False):
        src = np.atleast_2d(np.asarray(src, dtype=np.float64))
        out.append(dict(name=name, dim=2, ncells=tuple(nc), dx=dx, dz=dz, origin=tuple(origin),
                        slowness=np.asarray(s, dtype=np.float64), cell_slowness=cell,
                        src=src, t0=np.zeros(len(src)) if t0 is None else np.asarray(t0, float),
                        rcv=_rcv2(nc, dx, dz, origin, rng), translate=False))

    add("grad2d_65_corner", (64, 64), 0.3125, 0.3125, (0, 0), gradient2d((65, 65), 0.3125), [0.0, 0.0])
    add("grad2d_65_off", (64, 64), 0.3125, 0.3125, (0, 0), gradient2d((65, 65), 0.3125), [7.31, 11.27])
    add("random2d_64x96", (63, 95), 0.25, 0.25, (5.0, -3.0), random2d((64, 96)), [5.0 + 3.3, -3.0 + 17.9])
    add("random2d_64x96_node", (63, 95), 0.25, 0.25, (5.0, -3.0), random2d((64, 96), seed=12),
        [5.0 + 20 * 0.25, -3.0 + 95 * 0.25])
    add("random2d_xz", (63, 95), 0.25, 0.4, (0, 0), random2d((64, 96), seed=13), [3.3, 17.9])
    add("grad2d_xz_node", (40, 30), 0.5, 0.25, (0, 0), gradient2d((41, 31), 0.25), [10.0, 2.5])
    add("layers2d_cells", (40, 40), 0.5, 0.5, (0, 0), layers2d_cells((40, 40), 0.5), [10.2, 0.4], cell=True)
    add("randomcells2d_30x20", (30, 20), 0.5, 0.5, (0, 0),
        np.random.default_rng(22).uniform(0.25, 1.0, 30 * 20), [7.2, 3.1], cell=True)
    add("random2d_multisrc", (63, 95), 0.25, 0.25, (0, 0), random2d((64, 96), seed=14),
        [[1.0, 1.0], [1.1, 1.3], [9.0, 20.0]], t0=[0.0, 0.02, 0.4])
    return out
