"""Golden vectors of the ray-projection matrix L (compute_L, 2-D grids with cell slowness): tests/golden/l_golden.npz, made with
the COMPILED, UNMODIFIED reference (oracle/_ref, build container only) through its overloads Grid2D::raytrace(Tx, t0, Rx,
traveltimes, [r_data,] l_data, threadNo) (ttcr/Grid2D.h:583-640 -> Grid2Drn::getRaypath(..., l_data, ...), ttcr/Grid2Drn.h:1852-2190).

  <case>/slowness (cells, x-major z fastest), src, t0, rcv, meta = (ncx, ncz, dx, dz, ox, oz, weno)          inputs
  <case>/<dtype>/<norays|rays>/tt_rcv             traveltimes of the overload without / with r_data
  <case>/<dtype>/<norays|rays>/l_off, l_i, l_v    per receiver n the entries [l_off[n], l_off[n+1]) as the reference sorted them
  <case>/<dtype>/rays/r_off, r_pts                the rays of the overload with r_data

usage: python tests/golden/make_l_golden.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O


def cases_l():
    O.build(with_ref=True)
    rng = np.random.default_rng(4242)
    out = []
    for name, nc, dx, dz, org, weno, kind, src in (
            ("l_const", (24, 18), 1.0, 1.0, (0.0, 0.0), 0, "const", [[7.3, 5.1]]),
            ("l_layers", (30, 40), 0.5, 0.5, (0.0, 0.0), 0, "layers", [[6.0, 3.0]]),            # source on a node
            ("l_rough_xz", (21, 33), 1.25, 0.75, (8.0, -4.0), 0, "rough", [[20.4, 3.3]]),      # dx != dz, origin not 0 (all exact in fp32: the
                                                                                                 # Python layer takes dx = x[1] - x[0] in the grid dtype)
            ("l_weno", (32, 32), 1.0, 1.0, (0.0, 0.0), 1, "layers", [[15.5, 16.5]]),
            ("l_two_points", (28, 22), 1.0, 1.0, (0.0, 0.0), 0, "rough", [[5.2, 4.4], [22.7, 17.1]])):
        ncx, ncz = nc
        if kind == "const":
            s = np.full((ncx, ncz), 0.5)
        elif kind == "layers":
            s = np.tile(1.0 / (1.0 + 0.1 * (np.floor(np.arange(ncz) * dz) + 0.5)), (ncx, 1))
        else:
            s = rng.uniform(0.3, 1.0, (ncx, ncz))
        lo = np.array(org); hi = lo + np.array([ncx * dx, ncz * dz])
        srcs = np.atleast_2d(np.array(src, dtype=float))
        rcv = np.column_stack([rng.uniform(lo[0] + 0.3 * dx, hi[0] - 0.3 * dx, 12), rng.uniform(lo[1] + 0.3 * dz, hi[1] - 0.3 * dz, 12)])
        rcv = np.vstack([rcv, srcs[:1], [hi[0], hi[1]], [lo[0], hi[1] - 0.5 * dz], srcs[:1] + [0.4 * dx, 0.3 * dz]])
        # (the walk of these overloads gives up when a step leaves the grid: receivers the reference throws for are left out here;
        # tests/test_l_matrix.py checks the error separately)
        keep = []
        for q in range(rcv.shape[0]):
            try:
                O.ref_solve2d(np.float64, nc, dx, dz, org, s.ravel(), srcs, t0=np.array([0.25, 0.4][:srcs.shape[0]]), rcv=rcv[q:q + 1], weno=bool(weno),
                              cell_slowness=True, compute_L=True)
                O.ref_solve2d(np.float32, nc, dx, dz, org, s.ravel(), srcs, t0=np.array([0.25, 0.4][:srcs.shape[0]]), rcv=rcv[q:q + 1], weno=bool(weno),
                              cell_slowness=True, compute_L=True)
                keep.append(q)
            except RuntimeError as e:
                print(name, "receiver", rcv[q], "left out:", str(e).splitlines()[0])
        rcv = rcv[keep]
        out.append(dict(name=name, nc=nc, dx=dx, dz=dz, org=org, weno=weno, slowness=s.ravel(), src=srcs, t0=np.array([0.25, 0.4][:srcs.shape[0]]), rcv=rcv))
    return out


def main():
    assert O.have_ref(), "the compiled reference is needed"
    out = {}
    for c in cases_l():
        for dt in (np.float32, np.float64):
            for rays in (False, True):
                key = f"{c['name']}/{np.dtype(dt).name}/{'rays' if rays else 'norays'}"
                r = O.ref_solve2d(dt, c["nc"], c["dx"], c["dz"], c["org"], c["slowness"], c["src"], t0=c["t0"], rcv=c["rcv"], weno=bool(c["weno"]),
                                  cell_slowness=True, compute_L=True, return_rays=rays)
                out[key + "/tt_rcv"] = r["tt_rcv"]
                out[key + "/l_off"] = np.cumsum([0] + [len(i) for i, _ in r["l"]]).astype(np.int64)
                out[key + "/l_i"] = np.concatenate([i for i, _ in r["l"]]).astype(np.int64)
                out[key + "/l_v"] = np.concatenate([v for _, v in r["l"]]).astype(dt)
                if rays:
                    out[key + "/r_off"] = np.cumsum([0] + [len(p) for p in r["rays"]]).astype(np.int64)
                    out[key + "/r_pts"] = np.concatenate(r["rays"]).astype(dt)
                dup = sum(len(i) - len(set(i.tolist())) for i, _ in r["l"])
                print(key, "entries", out[key + "/l_i"].size, "of them in a cell that has another entry", dup)
        for k in ("slowness", "src", "t0", "rcv"):
            out[c["name"] + "/" + k] = c[k]
        out[c["name"] + "/meta"] = np.array(list(c["nc"]) + [c["dx"], c["dz"]] + list(c["org"]) + [c["weno"]], dtype=np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "l_golden.npz"), **out)
    print("wrote tests/golden/l_golden.npz")


if __name__ == "__main__":
    main()
