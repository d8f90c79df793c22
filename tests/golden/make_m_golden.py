"""Golden vectors of the matrix M (compute_M): tests/golden/m_golden.npz, made with the COMPILED, UNMODIFIED reference
(oracle/_ref, build container only) through its overload Grid3D::raytrace(Tx, t0, Rx, traveltimes, m_data, threadNo)
(ttcr/Grid3D.h:743-772 -> Grid3Drn::getRaypath(..., m_data, ...), ttcr/Grid3Drn.h:1503-1800).

  <case>/<dtype>/slowness, src, t0, rcv, meta = (ncx, ncy, ncz, dx, ox, oy, oz, translate, weno)      inputs
  <case>/<dtype>/tt_rcv           traveltimes of that overload (0 for a receiver on the source)
  <case>/<dtype>/m_off, m_j, m_v  per receiver n the entries [m_off[n], m_off[n+1]) in the order the reference pushed them
  <case>/<dtype>/rm_tt_rcv, rm_off, rm_j, rm_v, rm_ray_off, rm_ray_pts   the same through the overload that keeps the rays as well,
                                  Grid3D::raytrace(Tx, t0, Rx, traveltimes, r_data, m_data, threadNo) (ttcr/Grid3D.h:646-680 ->
                                  Grid3Drn.h:2144-2470; what ttcrpy calls for compute_M with return_rays): another matrix

usage: python tests/golden/make_m_golden.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O


def cases_m():
    rng = np.random.default_rng(2024)
    out = []
    # smooth and rough node grids, cubic and not, translated origin, source on a node / off node, receivers on the source,
    # next to the far faces (node indices one past the grid in the reference's weights) and in the interior
    for name, nc, dx, org, tr, weno, rough, src in (
            ("m_grad", (20, 20, 20), 1.0, (0.0, 0.0, 0.0), 0, 0, False, [3.3, 4.1, 5.7]),
            ("m_rough", (18, 14, 11), 0.5, (1.0, -2.0, 0.0), 0, 0, True, [3.25, 1.0, 2.5]),
            ("m_translate", (12, 16, 10), 2.0, (500000.0, 4000000.0, -1000.0), 1, 0, True, [500009.0, 4000011.5, -993.0]),
            ("m_weno", (16, 16, 16), 1.0, (0.0, 0.0, 0.0), 0, 1, False, [8.0, 8.0, 8.0]),
            # a source of two points with their own origin times (aggregate_src): every ray ends at the point it reaches first
            ("m_two_points", (20, 16, 14), 1.0, (0.0, 0.0, 0.0), 0, 0, True, [[3.3, 4.1, 5.7], [16.2, 11.4, 8.9]]),
            # three points of one source within a cell of each other: the end game of a ray runs once per point within a cell
            # diagonal, on a point the previous run has moved (ttcr/Grid3Drn.h:1628-1795 / :2262-2460)
            ("m_close_points", (16, 18, 14), 1.0, (0.0, 0.0, 0.0), 0, 0, True, [[8.3, 7.1, 6.4], [8.6, 6.9, 6.65], [7.9, 7.45, 6.5]])):
        nn = tuple(v + 1 for v in nc)
        if rough:
            s = rng.uniform(0.4, 1.0, nn[0] * nn[1] * nn[2])
        else:
            z = org[2] + np.arange(nn[2]) * dx
            s = np.repeat(1.0 / (1.0 + 0.1 * (z - org[2])), nn[0] * nn[1])
        lo = np.array(org); hi = lo + np.array(nc) * dx
        rcv = rng.uniform(lo + 0.6 * dx, hi - 0.6 * dx, (7, 3))
        srcs = np.atleast_2d(np.array(src, dtype=float))
        rcv = np.vstack([rcv, srcs[:1], hi - 0.25 * dx, lo + np.array([0.5, 0.5, 0.5]) * dx])
        out.append(dict(name=name, nc=nc, dx=dx, org=org, translate=tr, weno=weno, slowness=s, src=srcs, t0=np.array([0.25, 0.4, 0.1][:srcs.shape[0]]), rcv=rcv))
    return out


def main():
    O.build(with_ref=True)
    assert O.have_ref(), "the compiled reference is needed"
    out = {}
    for c in cases_m():
        for dt in (np.float32, np.float64):
            key = f"{c['name']}/{np.dtype(dt).name}"
            r = O.ref_solve3d(dt, c["nc"], c["dx"], c["org"], c["slowness"], c["src"], t0=c["t0"], rcv=c["rcv"], weno=bool(c["weno"]),
                              translate=bool(c["translate"]), compute_m=True)
            out[key + "/tt_rcv"] = r["tt_rcv"]
            out[key + "/m_off"] = np.cumsum([0] + [len(j) for j, _ in r["m"]]).astype(np.int64)
            out[key + "/m_j"] = np.concatenate([j for j, _ in r["m"]]).astype(np.int64)
            out[key + "/m_v"] = np.concatenate([v for _, v in r["m"]]).astype(dt)
            r2 = O.ref_solve3d(dt, c["nc"], c["dx"], c["org"], c["slowness"], c["src"], t0=c["t0"], rcv=c["rcv"], weno=bool(c["weno"]),
                               translate=bool(c["translate"]), compute_m=True, return_rays=True)
            out[key + "/rm_tt_rcv"] = r2["tt_rcv"]
            out[key + "/rm_off"] = np.cumsum([0] + [len(j) for j, _ in r2["m"]]).astype(np.int64)
            out[key + "/rm_j"] = np.concatenate([j for j, _ in r2["m"]]).astype(np.int64)
            out[key + "/rm_v"] = np.concatenate([v for _, v in r2["m"]]).astype(dt)
            out[key + "/rm_ray_off"] = np.cumsum([0] + [len(ray) for ray in r2["rays"]]).astype(np.int64)
            out[key + "/rm_ray_pts"] = np.concatenate(r2["rays"]).astype(dt)
            print(key, "rm entries", out[key + "/rm_j"].size, "nonzero", int(np.count_nonzero(out[key + "/rm_v"])))
            print(key, "entries", out[key + "/m_j"].size, "nonzero", int(np.count_nonzero(out[key + "/m_v"])),
                  "past the grid", int(np.sum(out[key + "/m_j"] >= np.prod([v + 1 for v in c["nc"]]))))
        out[c["name"] + "/slowness"] = c["slowness"]
        out[c["name"] + "/src"] = c["src"]; out[c["name"] + "/t0"] = c["t0"]; out[c["name"] + "/rcv"] = c["rcv"]
        out[c["name"] + "/meta"] = np.array(list(c["nc"]) + [c["dx"]] + list(c["org"]) + [c["translate"], c["weno"]], dtype=np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "m_golden.npz"), **out)


if __name__ == "__main__":
    main()
