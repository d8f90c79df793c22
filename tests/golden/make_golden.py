"""Generate tests/golden/fsm_golden.npz from the UNMODIFIED reference, compiled where it
lies under /root/reference (oracle/_ref/libttcr_ref.so, recipe in oracle/Makefile).

Run in the build container only:   python tests/golden/make_golden.py

For every case of tests/cases.py and both dtypes the file holds the reference's outputs
  <case>/<dtype>/tt       full node traveltime field (Grid3Drn::getTT, flat, x-fastest / z-fastest)
  <case>/<dtype>/niter    get_niter()
  <case>/<dtype>/tt_rcv   Grid3Drn::getTraveltime at the case's receivers (tt_from_rp = false)
  <case>/<dtype>/rp_tt_rcv, rpv_tt_rcv   receiver traveltimes of the weno + tt_from_rp solve
                          (Grid3Drn::getTraveltimeFromRaypath), without / with interp_vel; *_error = 1 when
                          the reference throws "going outside grid" for the case
  <case>/<dtype>/rot_tt, rot_niter, rot_tt_rcv   first-order solve with rotated_template=True (sweep45 after
                          every sweep, ttcr/Grid2Drnfs.h:277-286), for the cases of cases.rot_ok()
  <case>/<dtype>/rays_tt_rcv, rays_off, rays_pts   weno + return_rays solve: receiver traveltimes and raypaths of
                          Grid3Drn::getRaypath (ray n = points [off[n], off[n+1])), or rays_error = 1
  <case>/<dtype>/rp2_tt_rcv, rays2_tt_rcv, rays2_off, rays2_pts   2-D cases of cases.rp2_ok(): receiver traveltimes of
                          Grid2Drn::getTraveltimeFromRaypath, and traveltimes + raypaths ((x, z) points) of Grid2Drn::getRaypath
  <case>/<dtype>/weno_*   the same four outputs (+ niterw) of the two-stage weno=True solve, for the
                          cases of cases.weno_ok()
and the inputs  <case>/slowness (float64; cast to the dtype under test), so that the
vectors do not depend on numpy's random generator staying stable.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    O.build(with_ref=True)
    out = {}
    for c in cases.cases3d() + cases.cases2d():
        out[f"{c['name']}/slowness"] = c["slowness"]
        out[f"{c['name']}/src"] = c["src"]
        out[f"{c['name']}/t0"] = c["t0"]
        out[f"{c['name']}/rcv"] = c["rcv"]
        for dt in (np.float32, np.float64):
            if c["dim"] == 3:
                r = O.ref_solve3d(dt, c["ncells"], c["dx"], c["origin"], c["slowness"], c["src"], c["t0"],
                                  cell_slowness=c["cell_slowness"], translate=c["translate"], rcv=c["rcv"])
            else:
                r = O.ref_solve2d(dt, c["ncells"], c["dx"], c["dz"], c["origin"], c["slowness"], c["src"],
                                  c["t0"], cell_slowness=c["cell_slowness"], rcv=c["rcv"])
            key = f"{c['name']}/{np.dtype(dt).name}"
            out[key + "/tt"] = r["tt"]
            out[key + "/niter"] = np.int32(r["niter"])
            out[key + "/tt_rcv"] = r["tt_rcv"]
            print(key, "niter", r["niter"])
            if cases.rot_ok(c):
                r = O.ref_solve2d(dt, c["ncells"], c["dx"], c["dz"], c["origin"], c["slowness"], c["src"],
                                  c["t0"], cell_slowness=c["cell_slowness"], rcv=c["rcv"], rotated=True)
                out[key + "/rot_tt"] = r["tt"]
                out[key + "/rot_niter"] = np.int32(r["niter"])
                out[key + "/rot_tt_rcv"] = r["tt_rcv"]
                print(key, "rotated niter", r["niter"])
            if cases.weno_ok(c):
                # two-stage solve with the third-order WENO stage (weno=True, the ttcrpy default)
                if c["dim"] == 3:
                    r = O.ref_solve3d(dt, c["ncells"], c["dx"], c["origin"], c["slowness"], c["src"], c["t0"],
                                      cell_slowness=c["cell_slowness"], translate=c["translate"], rcv=c["rcv"], weno=True)
                else:
                    r = O.ref_solve2d(dt, c["ncells"], c["dx"], c["dz"], c["origin"], c["slowness"], c["src"],
                                      c["t0"], cell_slowness=c["cell_slowness"], rcv=c["rcv"], weno=True)
                out[key + "/weno_tt"] = r["tt"]
                out[key + "/weno_niter"] = np.int32(r["niter"])
                out[key + "/weno_niterw"] = np.int32(r["niterw"])
                out[key + "/weno_tt_rcv"] = r["tt_rcv"]
                print(key, "weno niter", r["niter"], r["niterw"])
            if cases.rp_ok(c):
                # default ttcrpy 3-D configuration: weno + traveltime integrated along the raypath
                for tag, iv in (("rp", False), ("rpv", True)):
                    try:
                        r = O.ref_solve3d(dt, c["ncells"], c["dx"], c["origin"], c["slowness"], c["src"], c["t0"],
                                          cell_slowness=c["cell_slowness"], translate=c["translate"], rcv=c["rcv"],
                                          weno=True, tt_from_rp=True, interp_vel=iv)
                        out[key + f"/{tag}_tt_rcv"] = r["tt_rcv"]
                        out[key + f"/{tag}_error"] = np.int32(0)
                    except RuntimeError as e:
                        assert "going outside grid" in str(e)
                        out[key + f"/{tag}_error"] = np.int32(1)
                    print(key, tag, "error" if int(out[key + f"/{tag}_error"]) else "ok")
            if cases.rp_ok(c):
                # raytrace(..., return_rays=True): Grid3D::raytrace(Tx,t0,Rx,tt,r_data,threadNo)
                try:
                    r = O.ref_solve3d(dt, c["ncells"], c["dx"], c["origin"], c["slowness"], c["src"], c["t0"],
                                      cell_slowness=c["cell_slowness"], translate=c["translate"], rcv=c["rcv"],
                                      weno=True, return_rays=True)
                    out[key + "/rays_tt_rcv"] = r["tt_rcv"]
                    out[key + "/rays_off"] = np.cumsum([0] + [len(x) for x in r["rays"]]).astype(np.int64)
                    out[key + "/rays_pts"] = np.vstack(r["rays"])
                    out[key + "/rays_error"] = np.int32(0)
                except RuntimeError as e:
                    assert "going outside grid" in str(e)
                    out[key + "/rays_error"] = np.int32(1)
                print(key, "rays", "error" if int(out[key + "/rays_error"]) else "ok")
            if cases.rp2_ok(c):
                # 2-D raypath family: Grid2d(..., tt_from_rp=1) and raytrace(..., return_rays=True), weno as in ttcrpy
                w = cases.weno_ok(c)
                kw = dict(cell_slowness=c["cell_slowness"], rcv=c["rcv"], weno=w)
                for tag, extra in (("rp2", dict(tt_from_rp=True)), ("rays2", dict(return_rays=True))):
                    try:
                        r = O.ref_solve2d(dt, c["ncells"], c["dx"], c["dz"], c["origin"], c["slowness"], c["src"], c["t0"],
                                          **kw, **extra)
                        out[key + f"/{tag}_tt_rcv"] = r["tt_rcv"]
                        if tag == "rays2":
                            out[key + "/rays2_off"] = np.cumsum([0] + [len(x) for x in r["rays"]]).astype(np.int64)
                            out[key + "/rays2_pts"] = np.vstack(r["rays"])
                        out[key + f"/{tag}_error"] = np.int32(0)
                    except RuntimeError as e:
                        assert "going outside grid" in str(e)
                        out[key + f"/{tag}_error"] = np.int32(1)
                    print(key, tag, "error" if int(out[key + f"/{tag}_error"]) else "ok", flush=True)
    path = os.path.join(HERE, "fsm_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
