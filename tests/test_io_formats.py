"""File formats either side of the FSM path: ttcr_amd/io.py against the reference's
own data files (tests/files/: models and analytic fields as .vtr, src/rcv text files -- copies of the DATA
files under the reference's tests/files/), and a replay of the reference's model-file tests
(tests/test_grid3d.cpp:149-200 testGrid3D, tests/test_grid2d.cpp:200-248 testGrid2D, FAST_SWEEPING rows)
with the CPU oracle here and with the HIP path in the `gpu` twin below."""
import os

import numpy as np
import pytest

import cases
from ttcr_amd import io

FILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "files")


def F(name):
    return os.path.join(FILES, name)


def test_src_rcv_plain_files():
    xyz, t0 = io.read_src(F("src.dat"))
    np.testing.assert_array_equal(xyz, [[0.0, 0.0, 0.0]])
    np.testing.assert_array_equal(t0, [0.0])
    rcv = io.read_rcv(F("rcv.dat"))
    v = np.arange(21.0)   # 21 x 21 integer lattice on the z = 0 plane, y fastest
    X, Y = np.meshgrid(v, v, indexing="ij")
    np.testing.assert_array_equal(rcv, np.stack([X.ravel(), Y.ravel(), np.zeros(441)], axis=1))
    xz, t0 = io.read_src(F("src2d.dat"), ndim=2)
    np.testing.assert_array_equal(xz, [[0.0, 0.0]])
    r2 = io.read_rcv(F("rcv2d.dat"), ndim=2)
    assert r2.shape == (441, 2) and r2.min() == 0.0 and r2.max() == 20.0


def test_src_rcv_other_layouts(tmp_path):
    """legacy-VTK ASCII and CRT ('/'-terminated rows) layouts of Src::init / Rcv::init"""
    p = tmp_path / "pts.vtk"
    p.write_text("# vtk DataFile Version 3.0\npoints\nASCII\nDATASET POLYDATA\nPOINTS 3 float\n"
                 "1 2 3\n4.5 5.5 6.5 7 8\n9\n")
    want = np.array([[1, 2, 3], [4.5, 5.5, 6.5], [7, 8, 9.0]])
    np.testing.assert_array_equal(io.read_rcv(str(p)), want)
    xyz, t0 = io.read_src(str(p))
    np.testing.assert_array_equal(xyz, want)
    np.testing.assert_array_equal(t0, np.zeros(3))
    np.testing.assert_array_equal(io.read_rcv(str(p), ndim=2), want[:, [0, 2]])
    q = tmp_path / "pts.crt"
    # (the first line of a CRT file only serves the format test, like in the reference)
    q.write_text("S0 0.0 0.0 0.0 /\nS1 1.0 2.0 3.0 /\nS2 4.0 5.0 6.0 /\n")
    np.testing.assert_array_equal(io.read_rcv(str(q)), [[1, 2, 3], [4, 5, 6.0]])
    q2 = tmp_path / "pts2.crt"
    q2.write_text("S0 0.0 0.0 /\nS1 1.0 3.0 /\nS2 4.0 6.0 /\n")
    np.testing.assert_array_equal(io.read_src(str(q2), ndim=2)[0], [[1, 3], [4, 6.0]])


def test_rcv_writers_text_format(tmp_path):
    c = np.array([[0.1, 2.0, -3.5e-7], [1e10, 1.0 / 3.0, 0.0]])
    io.save_rcvfile(str(tmp_path / "r.dat"), c)
    txt = (tmp_path / "r.dat").read_text().split("\n")
    assert txt[0] == "2"
    assert txt[1] == "1.00000000000000006e-01\t2.00000000000000000e+00\t-3.49999999999999984e-07"
    np.testing.assert_array_equal(io.read_rcv(str(tmp_path / "r.dat")), c)   # 17 digits round-trip exactly
    io.save_rcv_tt(str(tmp_path / "tt.dat"), np.array([1.0 / 3.0, 12345.678912345, 0.0]))
    assert (tmp_path / "tt.dat").read_text() == "0.333333333\n12345.6789\n0\n"


def test_vtr_models_match_the_generators():
    """the reference's model files decode to the formulas of tests/files/mk_models3d.py / mk_models2d.py"""
    g = io.read_vtr(F("gradient_medium.vtr"))
    for k in "xyz":
        np.testing.assert_allclose(g[k], np.arange(41) * 0.5, rtol=0, atol=1e-12)
    s = g["point_data"]["Slowness"].reshape(41, 41, 41, order="F")
    np.testing.assert_allclose(s, np.broadcast_to(1.0 / (cases.A + cases.B * g["z"]), s.shape), rtol=1e-14)
    lay = io.read_vtr(F("layers_medium.vtr"))
    sc = lay["cell_data"]["Slowness"].reshape(40, 40, 40, order="F")
    zlo = lay["z"][:-1]
    np.testing.assert_allclose(sc, np.broadcast_to(1.0 / (cases.A + cases.B * (np.floor(zlo) + 0.5)), sc.shape), rtol=1e-14)
    g2 = io.read_vtr(F("gradient_fine2d.vtr"))
    assert g2["y"].size == 1 and g2["x"].size == 101 and g2["z"].size == 101
    s2 = g2["point_data"]["Slowness"].reshape(101, 101, order="F")   # (x, z), x fastest in the file
    np.testing.assert_allclose(s2, np.broadcast_to(1.0 / (cases.A + cases.B * g2["z"]), s2.shape), rtol=1e-14)
    m = io.model_from_vtr(F("layers_fine2d.vtr"))
    assert m["cell_slowness"] == 1 and m["name"] == "Slowness" and m["slowness"].size == 100 * 100


def test_vtr_roundtrip_and_variants(tmp_path):
    rng = np.random.default_rng(5)
    x, y, z = np.arange(4) * 0.5, np.arange(3) * 1.5 + 1, np.arange(40) * 0.25
    pd = {"Travel time": rng.uniform(0, 5, 4 * 3 * 40), "Velocity": rng.uniform(1, 4, 4 * 3 * 40).astype(np.float32)}
    cd = {"Slowness": rng.uniform(0.2, 1, 3 * 2 * 39)}
    p = str(tmp_path / "m.vtr")
    io.write_vtr(p, x, y, z, point_data=pd, cell_data=cd)
    d = io.read_vtr(p)
    for k, v in (("x", x), ("y", y), ("z", z)):
        np.testing.assert_array_equal(d[k], v)
    for k in pd:
        np.testing.assert_array_equal(d["point_data"][k], pd[k])
        assert d["point_data"][k].dtype == pd[k].dtype
    np.testing.assert_array_equal(d["cell_data"]["Slowness"], cd["Slowness"])
    m = io.model_from_vtr(p)   # builder's name order: point 'Slowness' ... before cell, 'Velocity' is inverted
    assert m["cell_slowness"] == 1 and m["name"] == "Slowness"
    # the same content as an ascii file and as uncompressed inline binary / raw appended data
    import base64
    import struct
    raw = pd["Travel time"].tobytes()
    variants = {
        "ascii": ('format="ascii">' + " ".join(repr(float(v)) for v in pd["Travel time"]), "", ""),
        "binary": ('format="binary">' + base64.b64encode(struct.pack("<I", len(raw)) + raw).decode(), "", ""),
        "appended": ('format="appended" offset="0">', '<AppendedData encoding="raw">_',
                     "</AppendedData>"),
    }
    for name, (attr, app_open, app_close) in variants.items():
        body = ('<?xml version="1.0"?>\n<VTKFile type="RectilinearGrid" version="0.1" byte_order="LittleEndian">\n'
                '<RectilinearGrid WholeExtent="0 3 0 2 0 39"><Piece Extent="0 3 0 2 0 39">\n'
                '<PointData><DataArray type="Float64" Name="Travel time" ' + attr + '</DataArray></PointData>\n'
                '<Coordinates>' + "".join('<DataArray type="Float64" Name="c" format="ascii">%s</DataArray>'
                                          % " ".join(repr(float(v)) for v in c) for c in (x, y, z)) +
                '</Coordinates></Piece></RectilinearGrid>\n').encode()
        if app_open:
            body += app_open.encode() + struct.pack("<I", len(raw)) + raw + app_close.encode()
        body += b"\n</VTKFile>\n"
        q = tmp_path / (name + ".vtr")
        q.write_bytes(body)
        np.testing.assert_array_equal(io.read_vtr(str(q))["point_data"]["Travel time"], pd["Travel time"])


def test_vtp_roundtrip(tmp_path):
    rays = [np.array([[0, 0, 0], [1, 1, 1], [2, 2, 3.5]]), np.array([[5, 5, 5], [4, 4, 4.0]])]
    p = str(tmp_path / "r.vtp")
    io.write_vtp_lines(p, rays)
    back = io.read_vtp_lines(p)
    assert len(back) == 2
    for a, b in zip(rays, back):
        np.testing.assert_array_equal(a, b)


class _FakeGrid:
    """what save_tt needs from a ttcr_amd grid, without a device"""

    def __init__(self, dt, coords, tt, translate=False):
        self._dtype, self._ndim = dt, len(coords)
        if self._ndim == 3:
            self._x, self._y, self._z = (np.asarray(c, dtype=dt) for c in coords)
        else:
            self._x, self._z = (np.asarray(c, dtype=dt) for c in coords)
            self._dz = float(self._z[1] - self._z[0])
        self._dx = float(self._x[1] - self._x[0])
        self.translate_grid = translate
        self._tt = np.asarray(tt, dtype=dt)

    def _flat_tt(self, thread_no):
        return self._tt


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_save_load_tt_roundtrip(tmp_path, dt):
    rng = np.random.default_rng(8)
    x, y, z = 1.0 + np.arange(5) * 0.3, np.arange(4) * 0.3, -2 + np.arange(6) * 0.3
    tt = rng.uniform(0, 9, 5 * 4 * 6)
    g = _FakeGrid(dt, (x, y, z), tt)
    base = str(tmp_path / "f")
    for fmt in (1, 2, 3):
        io.save_tt(g, base, format=fmt)
        back = io.load_tt(base, (5, 4, 6), format=fmt, dtype=dt)
        if fmt == 1:
            np.testing.assert_allclose(back, g._tt, rtol=1e-11 if dt == np.float64 else 0)
        else:
            np.testing.assert_array_equal(back, g._tt)
    rows = open(base + ".dat").read().split("\n")
    assert len(rows) == 5 * 4 * 6 + 1 and rows[1].split("\t")[0] == "%.12g" % float(g._x[1])   # x fastest
    b = np.fromfile(base + ".bin", dtype=dt).reshape(-1, 4)
    np.testing.assert_array_equal(b[:5, 0], (dt(x[0]) + np.arange(5).astype(dt) * dt(g._dx)))
    # 2-D: nodes z fastest, the .vtr is written on the (nx, 1, nz) grid
    t2 = rng.uniform(0, 9, 5 * 6)
    g2 = _FakeGrid(dt, (x, z), t2)
    for fmt in (1, 2, 3):
        io.save_tt(g2, base + "2", format=fmt)
        back = io.load_tt(base + "2", (5, 6), format=fmt, dtype=dt)
        np.testing.assert_allclose(back, g2._tt, rtol=1e-11 if dt == np.float64 else 1e-7)
    with pytest.raises(RuntimeError, match="Unsupported format"):
        io.save_tt(g, base, format=4)


# ------------------------------------------------------------------ replay of the reference's model-file tests

def rel_error(ref_file, rcv, tt, ndim):
    """get_rel_error of tests/test_grid3d.cpp:67-96: analytic value at the grid point nearest to each
    receiver, mean relative misfit over receivers 1.. (receiver 0 sits on the source)"""
    d = io.read_vtr(ref_file)
    (name, a), = d["point_data"].items()
    assert name in ("Travel Time", "Travel time", "travel time")
    a = a.reshape(d["x"].size, d["y"].size, d["z"].size, order="F")
    ix = np.abs(rcv[:, [0]] - d["x"][None, :]).argmin(1)
    iz = np.abs(rcv[:, [-1]] - d["z"][None, :]).argmin(1)
    iy = np.abs(rcv[:, [1]] - d["y"][None, :]).argmin(1) if ndim == 3 else np.zeros_like(ix)
    ref = a[ix, iy, iz]
    return float(np.mean(np.abs((ref[1:] - tt[1:]) / ref[1:])))


MODELS3D = [("layers_medium.vtr", "sol_analytique_couches_tt.vtr"), ("gradient_medium.vtr", "sol_analytique_gradient_tt.vtr")]
MODELS2D = [("layers_fine2d.vtr", "sol_analytique_couches2d_tt.vtr"), ("gradient_fine2d.vtr", "sol_analytique_gradient2d_tt.vtr")]


@pytest.mark.parametrize("model,ref", MODELS3D)
def test_replay_testGrid3D_fast_sweeping_oracle(oracle, model, ref):
    m = io.model_from_vtr(F(model))
    src, t0 = io.read_src(F("src.dat"))
    rcv = io.read_rcv(F("rcv.dat"))
    nc = (m["x"].size - 1, m["y"].size - 1, m["z"].size - 1)
    dx = (m["x"][-1] - m["x"][0]) / nc[0]   # d = range / (nnodes-1), ttcr/grids.h:455-457
    r = oracle.solve3d(np.float64, nc, dx, (m["x"][0], m["y"][0], m["z"][0]), m["slowness"], src, t0,
                       cell_slowness=bool(m["cell_slowness"]), rcv=rcv, weno=True)
    assert rel_error(F(ref), rcv, r["tt_rcv"], 3) < 0.01


@pytest.mark.parametrize("model,ref", MODELS2D)
def test_replay_testGrid2D_fast_sweeping_oracle(oracle, model, ref):
    m = io.model_from_vtr(F(model))
    src, t0 = io.read_src(F("src2d.dat"), ndim=2)
    rcv = io.read_rcv(F("rcv2d.dat"), ndim=2)
    nx, nz = m["x"].size, m["z"].size
    dim = (nx - 1, nz - 1) if m["cell_slowness"] else (nx, nz)
    s = m["slowness"].reshape(dim, order="F").ravel()   # the solver is z-fastest in 2-D
    r = oracle.solve2d(np.float64, (nx - 1, nz - 1), (m["x"][-1] - m["x"][0]) / (nx - 1), (m["z"][-1] - m["z"][0]) / (nz - 1),
                       (m["x"][0], m["z"][0]), s, src, t0, cell_slowness=bool(m["cell_slowness"]), rcv=rcv, weno=True)
    assert rel_error(F(ref), rcv, r["tt_rcv"], 2) < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("model,ref", MODELS3D)
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_replay_testGrid3D_fast_sweeping_hip(tmp_path, oracle, model, ref, dt):
    """the same reference test through Grid3d.builder on the GPU, plus saveTT in the three formats"""
    import ttcr_amd

    cls = ttcr_amd.Grid3d_d if dt == np.float64 else ttcr_amd.Grid3d_f
    g = cls.builder(F(model), method="FSM", tt_from_rp=0, weno=1)
    src, t0 = io.read_src(F("src.dat"))
    rcv = io.read_rcv(F("rcv.dat"))
    tt = g.raytrace(np.hstack([t0[:, None], src]), rcv)
    assert rel_error(F(ref), rcv, tt, 3) < 0.01
    # bit-exact vs the oracle on the same file-borne model
    m = io.model_from_vtr(F(model))
    nc = (m["x"].size - 1, m["y"].size - 1, m["z"].size - 1)
    r = oracle.solve3d(dt, nc, g.dx, (m["x"][0], m["y"][0], m["z"][0]), m["slowness"].astype(dt), src, t0,
                       cell_slowness=bool(m["cell_slowness"]), rcv=rcv, weno=True)
    np.testing.assert_array_equal(tt, r["tt_rcv"])
    field = g.get_grid_traveltimes()
    np.testing.assert_array_equal(field.flatten("F"), r["tt"])
    base = str(tmp_path / "Grid3Drcfs_tt_grid")
    for fmt in (1, 2, 3):
        io.save_tt(g, base, 0, 0, fmt)
        back = io.load_tt(base, field.shape, format=fmt, dtype=dt)
        if fmt == 1:
            np.testing.assert_allclose(back, r["tt"], rtol=6e-12 if dt == np.float64 else 0)
        else:
            np.testing.assert_array_equal(back, r["tt"])
    g.to_vtk({"Travel time": field, "Slowness": g.get_slowness()}, str(tmp_path / "out"))
    d = io.read_vtr(str(tmp_path / "out.vtr"))
    np.testing.assert_array_equal(d["point_data"]["Travel time"], r["tt"].astype(np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("model,ref", MODELS2D)
def test_replay_testGrid2D_fast_sweeping_hip(model, ref):
    import ttcr_amd

    m = io.model_from_vtr(F(model))
    nx, nz = m["x"].size, m["z"].size
    dim = (nx - 1, nz - 1) if m["cell_slowness"] else (nx, nz)
    g = ttcr_amd.Grid2d(m["x"], m["z"], cell_slowness=m["cell_slowness"], method="FSM", weno=1)
    src, t0 = io.read_src(F("src2d.dat"), ndim=2)
    rcv = io.read_rcv(F("rcv2d.dat"), ndim=2)
    tt = g.raytrace(np.hstack([t0[:, None], src]), rcv, slowness=m["slowness"].reshape(dim, order="F"))
    assert rel_error(F(ref), rcv, tt, 2) < 0.02
