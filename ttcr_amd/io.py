"""File formats either side of the FSM path: what the reference reads
and writes around a solve, without VTK.

  read_src / read_rcv          Src<T>::init (ttcr/Src.h:62-131), Rcv<T>::init (ttcr/Rcv.h:78-166):
                               plain text (count, then rows), legacy-VTK ASCII POINTS, CRT ('/' rows)
  save_rcvfile / save_rcv_tt   Rcv<T>::save_rcvfile (ttcr/Rcv.h:192-206), Rcv<T>::save_tt (:168-190)
  save_tt / load_tt            Grid3Drn::saveTT / loadTT (ttcr/Grid3Drn.h:2679-2815), Grid2Drn::saveTT
                               (ttcr/Grid2Drn.h:419-500): 1 = text, 2 = VTK rectilinear grid, 3 = binary
  read_vtr / write_vtr         the vtkXMLRectilinearGrid files the reference's models and fields are kept
                               in (ttcr/grids.h:430-514, src/ttcrpy/rgrid.pyx:1201-1380): XML, inline
                               base64, vtkZLibDataCompressor blocks -- decoded here with zlib/base64 only
  model_from_vtr               the array-name conventions of Grid3d.builder (rgrid.pyx:1315-1379)

Host-side only: numpy arrays in, numpy arrays out; nothing here touches the device."""
import base64
import struct
import xml.etree.ElementTree as ET
import zlib

import numpy as np

_VTK_TYPES = {"Float32": "f4", "Float64": "f8", "Int8": "i1", "UInt8": "u1", "Int16": "i2", "UInt16": "u2",
              "Int32": "i4", "UInt32": "u4", "Int64": "i8", "UInt64": "u8"}
_VTK_NAMES = {np.dtype(v): k for k, v in _VTK_TYPES.items()}
_BLOCK = 32768  # vtkXMLWriter's default compression block


# ------------------------------------------------------------------ Src / Rcv text files

def _read_points(fname, ndim, with_t0):
    """points (and t0) in one of the three layouts Src::init / Rcv::init (and the 2-D twins, ttcr/Src2D.h:
    60-104, ttcr/Rcv2D.h:84-165) accept"""
    with open(fname) as f:
        text = f.read()
    lines = text.split("\n")
    first = lines[0] if lines else ""
    if "vtk" in first:
        # legacy VTK: line 3 must say ASCII, coordinates follow the POINTS line; 2-D keeps x and z
        if len(lines) < 3 or "ASCII" not in lines[2]:
            raise ValueError("Error: vtk file should be ascii.")
        k = 2
        while "POINTS" not in lines[k]:
            k += 1
        n = int(lines[k].split()[1])
        vals = " ".join(lines[k + 1:]).split()[:3 * n]
        pts = np.array(vals, dtype=np.float64).reshape(-1, 3)
        pts = pts if ndim == 3 else pts[:, [0, 2]]
        return pts, np.zeros(pts.shape[0])
    if first.rstrip("\r").endswith("/"):
        # CRT format: "label x [y] z /" per row.  Like the reference, the first line only serves the
        # format test: reading resumes on line 2 (ttcr/Src.h:106-118 does not rewind in this branch)
        rows = []
        for ln in lines[1:]:
            t = ln.split()
            if len(t) >= ndim + 2 and t[ndim + 1] == "/":
                rows.append([float(v) for v in t[1:1 + ndim]])
        pts = np.array(rows, dtype=np.float64).reshape(-1, ndim)
        return pts, np.zeros(pts.shape[0])
    tok = text.split()
    n = int(tok[0])
    ncol = ndim + (1 if with_t0 else 0)
    a = np.array(tok[1:1 + ncol * n], dtype=np.float64).reshape(n, ncol)
    return np.ascontiguousarray(a[:, :ndim]), (np.ascontiguousarray(a[:, ndim]) if with_t0 else np.zeros(n))


def read_src(fname, ndim=3):
    """-> (coords (n, ndim), t0 (n,)); plain files hold `nsrc` then `x y z t0` rows (ttcr/Src.h:119-129),
    `x z t0` rows in 2-D (ttcr/Src2D.h:92-101)"""
    return _read_points(fname, ndim, True)


def read_rcv(fname, ndim=3):
    """-> coords (n, ndim); plain files hold `nrcv` then `x y z` rows (ttcr/Rcv.h:147-165), `x z` in 2-D"""
    return _read_points(fname, ndim, False)[0]


def save_rcvfile(fname, coords):
    """Rcv::save_rcvfile: count, then x<TAB>y<TAB>z in scientific notation with 17 digits"""
    c = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
    with open(fname, "w") as f:
        f.write("%d\n" % c.shape[0])
        for p in c:
            f.write("%.17e\t%.17e\t%.17e\n" % (p[0], p[1], p[2]))


def save_rcv_tt(fname, tt):
    """Rcv::save_tt: one row per receiver, arrivals (direct, then reflectors) TAB-separated, precision 9"""
    a = np.asarray(tt)
    a = a.reshape(a.shape[0], -1)
    with open(fname, "w") as f:
        for row in a:
            f.write("\t".join("%.9g" % float(v) for v in row) + "\n")


# ------------------------------------------------------------------ VTK XML rectilinear grids

def _decode_array(el, byte_order, header_type, compressed, appended):
    dt = np.dtype(_VTK_TYPES[el.get("type")]).newbyteorder("<" if byte_order == "LittleEndian" else ">")
    fmt = el.get("format", "ascii")
    ncomp = int(el.get("NumberOfComponents", "1"))
    if fmt == "ascii":
        a = np.array((el.text or "").split(), dtype=np.float64).astype(dt.newbyteorder("="))
    else:
        hdt = np.dtype(_VTK_TYPES[header_type]).newbyteorder(dt.byteorder)
        hs = hdt.itemsize
        if fmt == "appended":
            enc, blob = appended
            off = int(el.get("offset"))
            if enc == "raw":
                raw = _decode_raw(blob[off:], hdt, compressed)
            else:
                raw = _decode_b64(blob[off:], hdt, hs, compressed)
        else:
            raw = _decode_b64("".join((el.text or "").split()).encode(), hdt, hs, compressed)
        a = np.frombuffer(raw, dtype=dt).astype(dt.newbyteorder("="))
    return a.reshape(-1, ncomp) if ncomp > 1 else a


def _b64len(nbytes):
    return 4 * ((nbytes + 2) // 3)


def _decode_b64(b, hdt, hs, compressed):
    if not compressed:
        # length prefix and data form ONE base64 stream
        n = int(np.frombuffer(base64.b64decode(b[:_b64len(hs)])[:hs], dtype=hdt)[0])
        return base64.b64decode(b[:_b64len(hs + n)])[hs:hs + n]
    head = np.frombuffer(base64.b64decode(b[:_b64len(3 * hs)])[:3 * hs], dtype=hdt)
    nblocks = int(head[0])
    if nblocks == 0:
        return b""
    hl = _b64len((3 + nblocks) * hs)
    sizes = np.frombuffer(base64.b64decode(b[:hl])[:(3 + nblocks) * hs], dtype=hdt)[3:3 + nblocks]
    data = base64.b64decode(b[hl:hl + _b64len(int(sizes.sum()))])
    out, pos = [], 0
    for cs in sizes:
        out.append(zlib.decompress(data[pos:pos + int(cs)]))
        pos += int(cs)
    return b"".join(out)


def _decode_raw(b, hdt, compressed):
    hs = hdt.itemsize
    if not compressed:
        n = int(np.frombuffer(b[:hs], dtype=hdt)[0])
        return b[hs:hs + n]
    nblocks = int(np.frombuffer(b[:hs], dtype=hdt)[0])
    if nblocks == 0:
        return b""
    sizes = np.frombuffer(b[3 * hs:(3 + nblocks) * hs], dtype=hdt)
    pos, out = (3 + nblocks) * hs, []
    for cs in sizes:
        out.append(zlib.decompress(b[pos:pos + int(cs)]))
        pos += int(cs)
    return b"".join(out)


def read_vtr(fname):
    """Read a vtkXMLRectilinearGrid (.vtr) file: one piece; ascii, inline-binary or appended arrays,
    with or without the zlib compressor.

    Returns dict(x, y, z, point_data={name: array}, cell_data={name: array}); data arrays are flat in VTK
    order (x fastest), as vtk_to_numpy returns them."""
    with open(fname, "rb") as f:
        blob = f.read()
    appended = None
    k = blob.find(b"<AppendedData")
    if k >= 0:
        # raw appended data is not XML: cut it out before parsing
        e = blob.find(b">", k)
        enc = "raw" if b'encoding="raw"' in blob[k:e] else "base64"
        us = blob.find(b"_", e)
        end = blob.rfind(b"</AppendedData>")
        payload = blob[us + 1:end]
        appended = (enc, payload if enc == "raw" else b"".join(payload.split()))
        blob = blob[:e + 1] + blob[end:]
    root = ET.fromstring(blob)
    if root.get("type") != "RectilinearGrid":
        raise ValueError("%s: not a RectilinearGrid file" % fname)
    bo = root.get("byte_order", "LittleEndian")
    ht = root.get("header_type", "UInt32")
    comp = root.get("compressor")
    if comp not in (None, "vtkZLibDataCompressor"):
        raise ValueError("%s: unsupported compressor %s" % (fname, comp))
    rg = root.find("RectilinearGrid")
    pieces = rg.findall("Piece")
    if len(pieces) != 1:
        raise ValueError("%s: %d pieces (one expected)" % (fname, len(pieces)))
    piece = pieces[0]

    def dec(el):
        return _decode_array(el, bo, ht, comp is not None, appended)

    coords = [dec(el) for el in piece.find("Coordinates").findall("DataArray")]
    out = dict(x=coords[0], y=coords[1], z=coords[2], point_data={}, cell_data={})
    for tag, key in (("PointData", "point_data"), ("CellData", "cell_data")):
        sec = piece.find(tag)
        if sec is not None:
            for el in sec.findall("DataArray"):
                out[key][el.get("Name")] = dec(el)
    return out


def _encode_array(a):
    raw = np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
    blocks = [raw[i:i + _BLOCK] for i in range(0, len(raw), _BLOCK)]
    comp = [zlib.compress(b) for b in blocks]
    last = len(blocks[-1]) if blocks else 0
    head = struct.pack("<%dI" % (3 + len(blocks)), len(blocks), _BLOCK, last if last != _BLOCK else 0, *[len(c) for c in comp]) \
        if blocks else struct.pack("<3I", 0, _BLOCK, 0)
    return (base64.b64encode(head) + base64.b64encode(b"".join(comp))).decode()


def write_vtr(fname, x, y, z, point_data=None, cell_data=None, scalars=None):
    """Write a .vtr file the way vtkXMLRectilinearGridWriter does in binary mode (inline base64, zlib
    blocks of 32 KiB, UInt32 headers).  Data arrays are flat in VTK order (x fastest)."""
    x, y, z = (np.asarray(v, dtype=np.float64).ravel() for v in (x, y, z))
    ext = "0 %d 0 %d 0 %d" % (x.size - 1, y.size - 1, z.size - 1)
    npts, ncell = x.size * y.size * z.size, max(x.size - 1, 1) * max(y.size - 1, 1) * max(z.size - 1, 1)

    def arr(name, a, indent):
        a = np.asarray(a)
        if a.dtype not in _VTK_NAMES:
            a = a.astype(np.float64)
        rng = ' RangeMin="%r" RangeMax="%r"' % (float(a.min()), float(a.max())) if a.size else ""
        return ('%s<DataArray type="%s" Name="%s" format="binary"%s>\n%s  %s\n%s</DataArray>\n'
                % (indent, _VTK_NAMES[a.dtype], name, rng, indent, _encode_array(a.ravel()), indent))

    s = ['<?xml version="1.0"?>\n',
         '<VTKFile type="RectilinearGrid" version="0.1" byte_order="LittleEndian" header_type="UInt32" '
         'compressor="vtkZLibDataCompressor">\n',
         '  <RectilinearGrid WholeExtent="%s">\n' % ext, '  <Piece Extent="%s">\n' % ext]
    s.append('    <PointData%s>\n' % (' Scalars="%s"' % scalars if scalars else ""))
    for name, a in (point_data or {}).items():
        if np.asarray(a).size != npts:
            raise ValueError("Field %s has incorrect size" % name)
        s.append(arr(name, a, "      "))
    s.append("    </PointData>\n    <CellData>\n")
    for name, a in (cell_data or {}).items():
        if np.asarray(a).size != ncell:
            raise ValueError("Field %s has incorrect size" % name)
        s.append(arr(name, a, "      "))
    s.append("    </CellData>\n    <Coordinates>\n")
    for name, c in (("x", x), ("y", y), ("z", z)):
        s.append(arr(name, c, "      "))
    s.append("    </Coordinates>\n  </Piece>\n  </RectilinearGrid>\n</VTKFile>\n")
    with open(fname, "w") as f:
        f.write("".join(s))


def write_vtp_lines(fname, rays):
    """Raypaths as a vtkPolyData of polylines (rgrid.pyx:1284-1312 _save_raypaths): `rays` is a list of
    (npts, 3) [or (npts, 2): x, z] arrays; binary inline arrays like write_vtr."""
    pts, conn, offs, n0 = [], [], [], 0
    for r in rays:
        r = np.asarray(r, dtype=np.float64)
        if r.shape[1] == 2:
            r = np.column_stack([r[:, 0], np.zeros(r.shape[0]), r[:, 1]])
        pts.append(r)
        conn.append(np.arange(n0, n0 + r.shape[0], dtype=np.int64))
        n0 += r.shape[0]
        offs.append(n0)
    P = np.vstack(pts).astype(np.float32) if pts else np.zeros((0, 3), np.float32)
    conn = np.concatenate(conn) if conn else np.zeros(0, np.int64)
    offs = np.asarray(offs, dtype=np.int64)

    def arr(name, a, ncomp=1):
        return ('        <DataArray type="%s" Name="%s"%s format="binary">\n          %s\n        </DataArray>\n'
                % (_VTK_NAMES[a.dtype], name, ' NumberOfComponents="%d"' % ncomp if ncomp > 1 else "", _encode_array(a.ravel())))

    empty = '        <DataArray type="Int64" Name="%s" format="binary">\n          %s\n        </DataArray>\n'
    e64 = np.zeros(0, np.int64)
    s = ['<?xml version="1.0"?>\n<VTKFile type="PolyData" version="0.1" byte_order="LittleEndian" header_type="UInt32" '
         'compressor="vtkZLibDataCompressor">\n  <PolyData>\n',
         '    <Piece NumberOfPoints="%d" NumberOfVerts="0" NumberOfLines="%d" NumberOfStrips="0" NumberOfPolys="0">\n'
         % (P.shape[0], len(rays)),
         '      <PointData>\n      </PointData>\n      <CellData>\n      </CellData>\n      <Points>\n',
         arr("Points", P, 3), '      </Points>\n']
    for tag, c, o in (("Verts", e64, e64), ("Lines", conn, offs), ("Strips", e64, e64), ("Polys", e64, e64)):
        s.append('      <%s>\n' % tag)
        s.append(arr("connectivity", c) if c.size else empty % ("connectivity", _encode_array(c)))
        s.append(arr("offsets", o) if o.size else empty % ("offsets", _encode_array(o)))
        s.append('      </%s>\n' % tag)
    s.append('    </Piece>\n  </PolyData>\n</VTKFile>\n')
    with open(fname, "w") as f:
        f.write("".join(s))


def read_vtp_lines(fname):
    """read back the polylines of a PolyData file written by write_vtp_lines / vtkXMLPolyDataWriter (inline
    binary): -> list of (npts, 3) arrays"""
    root = ET.parse(fname).getroot()
    bo, ht, comp = root.get("byte_order", "LittleEndian"), root.get("header_type", "UInt32"), root.get("compressor")
    piece = root.find("PolyData").find("Piece")

    def dec(el):
        return _decode_array(el, bo, ht, comp is not None, None)

    P = dec(piece.find("Points").find("DataArray")).reshape(-1, 3)
    lines = {el.get("Name"): dec(el) for el in piece.find("Lines").findall("DataArray")}
    out, a = [], 0
    for b in lines["offsets"]:
        out.append(np.asarray(P[lines["connectivity"][a:int(b)]], dtype=np.float64))
        a = int(b)
    return out


_MODEL_NAMES = ("Slowness", "slowness", "Velocity", "velocity", "P-wave velocity")


def model_from_vtr(fname):
    """What Grid3d.builder / Grid2d.builder take from a model file (rgrid.pyx:1346-1372): node
    coordinates, the slowness (1/velocity when the array is a velocity) in VTK order, and whether it is
    cell data.  -> dict(x, y, z, slowness (flat, x fastest), cell_slowness (0/1), name)"""
    d = read_vtr(fname)
    for name in _MODEL_NAMES:
        if name in d["point_data"]:
            cell, a = 0, d["point_data"][name]
            break
        if name in d["cell_data"]:
            cell, a = 1, d["cell_data"][name]
            break
    else:
        raise ValueError("File should contain slowness or velocity data")
    a = np.asarray(a, dtype=np.float64)
    s = a if "lowness" in name else 1.0 / a
    return dict(x=d["x"], y=d["y"], z=d["z"], slowness=s, cell_slowness=cell, name=name)


# ------------------------------------------------------------------ traveltime fields on disk

def _node_coords(grid):
    """node coordinates as the reference stores them: xmin + i*dx in the grid's precision
    (ttcr/Grid3Drn.h:389-399, dy = dz = dx for FSM grids :44); a translated grid keeps coordinates
    relative to its origin (:362-372)"""
    t = np.dtype(grid._dtype).type
    if grid._ndim == 3:
        axes = ((grid._x, grid._dx), (grid._y, grid._dx), (grid._z, grid._dx))
    else:
        axes = ((grid._x, grid._dx), (grid._z, grid._dz))
    out = []
    for c, d in axes:
        c0 = t(0) if getattr(grid, "translate_grid", False) else t(c[0])
        out.append(c0 + np.arange(c.size).astype(t) * t(d))
    return out


def save_tt(grid, fname, all=0, thread_no=0, format=1):
    """Grid3Drn::saveTT / Grid2Drn::saveTT for a ttcr_amd grid: the field of slot `thread_no` to
    fname + '.dat' (1: text, precision 12), '.vtr' (2: point array "Travel time", Float64) or '.bin'
    (3: x, y, z, tt per node in the grid's precision).  FSM grids only have primary nodes, `all` is moot."""
    nd = grid._ndim
    tt = grid._flat_tt(int(thread_no))
    cs = _node_coords(grid)
    if nd == 3:
        x, y, z = cs
        cols = [np.tile(x, y.size * z.size), np.tile(np.repeat(y, x.size), z.size), np.repeat(z, x.size * y.size), tt]
    else:
        x, z = cs
        cols = [np.repeat(x, z.size), np.tile(z, x.size), tt]
    if format == 1:
        with open(fname + ".dat", "w") as f:
            for row in zip(*[c.astype(np.float64) for c in cols]):
                f.write("\t".join("%.12g" % v for v in row) + "\n")
    elif format == 2:
        if nd == 3:
            write_vtr(fname + ".vtr", x, y, z, point_data={"Travel time": tt.astype(np.float64)}, scalars="Travel time")
        else:
            # nodes are z-fastest in 2-D; VTK wants x fastest on the (nx, 1, nz) grid
            t2 = tt.reshape(x.size, z.size).T.ravel()
            write_vtr(fname + ".vtr", x, [0.0], z, point_data={"Travel time": t2.astype(np.float64)}, scalars="Travel time")
    elif format == 3:
        np.stack(cols, axis=1).astype(grid._dtype).tofile(fname + ".bin")
    else:
        raise RuntimeError("Unsupported format for saving traveltimes")


def load_tt(fname, shape, format=1, dtype=np.float64):
    """Grid3Drn::loadTT: read back a field written by save_tt.  `shape` = node counts ((nx, ny, nz) or
    (nx, nz)); returns the flat field in the solver's node order."""
    n, nd = int(np.prod(shape)), len(shape)
    if format == 1:
        return np.loadtxt(fname + ".dat", dtype=np.float64, ndmin=2)[:n, nd].astype(dtype)
    if format == 2:
        d = read_vtr(fname + ".vtr")
        a = np.asarray(d["point_data"]["Travel time"], dtype=dtype)
        return a if nd == 3 else a.reshape(shape[1], shape[0]).T.ravel()
    if format == 3:
        return np.fromfile(fname + ".bin", dtype=dtype).reshape(n, nd + 1)[:, nd].copy()
    raise RuntimeError("Unsupported format for traveltimes")
