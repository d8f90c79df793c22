#!/bin/bash
# INTEGRATION.md section 1 as commands: the three edits that seat the MI355X backend in ttcrpy's own Cython wrapper, applied to
# a SCRATCH copy of the reference's src/ttcrpy (nothing of the reference is kept in this repository -- the script holds only
# the edit commands), then cythonized and compiled against the adapters and libttcr_amd.so.
#   usage: integration/patch_ttcrpy.sh <reference root> <scratch dir> [-O0]
# Edits (line numbers of the reference as surveyed):
#   rgrid.pxd:122-125  the `fsm_gpu` 3-D class declaration -> Grid3Drnfs_amd (one adapter serves node and cell grids: a leading bool)
#   rgrid.pxd:142-145  the cell twin of that declaration: dropped
#   rgrid.pxd:277-283  the same for the 2-D pair (the adapter takes rotated_template as well)
#   rgrid.pyx:209-216, 246-253, 2932-2936, 2957-2961, 4586-4590, 4611-4615  the `method == 'FSM' and fsm_gpu` constructor calls
#   setup.py:80,95     libraries=['OpenCL'] -> ['ttcr_amd'] (here: the link line below)
set -euo pipefail
REF=${1:?reference root}; OUT=${2:?scratch dir}; OPT=${3:--O1}
REPO=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$OUT"; mkdir -p "$OUT/ttcrpy"
cp "$REF"/src/ttcrpy/{rgrid.pyx,rgrid.pxd,common.pxd,verbose.cpp,verbose.h,typedefs.h,utils_cython.h,__init__.py} "$OUT/ttcrpy/" 2>/dev/null || true
cd "$OUT/ttcrpy"
# ---- edit 2: rgrid.pxd
sed -i -e 's/cdef extern from "Grid3Drnfs_OpenCL.h"/cdef extern from "Grid3Drnfs_amd.h"/' \
       -e 's/cdef cppclass Grid3Drnfs_OpenCL\[T1,T2\](Grid3Drn\[T1,T2,Node3Dn\[T1,T2\]\]):/cdef cppclass Grid3Drnfs_amd[T1,T2](Grid3D[T1,T2]):/' \
       -e 's/^\( *\)Grid3Drnfs_OpenCL(T2, T2, T2,/\1Grid3Drnfs_amd(bool, T2, T2, T2,/' \
       -e '/cdef extern from "Grid3Drcfs_OpenCL.h"/,+3d' \
       -e '/cdef extern from "Grid2Drcfs_OpenCL.h"/,+3d' \
       -e 's/cdef extern from "Grid2Drnfs_OpenCL.h"/cdef extern from "Grid2Drnfs_amd.h"/' \
       -e 's/cdef cppclass Grid2Drnfs_OpenCL\[T1,T2,S\](Grid2Drn\[T1,T2,S,Node2Dn\[T1,T2\]\]):/cdef cppclass Grid2Drnfs_amd[T1,T2,S](Grid2D[T1,T2,S]):/' \
       -e 's/^\( *\)Grid2Drnfs_OpenCL(T2, T2, T1, T1, T1, T1, T1, int, bool, bool, size_t) except +/\1Grid2Drnfs_amd(bool, T2, T2, T1, T1, T1, T1, T1, int, bool, bool, bool, size_t) except +/' rgrid.pxd
# ---- edit 3: rgrid.pyx, the constructor calls behind `method == 'FSM' and fsm_gpu`
sed -i -e 's/new Grid3Drcfs_OpenCL\[double,uint32_t\](nx,/new Grid3Drnfs_amd[double,uint32_t](True, nx,/' \
       -e 's/new Grid3Drnfs_OpenCL\[double,uint32_t\](nx,/new Grid3Drnfs_amd[double,uint32_t](False, nx,/' \
       -e 's/new Grid2Drcfs_OpenCL\[\(double\|float\),uint32_t,sxz\[\(double\|float\)\]\](nx, nz,/new Grid2Drnfs_amd[\1,uint32_t,sxz[\2]](True, nx, nz,/' \
       -e 's/new Grid2Drnfs_OpenCL\[\(double\|float\),uint32_t,sxz\[\(double\|float\)\]\](nx, nz,/new Grid2Drnfs_amd[\1,uint32_t,sxz[\2]](False, nx, nz,/' rgrid.pyx
sed -i -e '/new Grid2Drnfs_amd\[/,+2 s/maxit, weno, tt_from_rp, n_threads)/maxit, weno, rotated_template, tt_from_rp, n_threads)/' rgrid.pyx
if grep -n "OpenCL\[" rgrid.pyx rgrid.pxd; then echo "an OpenCL class is still referenced" >&2; exit 3; fi
# ---- edit 1: build -- the adapters' directory and include/ on the include path, libttcr_amd.so instead of OpenCL on the link line
cd "$OUT"
python3 - "$REF" "$REPO" "$OPT" <<'PY'
import os, subprocess, sys, sysconfig
import numpy as np
from Cython.Build import cythonize
ref, repo, opt = sys.argv[1:4]
from setuptools import Extension
cythonize([Extension("ttcrpy.rgrid", ["ttcrpy/rgrid.pyx"], language="c++")], language_level=3, include_path=["."], quiet=True)
inc = [os.path.join(ref, "ttcr"), os.path.join(ref, "boost_1_91_0"), os.path.join(ref, "eigen-5.0.0"), np.get_include(),
       sysconfig.get_paths()["include"], os.path.join(repo, "integration"), os.path.join(repo, "include"), "ttcrpy"]
so = "ttcrpy/rgrid" + sysconfig.get_config_var("EXT_SUFFIX")
libdir = os.path.join(repo, "ttcr_amd")
cmd = ["g++", "-std=c++17", opt, "-fPIC", "-shared", "-Wno-sign-compare", "-Wno-unused-result", "-w"] + ["-I" + d for d in inc] + \
      ["ttcrpy/rgrid.cpp", "ttcrpy/verbose.cpp", "-o", so, "-L" + libdir, "-lttcr_amd", "-Wl,-rpath," + libdir, "-pthread"]
subprocess.check_call(cmd)
print(os.path.abspath(so))
PY
