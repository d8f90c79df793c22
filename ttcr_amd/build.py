"""Build the HIP extension in-tree: ttcr_amd/libttcr_amd.so (gfx950).

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED for parity: the local
solver must round a1 + s*dx (and every other product/sum pair) exactly like the reference,
which a fused multiply-add would not (ttcr_amd/csrc/fsm_kernels.h header).

Two translation units, compiled to objects under ttcr_amd/csrc/_obj and linked:
  fsm_capi.hip    the C ABI, the host side and every kernel but one
  fsm_fast.hip    the sweep kernels with tolerance-grade arithmetic (option "arith" = 1)
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libttcr_amd.so")
INC = os.path.join("..", "..", "include", "ttcr_amd.h")
# source -> (extra flags, files it is compiled from)
UNITS = {
    "fsm_capi.hip": ([], ["fsm_capi.hip", "fsm_kernels.h", "fsm_fast_api.h", "fsm_march_levels.inc", "fsm_fast.hip", INC]),
    "fsm_fast.hip": ([], ["fsm_fast.hip", "fsm_fast_api.h", "fsm_kernels.h", "fsm_march_levels.inc"]),
}
SOURCES = list(UNITS)
DEPS = sorted({d for _, ds in UNITS.values() for d in ds})
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def source_hash():
    """sha256 (16 hex digits) of everything the device code is compiled from: kernel sources + compiler flags.  Profiles
    under profiles/ are tagged with it so that bench.py never reports counters taken with other kernels."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    for src in SOURCES:
        h.update(" ".join(UNITS[src][0]).encode())
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


def needs_build():
    return _stale(LIB, DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    procs = []
    for src, (extra, deps) in UNITS.items():
        if force or _stale(_obj(src), deps):
            # (the id is that of ALL device sources: a change to any of them recompiles fsm_capi.hip, which depends on every header)
            cmd = [_hipcc()] + FLAGS + extra + ['-DTTCR_BUILD_ID="%s"' % source_hash(), "-c", os.path.join(CSRC, src), "-o", _obj(src)]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
