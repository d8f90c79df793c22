"""Build the HIP extension in-tree: ttcr_amd/libttcr_amd.so (gfx950).

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED for parity: the local
solver must round a1 + s*dx (and every other product/sum pair) exactly like the reference,
which a fused multiply-add would not (ttcr_amd/csrc/fsm_kernels.h header).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libttcr_amd.so")
SOURCES = ["fsm_capi.hip"]
DEPS = ["fsm_capi.hip", "fsm_kernels.h", "fsm_wave_kernels.h", "fsm_march_levels.inc", os.path.join("..", "..", "include", "ttcr_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
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
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
