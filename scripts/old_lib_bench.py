"""bench.py with a library built from an OLDER commit (TTCR_AMD_LIB): symbols added since are dropped from the
loader's table first.  For A/B runs of kernel changes on one box only."""
import ctypes, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # first: the library must bind to the HIP runtime torch ships, like in bench.py
import ttcr_amd._lib as L
lib = ctypes.CDLL(L.LIB_PATH)
for name in list(L.SYMBOLS):
    if not hasattr(lib, name):
        L.SYMBOLS.pop(name)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
