#!/bin/bash
# same-box A/B of tuning builds: scripts/ab.sh variants/a.so variants/b.so ...
# (512^3: one source, 8 and 64 sources with / without skipping; WENO 256^3; 2-D 4096^2)
for lib in "$@"; do
  echo "== $lib"
  export TTCR_AMD_LIB=$PWD/$lib
  timeout 300 python scripts/solve_time.py 512 1 4 3 | tail -1
  timeout 600 python scripts/skip_sweep.py 512 8,64
  timeout 600 python scripts/skip_sweep.py 256 1,8 1
  timeout 300 python scripts/skip_sweep2d.py 4096 1,16,64 2>&1 | tail -3
done
