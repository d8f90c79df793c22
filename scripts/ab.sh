#!/bin/bash
# same-box A/B of two tuning builds: scripts/ab.sh variants/a.so variants/b.so  (512^3: one source, 8 and 64 sources with / without skipping)
for lib in "$@"; do
  echo "== $lib"
  export TTCR_AMD_LIB=$PWD/$lib
  timeout 300 python scripts/solve_time.py 512 1 4 3 | tail -1
  timeout 600 python scripts/skip_sweep.py 512 8,64
done
