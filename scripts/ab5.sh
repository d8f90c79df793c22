#!/bin/bash
# same-box A/B of tuning builds (round 5): scripts/ab5.sh variants/a.so variants/b.so ...
for lib in "$@"; do
  echo "== $lib"
  export TTCR_AMD_LIB=$PWD/$lib
  python scripts/lone_time.py 512 3 1
  python scripts/lone_time.py 256 3 1
  TTCR_FSM_SKIP=0 python scripts/lone_time.py 512 3 8
  python scripts/lone_time.py 512 3 8
  python scripts/lone_time.py 512 2 64
done
