#!/bin/bash
# Run on the GPU box: the headline bench line for every library build under variants/ (compiler-flag
# or tuning variants built with the same sources), selected through TTCR_AMD_LIB.
# usage: scripts/variants_run.sh [bench args...]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out
for L in $ROOT/ttcr_amd/libttcr_amd.so $ROOT/variants/*.so; do
  [ -f "$L" ] || continue
  R=$(TTCR_AMD_LIB=$L timeout 300 python $ROOT/bench.py --no-cpu-baseline "$@" 2>&1 | tail -1)
  echo "$(basename $L) $(echo "$R" | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("avg_launch_us_hip_events"))
except Exception as e: print("FAILED", e)')" | tee -a $ROOT/gpurun_out/variants.log
done
