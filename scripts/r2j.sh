#!/bin/bash
# GPU-box script: skewed level march -- parity first, then timings against the unskewed build
O=gpurun_out/r2j; mkdir -p $O
(time python -m pytest tests/test_parity_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q -x -k "not 2d" --durations=5) > $O/pytest_parity.txt 2>&1
tail -6 $O/pytest_parity.txt
for L in "" $PWD/variants/noskew.so; do
  echo "== lib ${L:-default}"
  for S in 1 2 8 64; do TTCR_AMD_LIB=$L python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done
  TTCR_AMD_LIB=$L python scripts/solve_time.py 256 1 2 3 2>&1 | tail -1
done > $O/ab.txt 2>&1
cat $O/ab.txt
TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 python scripts/solve_time.py 512 1 2 2 2>&1 | grep prof | tail -1
TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 python scripts/solve_time.py 512 64 2 2 2>&1 | grep prof | tail -1
