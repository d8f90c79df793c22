#!/bin/bash
# GPU-box script: trial pass (chunks evaluated level-independently first) vs the plain level march, same box
O=gpurun_out/r2e; mkdir -p $O
for L in "" $PWD/variants/notrial.so; do
  echo "== lib ${L:-default}"
  for S in 1 8 64; do TTCR_AMD_LIB=$L python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done
  TTCR_AMD_LIB=$L python scripts/solve_time.py 256 1 2 3 2>&1 | tail -1
  TTCR_AMD_LIB=$L python scripts/weno_batch.py 256 1 2>&1 | tail -1
  TTCR_AMD_LIB=$L python scripts/weno_batch.py 256 64 2>&1 | tail -1
  TTCR_AMD_LIB=$L python scripts/c5_run.py 2>&1 | tail -2
done > $O/ab.txt 2>&1
TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 TTCR_FSM_PROF_TRACE=$PWD/$O/trace1.bin python scripts/solve_time.py 512 1 2 2 > $O/prof1.txt 2>&1
python scripts/trace_analyze.py $O/trace1.bin > $O/trace1.txt 2>&1; rm -f $O/trace1.bin
TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 python scripts/solve_time.py 512 64 2 2 > $O/prof64.txt 2>&1
(time python -m pytest tests -m gpu -x -q --durations=6) > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt; cat $O/ab.txt; grep prof $O/prof1.txt | tail -1; grep prof $O/prof64.txt | tail -1; cat $O/trace1.txt
