#!/bin/bash
O=gpurun_out/r2k; mkdir -p $O
for L in "" $PWD/variants/bar1.so $PWD/variants/bar2w3.so; do
  echo "== lib ${L:-default (barrier every 2 levels: timing only, racy)}"
  for S in 1 8 64; do TTCR_AMD_LIB=$L python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done
done > $O/ab.txt 2>&1
cat $O/ab.txt
TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 python scripts/solve_time.py 512 1 2 2 2>&1 | grep prof | tail -1
TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 python scripts/solve_time.py 512 64 2 2 2>&1 | grep prof | tail -1
