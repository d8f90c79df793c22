#!/bin/bash
# GPU-box script: early-publish / PRE knobs for few sources, then the whole -m gpu suite
O=gpurun_out/r2g; mkdir -p $O
for L in "" $PWD/variants/early3.so $PWD/variants/earlyall.so; do
  echo "== lib ${L:-default}"
  for S in 1 2 4 8 64; do TTCR_AMD_LIB=$L python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done
done > $O/early.txt 2>&1
echo "== PRE from 1 group" >> $O/early.txt
for S in 1 2 4; do TTCR_FSM_PRE_MIN=1 python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done >> $O/early.txt 2>&1
echo "== PRE from 1 group + early3" >> $O/early.txt
for S in 1 2 4; do TTCR_AMD_LIB=$PWD/variants/early3.so TTCR_FSM_PRE_MIN=1 python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done >> $O/early.txt 2>&1
cat $O/early.txt
(time python -m pytest tests -m gpu -q --durations=6) > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt
