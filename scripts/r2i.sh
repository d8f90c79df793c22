#!/bin/bash
O=gpurun_out/r2i; mkdir -p $O
for L in prof prof_nosync; do echo "== $L"; TTCR_AMD_LIB=$PWD/variants/$L.so TTCR_FSM_PROF=1 python scripts/solve_time.py 512 1 2 2 2>&1 | tail -3; TTCR_AMD_LIB=$PWD/variants/$L.so TTCR_FSM_PROF=1 python scripts/solve_time.py 512 64 2 2 2>&1 | tail -3; done > $O/nosync.txt 2>&1
cat $O/nosync.txt
(time python -m pytest tests -m gpu -q --durations=6) > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
