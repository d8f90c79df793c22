#!/bin/bash
# GPU-box script: ticket order by expected start time vs sweep by sweep (round 2)
O=gpurun_out/r2d; mkdir -p $O
for S in 1 2 4 8; do python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done > $O/default.txt 2>&1
for S in 1 2 4; do TTCR_FSM_TIME_ORDER_BELOW=0 python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done > $O/sweep_major.txt 2>&1
for S in 8 16 64; do TTCR_FSM_TIME_ORDER_BELOW=1000 python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done > $O/time_major_all.txt 2>&1
python scripts/solve_time.py 512 64 2 3 2>&1 | tail -1 >> $O/default.txt
python scripts/solve_time.py 256 1 2 3 2>&1 | tail -1 > $O/n256.txt
TTCR_FSM_TIME_ORDER_BELOW=0 python scripts/solve_time.py 256 1 2 3 2>&1 | tail -1 >> $O/n256.txt
TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 TTCR_FSM_PROF_TRACE=$PWD/$O/trace1.bin python scripts/solve_time.py 512 1 2 2 > $O/prof1.txt 2>&1
python scripts/trace_analyze.py $O/trace1.bin > $O/trace1.txt 2>&1; rm -f $O/trace1.bin
(time python -m pytest tests -m gpu -x -q --durations=10) > $O/pytest.txt 2>&1
tail -14 $O/pytest.txt; cat $O/default.txt $O/sweep_major.txt $O/time_major_all.txt $O/n256.txt; cat $O/prof1.txt | grep prof | tail -1; cat $O/trace1.txt
