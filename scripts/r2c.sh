#!/bin/bash
# GPU-box script of the dynamic-dispatch experiment (round 2): timings by dispatch mode, phase profile, tests
O=gpurun_out/r2c; mkdir -p $O
for S in 1 2 4 8; do python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done > $O/dyn_default.txt 2>&1
for S in 1 2 4; do TTCR_FSM_DYN_BELOW=0 python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done > $O/static.txt 2>&1
for S in 8 16 64; do TTCR_FSM_DYN_BELOW=1000 python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done > $O/dyn_all.txt 2>&1
for S in 64; do python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done >> $O/dyn_default.txt 2>&1
for S in 1 4; do TTCR_AMD_LIB=$PWD/variants/dyn_early3.so python scripts/solve_time.py 512 $S 2 3 2>&1 | tail -1; done > $O/early3.txt 2>&1
python scripts/solve_time.py 256 1 2 3 2>&1 | tail -1 > $O/n256.txt
TTCR_FSM_DYN_BELOW=0 python scripts/solve_time.py 256 1 2 3 2>&1 | tail -1 >> $O/n256.txt
TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 TTCR_FSM_PROF_TRACE=$PWD/$O/trace1.bin python scripts/solve_time.py 512 1 2 2 > $O/prof1.txt 2>&1
python scripts/trace_analyze.py $O/trace1.bin > $O/trace1.txt 2>&1; rm -f $O/trace1.bin
(time python -m pytest tests -m gpu -x -q --durations=10) > $O/pytest.txt 2>&1
tail -14 $O/pytest.txt; cat $O/dyn_default.txt $O/static.txt $O/dyn_all.txt $O/early3.txt $O/n256.txt; cat $O/prof1.txt | grep prof | tail -1; cat $O/trace1.txt
