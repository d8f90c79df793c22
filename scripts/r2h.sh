#!/bin/bash
O=gpurun_out/r2h; mkdir -p $O
TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 TTCR_FSM_PROF_TRACE=$PWD/$O/trace1.bin python scripts/solve_time.py 512 1 2 2 > $O/prof1.txt 2>&1
python scripts/trace_analyze.py $O/trace1.bin > $O/trace1.txt 2>&1
python scripts/chunk_trace.py $O/trace1.bin 4 5 5 > $O/chunks_dir4.txt 2>&1
python scripts/chunk_trace.py $O/trace1.bin 0 5 5 > $O/chunks_dir0.txt 2>&1
TTCR_FSM_MODE=1 TTCR_AMD_LIB=$PWD/variants/prof.so TTCR_FSM_PROF=1 TTCR_FSM_PROF_TRACE=$PWD/$O/trace1m1.bin python scripts/solve_time.py 512 1 2 2 > $O/prof1m1.txt 2>&1
python scripts/chunk_trace.py $O/trace1m1.bin 7 5 5 > $O/chunks_m1.txt 2>&1
rm -f $O/*.bin
for K in 0 1; do echo "== skip $K"; TTCR_FSM_SKIP=$K python scripts/weno_batch.py 256 1 2>&1 | tail -1; TTCR_FSM_SKIP=$K python scripts/weno_batch.py 256 8 2>&1 | tail -1; done > $O/weno_skip.txt 2>&1
(time python -m pytest tests/test_fuzz_gpu.py tests/test_integration.py tests/test_parity_gpu.py -m gpu -q -k "random_configurations or adapter or receivers_next or device_views or more_sources" ) > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt; cat $O/weno_skip.txt; grep prof $O/prof1.txt | tail -1; cat $O/chunks_dir4.txt $O/chunks_m1.txt | cut -c1-700
