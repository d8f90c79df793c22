mkdir -p gpurun_out/r5n
O=gpurun_out/r5n
python -m pytest tests/test_lone_chunk_gpu.py -m gpu -x -q > $O/pytest_lone_chunk.txt 2>&1; tail -3 $O/pytest_lone_chunk.txt
{
python scripts/lone_time.py 512 3 1; python scripts/lone_time.py 512 2 2; python scripts/lone_time.py 512 2 4; python scripts/lone_time.py 256 3 1; python scripts/lone_time.py 256 3 4
python scripts/f64_batch_time.py 256 0 1 2 4 8; python scripts/f64_batch_time.py 256 1 1 2
} > $O/timings.txt 2>&1
sed 's/ lib=lib[a-z0-9_.]*//; s/ pair=default//' $O/timings.txt
{
echo "== scripts/fuzz_modes.py 240 s seed 63"
python scripts/fuzz_modes.py 240 63 2>&1 | tail -2
echo "== tests/test_fuzz_gpu.py with a 160 s budget (HIP path against the oracle)"
TTCR_FUZZ_SECONDS=160 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -2
} > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
rm -rf gpurun_out/r05; bash scripts/r5_run_final.sh
