mkdir -p gpurun_out/r5l
O=gpurun_out/r5l
python -c "import ttcr_amd.build as b; print('library', b.source_hash())" > $O/hash.txt 2>&1
python -m pytest tests/test_lone_chunk_gpu.py -m gpu -x -q > $O/pytest_lone_chunk.txt 2>&1; tail -3 $O/pytest_lone_chunk.txt
{
python scripts/lone_time.py 512 3 1; python scripts/lone_time.py 256 3 1
python scripts/weno_time.py 256 | tail -1; python scripts/weno_batch.py 256 2; python scripts/weno_batch.py 256 4
python scripts/f64_time.py; python scripts/weno_time.py 256 f64 | tail -1
} > $O/timings.txt 2>&1
sed 's/ lib=lib[a-z0-9_.]*//; s/ pair=default//' $O/timings.txt
{
echo "== scripts/fuzz_modes.py 300 s seed 62 (driver modes x exact skipping, pairs forced at random, fp32 and fp64, WENO on / off; whole-iteration launches on chunks of 16 levels where the library uses them, per-sweep and tile launches on chunks of 8)"
python scripts/fuzz_modes.py 300 62 2>&1 | tail -2
echo "== tests/test_fuzz_gpu.py with a 200 s budget (HIP path against the oracle)"
TTCR_FUZZ_SECONDS=200 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -2
} > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
