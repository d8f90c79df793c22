mkdir -p gpurun_out/r05
O=gpurun_out/r05
python -c "import ttcr_amd.build as b; print('library', b.source_hash())" > $O/hash.txt 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/pytest_gpu.txt
bash scripts/round5_evidence.sh $O 2>&1 | tail -40
{
echo "== fp64 batches of 2 / 4: library (chunks of 8 for fp64 batches) / variant with chunks of 16 everywhere"
python scripts/f64_batch_time.py 256 0 2 4; TTCR_AMD_LIB=$PWD/variants/c16all.so python scripts/f64_batch_time.py 256 0 2 4
python scripts/f64_batch_time.py 256 1 2; TTCR_AMD_LIB=$PWD/variants/c16all.so python scripts/f64_batch_time.py 256 1 2
} > $O/f64_batches.txt 2>&1
cat $O/f64_batches.txt
