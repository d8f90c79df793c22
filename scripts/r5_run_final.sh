mkdir -p gpurun_out/r05
O=gpurun_out/r05
python -c "import ttcr_amd.build as b; print('library', b.source_hash())" > $O/hash.txt 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/pytest_gpu.txt
bash scripts/round5_evidence.sh $O 2>&1 | tail -30
# the bench line once more, now that the PMC traffic of this library exists (bench.py reports it as roofline.traffic)
cp $O/traffic.json profiles/r05/traffic.json; cp $O/traffic_skip0.json profiles/r05/traffic_skip0.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_512x64_with_traffic.json 2> $O/bench_512x64_with_traffic.err; echo "bench rc $?"; tail -c 200 $O/bench_512x64_with_traffic.json; echo
{
echo "== scripts/fuzz_modes.py 240 s seed 66 (final library)"
python scripts/fuzz_modes.py 240 66 2>&1 | tail -2
echo "== scripts/fuzz_pairing.py 100 s"
python scripts/fuzz_pairing.py 100 2>&1 | tail -1
} > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
