mkdir -p gpurun_out/r05b
O=gpurun_out/r05b
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_512x64.json 2> $O/bench_512x64.err; echo "bench rc $?"; tail -c 200 $O/bench_512x64.json; echo
{
echo "== scripts/fuzz_modes.py 420 s seed 64 (final library)"
python scripts/fuzz_modes.py 420 64 2>&1 | tail -2
echo "== scripts/fuzz_modes.py 200 s seed 65 with TTCR_FSM_PREFILL=1 (second set of fields on every grid)"
TTCR_FSM_PREFILL=1 python scripts/fuzz_modes.py 200 65 2>&1 | tail -2
echo "== scripts/fuzz_pairing.py 160 s"
python scripts/fuzz_pairing.py 160 2>&1 | tail -1
echo "== scripts/piped_check.py 24 cases seed 78 (pipelined kernel against the default kernel)"
python scripts/piped_check.py --cases 24 --no-time --seed 78 2>&1 | tail -1
} > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
