mkdir -p gpurun_out/r5g
{
echo "== scripts/fuzz_modes.py 420 s seed 55 (driver modes x exact skipping, pairs forced at random)"; timeout 600 python scripts/fuzz_modes.py 420 55 2>&1 | tail -2
echo "== scripts/fuzz_modes.py 240 s seed 56 with TTCR_FSM_PREFILL=1 (second set of fields on every grid)"; TTCR_FSM_PREFILL=1 timeout 400 python scripts/fuzz_modes.py 240 56 2>&1 | tail -2
echo "== scripts/fuzz_pairing.py 200 s"; timeout 400 python scripts/fuzz_pairing.py 200 7 2>&1 | tail -2
echo "== tests/test_fuzz_gpu.py with a 400 s budget (HIP path against the oracle)"; TTCR_FUZZ_SECONDS=400 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -2
echo "== scripts/piped_check.py 40 cases seed 77 (pipelined kernel against the default kernel)"; timeout 900 python scripts/piped_check.py --cases 40 --seed 77 --no-time 2>&1 | tail -1
} > gpurun_out/r5g/fuzz.txt 2>&1
cat gpurun_out/r5g/fuzz.txt
