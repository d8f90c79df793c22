mkdir -p gpurun_out/r5g
{
python scripts/lone_time.py 512 3 1
TTCR_FSM_LONE_CHUNK=8 python scripts/lone_time.py 512 3 1
python scripts/lone_time.py 256 3 1
python scripts/lone_time.py 128 3 1
TTCR_FSM_LONE_CHUNK=8 python scripts/lone_time.py 128 3 1
python scripts/lone_time.py 512 2 8
python scripts/weno_time.py 256 | tail -1
} > gpurun_out/r5g/lc16.txt 2>&1
sed 's/ lib=[a-z0-9_.]*//; s/ pair=default//' gpurun_out/r5g/lc16.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r5g/pytest_lc16.txt 2>&1; tail -3 gpurun_out/r5g/pytest_lc16.txt
