#!/bin/bash
# GPU-box script run at the end of a round: full -m gpu suite, bench line, rocprofv3 kernel stats + PMC traffic of the bench command
O=gpurun_out/round_end; mkdir -p $O
(time python -m pytest tests -m gpu -x -q --durations=8) > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt
python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
bash scripts/profile_run.sh r02_512x64 --steps 2 --warmup 1 --no-single-source > $O/profile.log 2>&1; tail -12 $O/profile.log
bash scripts/pmc_run.sh r02_512x64 --steps 1 --warmup 0 --no-single-source > $O/pmc.log 2>&1; tail -8 $O/pmc.log
python scripts/pmc_to_json.py gpurun_out/pmc_r02_512x64 r02_512x64 512 64 $O/traffic.json
bash scripts/profile_run.sh r02_512x1 --steps 3 --warmup 1 --sources 1 --no-single-source > $O/profile1.log 2>&1; tail -12 $O/profile1.log
python scripts/configs_run.py > $O/configs.txt 2>&1; cat $O/configs.txt
