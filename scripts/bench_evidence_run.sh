#!/bin/bash
# GPU-box script: the bench line and its rocprofv3 evidence (kernel stats, PMC traffic, 1-source profile) in one call,
# without the 6-minute -m gpu suite that scripts/round_end_run.sh runs first
O=gpurun_out/round_end; mkdir -p $O
python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
bash scripts/profile_run.sh r02_512x64 --steps 2 --warmup 1 --no-single-source > $O/profile.log 2>&1; tail -4 $O/profile.log
bash scripts/pmc_run.sh r02_512x64 --steps 1 --warmup 0 --no-single-source > $O/pmc.log 2>&1; tail -4 $O/pmc.log
python scripts/pmc_to_json.py gpurun_out/pmc_r02_512x64 r02_512x64 512 64 $O/traffic.json
bash scripts/profile_run.sh r02_512x1 --steps 3 --warmup 1 --sources 1 --no-single-source > $O/profile1.log 2>&1; tail -4 $O/profile1.log
