#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/stress
B="python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-single-source --per-step"
runb() { tag=$1; shift; echo "=== $tag"; env "$@" timeout 600 $B $EXTRA > gpurun_out/stress/$tag.json 2> gpurun_out/stress/$tag.err; echo "rc $?"; grep -c "launches 3" gpurun_out/stress/$tag.err; grep "^step" gpurun_out/stress/$tag.err | awk '{print $6}' | tr '\n' ' '; echo; }
runb b_default A=1
EXTRA="--opt use_graph=0" runb b_nograph A=1
EXTRA="--opt pair_sources=0" runb b_nopair A=1
runb b_nosw TTCR_FSM_NO_SW=1
runb b_mode1 TTCR_FSM_MODE=1
runb b_wgs0 TTCR_FSM_WGS=0
echo "=== stress devptr"; python scripts/stress_niter.py --tag s_devptr --steps 60 --torch --devptr > gpurun_out/stress/s_devptr.log 2>&1; echo rc $?; grep "evaluated" gpurun_out/stress/s_devptr.log | awk '{print $5}' | tr '\n' ' '; tail -1 gpurun_out/stress/s_devptr.log | cut -c1-300
echo "=== stress devptr lean"; python scripts/stress_niter.py --tag s_lean --steps 60 --torch --devptr --lean > gpurun_out/stress/s_lean.log 2>&1; echo rc $?; grep "evaluated" gpurun_out/stress/s_lean.log | awk '{print $5}' | tr '\n' ' '; tail -1 gpurun_out/stress/s_lean.log | cut -c1-300
