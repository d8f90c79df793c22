#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/stress
runs() { tag=$1; shift; echo "=== $tag: $@"; env $ENVX python scripts/stress_niter.py --tag $tag "$@" > gpurun_out/stress/$tag.log 2>&1; echo rc $?; grep "HIP runtime\|hipDeviceSync" gpurun_out/stress/$tag.log; grep "evaluated" gpurun_out/stress/$tag.log | awk '{print $5}' | tr '\n' ' ' | cut -c1-330; echo; tail -1 gpurun_out/stress/$tag.log | cut -c1-260; }
runs e1 --steps 60 --torch --lean --warm 5 --devsync hip
runs e2 --steps 60 --torch --lean --warm 5 --devsync hip --use-graph 2
ENVX="TTCR_FSM_MODE=1" runs e3 --steps 60 --torch --lean --warm 5 --devsync hip --use-graph 2
runs e4 --steps 40 --torch --lean --warm 5 --devsync hip --model blocks --sources 8 --fields
python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/stress/bench60.json 2> gpurun_out/stress/bench60.err; echo "bench rc $?"; tail -3 gpurun_out/stress/bench60.err; cut -c1-1500 gpurun_out/stress/bench60.json
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or baseline" 2>&1 | tail -5
