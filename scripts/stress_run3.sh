#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/stress
runs() { tag=$1; shift; echo "=== $tag: $@"; env $ENVX python scripts/stress_niter.py --tag $tag --steps 45 "$@" > gpurun_out/stress/$tag.log 2>&1; echo rc $?; grep "HIP runtime\|hipDeviceSync" gpurun_out/stress/$tag.log; grep "evaluated" gpurun_out/stress/$tag.log | awk '{print $5}' | tr '\n' ' ' | cut -c1-400; echo; }
runs r1 --lean --warm 5 --devsync hip
runs r2 --lean --warm 5 --devsync none
runs r3 --torch --lean --warm 5 --devsync hip
runs r4 --torch --lean --warm 5 --devsync none
runs r5 --lean --warm 5 --devsync hip --use-graph 0
ENVX="TTCR_FSM_MODE=0" runs r6 --lean --warm 5 --devsync hip --sources 8
