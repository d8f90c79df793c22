#!/bin/bash
# repeat-stress of the 512^3 x 64 batch under the default driver and with parts of it switched off (bisecting)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/stress
S=${STEPS:-100}
run() { tag=$1; shift; echo "=== $tag"; env "$@" timeout 900 python scripts/stress_niter.py --tag $tag --steps $S $EXTRA > gpurun_out/stress/$tag.log 2>&1; echo "rc $?"; grep -v "^\[$tag\] step [0-9]*: evaluated" gpurun_out/stress/$tag.log | cut -c1-600 | tail -${TAILN:-8}; }
EXTRA="--torch" run t_default A=1
EXTRA="--torch --use-graph 0" run t_nograph A=1
EXTRA="--torch --skip 0" run t_skip0 A=1
EXTRA="--torch" run t_nosw TTCR_FSM_NO_SW=1
EXTRA="--torch" run t_mode1 TTCR_FSM_MODE=1
run plain A=1
