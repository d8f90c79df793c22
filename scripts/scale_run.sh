#!/bin/bash
# The scaling curve of the workload on ONE node with up to 8 MI355X, one process per GPU over RCCL/xGMI -- the first multi-GPU lease is one command:
#   scripts/scale_run.sh [outdir] [steps] [warmup]
# strong scaling: the 64 sources of BASELINE.json's config 3 block-distributed over 1 / 2 / 4 / 8 ranks (ttcr/Grid3D.h:451-465, 810-853);
# weak scaling: 8 sources per rank.  Every run prints bench.py's JSON line (value = whole-job Mnodes/s per sweep-iteration, sources_per_s, the
# collective backend and the device of every rank); the table at the end is sources/s and the efficiency against N x the one-GPU figure.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$(realpath -m ${1:-$ROOT/gpurun_out/scale}); STEPS=${2:-10}; WARM=${3:-2}
mkdir -p $O; cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
PORT=29611
for MODE in strong weak; do
  for N in 1 2 4 8; do
    [ $N -gt $NGPU ] && { echo "skip N=$N: $NGPU device(s) visible"; continue; }
    EXTRA="--no-cpu-baseline --no-single-source"; [ $MODE = weak ] && EXTRA="$EXTRA --sources-per-gpu 8"
    PORT=$((PORT + 1))
    if [ $N -eq 1 ]; then
      timeout 1800 python bench.py --gpus 1 --steps $STEPS --warmup $WARM --force-dist $EXTRA > $O/${MODE}_$N.json 2> $O/${MODE}_$N.err
    else
      timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $N --steps $STEPS --warmup $WARM $EXTRA > $O/${MODE}_$N.json 2> $O/${MODE}_$N.err
    fi
    echo "$MODE N=$N rc $?"
  done
done
python3 - <<PY
import json, glob, os
for mode in ("strong", "weak"):
    rows = []
    for n in (1, 2, 4, 8):
        p = os.path.join("$O", f"{mode}_{n}.json")
        if not os.path.exists(p): continue
        ls = [l for l in open(p) if l.startswith("{")]
        if not ls: continue
        d = json.loads(ls[-1]); rows.append((n, d["sources_per_s"], d["value"], d["ms_per_step"], d["config"]["collective_backend"]))
    if not rows: continue
    base = rows[0][1] / rows[0][0]
    print(f"{mode} scaling: N, sources/s, Mnodes/s per sweep-iteration, ms per step, backend, efficiency vs N x one GPU")
    for n, sps, v, ms, be in rows: print(f"  {n}  {sps:9.2f}  {v:10.1f}  {ms:8.2f}  {be}  {sps / (n * base):.3f}")
PY
