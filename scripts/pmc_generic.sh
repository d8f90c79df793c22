#!/bin/bash
# usage: scripts/pmc_generic.sh <tag> "<counter list>" <bench args...>   (one pass, no trace domains)
set -u
TAG=$1; CTRS=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcg_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-120
python - <<PY
import csv, glob, collections
tot=collections.defaultdict(collections.Counter); cnt=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","?")[:40]
        tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in tot.items():
    if "sweep" in k: print(k, dict(v))
PY
find $OUT -name "*counter_collection.csv" -size +3M -delete
