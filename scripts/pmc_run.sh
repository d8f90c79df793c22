#!/bin/bash
# Run on the GPU box: HBM traffic counters of the sweep kernel, separate passes per counter
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains with --pmc).
# usage: scripts/pmc_run.sh <tag> <bench args...>
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT -o ${TAG}_$C -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_$C.log 2>&1
  tail -1 $OUT/bench_$C.log | cut -c1-200
done
python - <<PY
import csv, glob, collections
for C in ("FETCH_SIZE","WRITE_SIZE"):
    tot=collections.Counter(); cnt=collections.Counter()
    for f in glob.glob("$OUT/**/*${TAG}_%s*counter_collection.csv"%C, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r.get("Kernel_Name","?")
            tot[k]+=float(r["Counter_Value"]); cnt[k]+=1
    with open("$OUT/${TAG}_%s_summary.csv"%C, "w") as o:
        o.write("Kernel_Name,Dispatches,Counter,Sum_KB,PerDispatch_KB\n")
        for k,v in tot.most_common(8):
            o.write('"%s",%d,%s,%.1f,%.1f\n'%(k,cnt[k],C,v,v/cnt[k]))
    for k,v in tot.most_common(3):
        print(C, k[:60], "dispatches", cnt[k], "sum", v, "per-dispatch", v/cnt[k])
PY
find $OUT -name "*counter_collection.csv" -size +3M -delete
find $OUT -name "*.db" -delete
