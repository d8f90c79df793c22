#!/bin/bash
# GPU box: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate, no trace domains) of any command.
# usage: scripts/prof_cmd.sh <tag> <outdir> <command...>      -> <outdir>/<tag>_kernel_stats.csv, <tag>_{FETCH,WRITE}_SIZE_summary.csv
set -u
TAG=$1; OUT=$(realpath -m $2); shift; shift
mkdir -p $OUT/raw_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_$TAG -o $TAG -- "$@" > $OUT/${TAG}_run.txt 2>&1
find $OUT/raw_$TAG -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/raw_$TAG -o ${TAG}_$C -- "$@" > $OUT/raw_$TAG/run_$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for C in ("FETCH_SIZE","WRITE_SIZE"):
    tot=collections.Counter(); cnt=collections.Counter()
    for f in glob.glob("$OUT/raw_$TAG/**/*${TAG}_%s*counter_collection.csv"%C, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r.get("Kernel_Name","?"); tot[k]+=float(r["Counter_Value"]); cnt[k]+=1
    with open("$OUT/${TAG}_%s_summary.csv"%C, "w") as o:
        o.write("Kernel_Name,Dispatches,Counter,Sum_KB,PerDispatch_KB\n")
        for k,v in tot.most_common(6): o.write('"%s",%d,%s,%.1f,%.1f\n'%(k,cnt[k],C,v,v/cnt[k]))
PY
rm -rf $OUT/raw_$TAG
tail -2 $OUT/${TAG}_run.txt; head -4 $OUT/${TAG}_kernel_stats.csv | cut -c1-200
