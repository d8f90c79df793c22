#!/bin/bash
# GPU box: the bench line of the final library and its rocprofv3 evidence (kernel stats; FETCH_SIZE / WRITE_SIZE in separate --pmc
# passes, no trace domains), every step under its own timeout.  usage: scripts/round4_evidence.sh [outdir]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$(realpath -m ${1:-$ROOT/gpurun_out/r04}); mkdir -p $O
cd $ROOT
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_512x64.json 2> $O/bench_512x64.err; echo "bench rc $?"; tail -c 400 $O/bench_512x64.json; echo
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-source"
cd /tmp && export TMPDIR=/tmp
TAG=r04_512x64
mkdir -p $O/raw
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o $TAG -- $CMD > $O/${TAG}_run.txt 2>&1; echo "stats rc $?"
find $O/raw -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --output-format csv -d $O/raw -o ${TAG}_$C -- $CMD > $O/raw/run_$C.log 2>&1; echo "$C rc $?"
done
python3 - <<PY
import csv, glob, collections
for C in ("FETCH_SIZE","WRITE_SIZE"):
    tot=collections.Counter(); cnt=collections.Counter()
    for f in glob.glob("$O/raw/**/*${TAG}_%s*counter_collection.csv"%C, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r.get("Kernel_Name","?"); tot[k]+=float(r["Counter_Value"]); cnt[k]+=1
    with open("$O/${TAG}_%s_summary.csv"%C, "w") as o:
        o.write("Kernel_Name,Dispatches,Counter,Sum_KB,PerDispatch_KB\n")
        for k,v in tot.most_common(6): o.write('"%s",%d,%s,%.1f,%.1f\n'%(k,cnt[k],C,v,v/cnt[k]))
PY
rm -rf $O/raw
cd $ROOT && python scripts/pmc_to_json.py $O $TAG 512 64 $O/traffic.json
head -5 $O/${TAG}_kernel_stats.csv | cut -c1-220
# the other configurations: kernel stats per configuration (one process each)
cd /tmp && export TMPDIR=/tmp
for CF in S1 C2 C4 C5 W1 W8; do
  mkdir -p $O/raw_$CF
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$CF -o r04_$CF -- python $ROOT/scripts/config_one.py $CF 2 > $O/r04_${CF}_run.txt 2>&1
  find $O/raw_$CF -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/r04_${CF}_kernel_stats.csv
  rm -rf $O/raw_$CF
  grep "^$CF:" $O/r04_${CF}_run.txt | tee -a $O/configs.txt
done
cd $ROOT
