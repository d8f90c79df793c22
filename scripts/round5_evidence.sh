#!/bin/bash
# GPU box: the bench line of the final library and its rocprofv3 evidence (kernel stats of the default and the evaluate-all run; FETCH_SIZE /
# WRITE_SIZE in separate --pmc passes, no trace domains; SQ counters of both kernels; kernel stats per configuration), every step under its
# own timeout.  usage: scripts/round5_evidence.sh [outdir]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$(realpath -m ${1:-$ROOT/gpurun_out/r05}); mkdir -p $O
cd $ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_512x64.json 2> $O/bench_512x64.err; echo "bench rc $?"; tail -c 300 $O/bench_512x64.json; echo
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-source"
cd /tmp && export TMPDIR=/tmp
for TAG in r05_512x64 r05_512x64_skip0; do
  mkdir -p $O/raw
  if [ $TAG = r05_512x64_skip0 ]; then export TTCR_FSM_SKIP=0; else unset TTCR_FSM_SKIP; fi
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o $TAG -- $CMD > $O/${TAG}_run.txt 2>&1; echo "stats $TAG rc $?"
  find $O/raw -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/${TAG}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $C --output-format csv -d $O/raw -o ${TAG}_$C -- $CMD > $O/raw/run_$C.log 2>&1; echo "$C rc $?"
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
  head -4 $O/${TAG}_kernel_stats.csv | cut -c1-200
done
unset TTCR_FSM_SKIP
cd $ROOT && python scripts/pmc_to_json.py $O r05_512x64 512 64 $O/traffic.json
python scripts/pmc_to_json.py $O r05_512x64_skip0 512 64 $O/traffic_skip0.json
# SQ counters of the two kernels (two --pmc passes each)
SQCMD="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single-source"
timeout 600 bash scripts/sq_counters.sh $O/sq_counters_512x64_default.txt $SQCMD > /dev/null 2>&1
TTCR_FSM_SKIP=0 timeout 600 bash scripts/sq_counters.sh $O/sq_counters_512x64_skip0.txt $SQCMD > /dev/null 2>&1
tail -3 $O/sq_counters_512x64_default.txt; tail -3 $O/sq_counters_512x64_skip0.txt
# the other configurations: kernel stats per configuration (one process each)
cd /tmp && export TMPDIR=/tmp
for CF in S1 C2 C4 C5 W1 W8; do
  mkdir -p $O/raw_$CF
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$CF -o r05_$CF -- python $ROOT/scripts/config_one.py $CF 2 > $O/r05_${CF}_run.txt 2>&1
  find $O/raw_$CF -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/r05_${CF}_kernel_stats.csv
  rm -rf $O/raw_$CF
  grep "^$CF:" $O/r05_${CF}_run.txt | tee -a $O/configs.txt
done
cd $ROOT
