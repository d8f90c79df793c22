#!/bin/bash
# GPU box: the bench line of the final library and its rocprofv3 evidence.  Kernel stats of the bench command (default and arith = 1); FETCH_SIZE /
# WRITE_SIZE in separate --pmc passes (no trace domains) of the bench command and of one process per configuration -- the lone 512^3 source in
# both arithmetic modes, 8 sources, the WENO stage, C5 -- each with its calibration copy in the same run (scripts/pmc_to_json.py); kernel stats per
# configuration.  Every step under its own timeout.  usage: scripts/round6_evidence.sh [outdir]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$(realpath -m ${1:-$ROOT/gpurun_out/r06}); mkdir -p $O
cd $ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_512x64.json 2> $O/bench_512x64.err; echo "bench rc $?"; tail -c 200 $O/bench_512x64.json; echo
summarise() {   # <raw dir> <tag>: per-kernel sums of the two counter passes -> $O/<tag>_{FETCH,WRITE}_SIZE_summary.csv
  python3 - <<PY
import csv, glob, collections
for C in ("FETCH_SIZE","WRITE_SIZE"):
    tot=collections.Counter(); cnt=collections.Counter()
    for f in glob.glob("$1/**/*$2_%s*counter_collection.csv"%C, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r.get("Kernel_Name","?"); tot[k]+=float(r["Counter_Value"]); cnt[k]+=1
    with open("$O/$2_%s_summary.csv"%C, "w") as o:
        o.write("Kernel_Name,Dispatches,Counter,Sum_KB,PerDispatch_KB\n")
        for k,v in tot.most_common(16): o.write('"%s",%d,%s,%.1f,%.1f\n'%(k,cnt[k],C,v,v/cnt[k]))
PY
}
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-source"
for TAG in r06_512x64 r06_512x64_arith1; do
  mkdir -p $O/raw
  CMD="$BENCH"; [ $TAG = r06_512x64_arith1 ] && CMD="$BENCH --opt arith=1"
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o $TAG -- $CMD > $O/${TAG}_run.txt 2>&1; echo "stats $TAG rc $?"
  find $O/raw -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/${TAG}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $C --output-format csv -d $O/raw -o ${TAG}_$C -- $CMD > $O/raw/run_$C.log 2>&1; echo "$C rc $?"
  done
  summarise $O/raw $TAG
  rm -rf $O/raw
  head -3 $O/${TAG}_kernel_stats.csv | cut -c1-200
done
(cd $ROOT && python scripts/pmc_to_json.py $O r06_512x64 512 64 $O/traffic.json > /dev/null && python scripts/pmc_to_json.py $O r06_512x64_arith1 512 64 $O/traffic_arith1.json > /dev/null)
# one process per configuration: kernel stats + the two counter passes
for CF in S1 S1a S1d S1da E8 E8a C2 C4 C5 W1 W8; do
  mkdir -p $O/raw_$CF
  C1=${CF%a}; if [ $CF != $C1 ]; then export TTCR_FSM_ARITH=1; else unset TTCR_FSM_ARITH; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$CF -o r06_$CF -- python $ROOT/scripts/config_one.py $C1 2 > $O/r06_${CF}_run.txt 2>&1
  find $O/raw_$CF -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/r06_${CF}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/raw_$CF -o r06_${CF}_$C -- python $ROOT/scripts/config_one.py $C1 2 > /dev/null 2>&1
  done
  summarise $O/raw_$CF r06_$CF
  rm -rf $O/raw_$CF
  case $C1 in S1|S1d|E8) SZ=512;; C5) SZ=4096;; *) SZ=256;; esac
  case $C1 in E8|C4|W8) NS=8;; C5) NS=16;; *) NS=1;; esac
  (cd $ROOT && python scripts/pmc_to_json.py $O r06_$CF $SZ $NS $O/traffic_$CF.json > /dev/null 2>&1)
  grep "^$C1:" $O/r06_${CF}_run.txt | sed "s/^$C1:/$CF:/" | tee -a $O/configs.txt
done
unset TTCR_FSM_ARITH
cd $ROOT
