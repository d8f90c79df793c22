#!/bin/bash
# GPU box: SQ counters of the sweep kernels of a command (two --pmc passes, no trace domains).
# usage: scripts/sq_counters.sh <out.txt> <command...>
set -u
OUT=$(realpath -m $1); shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RAW=$(dirname $OUT)/raw_sq; mkdir -p $RAW
cd /tmp && export TMPDIR=/tmp
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  T=$(echo $G | cut -d' ' -f1)
  rocprofv3 --pmc $G --output-format csv -d $RAW -o sq_$T -- "$@" > /dev/null 2>&1
done
python3 - <<PY > $OUT
import csv, glob, collections
tot=collections.defaultdict(collections.Counter); n=collections.defaultdict(collections.Counter)
for f in glob.glob("$RAW/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","?")
        if "fsm_sweep" in k: tot[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k,v in tot.items():
    print(k, "dispatches", max(n[k].values()))
    for c, x in sorted(v.items()): print("   %-24s %.6g" % (c, x))
    if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_ACTIVE_INST_VALU"):
        # convention of profiles/r03/README.md (WENO stage): GRBM_GUI_ACTIVE is summed over the 8 XCDs, 1 024 SIMDs on the chip
        simd_cycles = v["GRBM_GUI_ACTIVE"] / 8 * 1024
        print("   VALU issuing %.2f of the SIMD cycles; resident waves per SIMD %.2f; SALU / VALU instructions %.2f" %
              (v["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles, v["SQ_WAVE_CYCLES"] * 4 / simd_cycles, v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"]))
PY
rm -rf $RAW
cat $OUT
