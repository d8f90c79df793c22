#!/bin/bash
# one rocprofv3 --pmc pass of an arbitrary command (no trace domains): scripts/pmc_cmd.sh "<counters>" <command...>; sums per sweep kernel
set -u
CTRS=$1; shift
RAW=$(mktemp -d /tmp/pmcXXXX)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $RAW -o p -- "$@" > $RAW/log 2>&1
tail -1 $RAW/log | cut -c1-200
python3 - <<PY
import csv, glob, collections
tot=collections.defaultdict(collections.Counter)
for f in glob.glob("$RAW/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","?")
        if "sweep" in k: tot[k[:60]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in tot.items(): print(k, {a: "%.4g" % b for a, b in v.items()})
PY
rm -rf $RAW
