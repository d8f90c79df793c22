#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of a bench configuration.
# usage: scripts/profile_run.sh <tag> <bench args...>
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -r head -8
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete; ls -R $OUT | head -20
