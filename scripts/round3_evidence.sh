#!/bin/bash
# GPU box, round 3: the bench line and everything profiles/r03 cites, in one call (library = the committed sources).
# usage: scripts/round3_evidence.sh   -> gpurun_out/r3_ev/
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$ROOT}
O=$ROOT/gpurun_out/r3_ev; mkdir -p $O; cd $ROOT
python bench.py --steps 5 --warmup 1 > $O/bench_512x64.json 2> $O/bench.err; tail -c 300 $O/bench_512x64.json; echo
# the evaluate-all kernel on the same workload (exact skipping off): what the roofline of the kernel itself is
TTCR_FSM_SKIP=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-source > $O/bench_512x64_skip0.json 2>> $O/bench.err
# kernel stats + PMC traffic: default run, skip off, one source
bash scripts/prof_cmd.sh r03_512x64 $O python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-source
TTCR_FSM_SKIP=0 bash scripts/prof_cmd.sh r03_512x64_skip0 $O python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-source
bash scripts/prof_cmd.sh r03_512x1 $O python $ROOT/bench.py --steps 3 --warmup 1 --sources 1 --no-cpu-baseline --no-single-source
python - <<PY
import csv, json, os, sys
sys.path.insert(0, "$ROOT")
from ttcr_amd.build import source_hash
O = "$O"
def per(tag, c, key):
    for r in csv.DictReader(open(os.path.join(O, f"{tag}_{c}_summary.csv"))):
        if key in r["Kernel_Name"]: return float(r["PerDispatch_KB"]), r["Kernel_Name"]
    return None, None
for tag, name in (("r03_512x64", "traffic.json"), ("r03_512x64_skip0", "traffic_skip0.json")):
    f, k = per(tag, "FETCH_SIZE", "fsm_sweep_persistent"); w, _ = per(tag, "WRITE_SIZE", "fsm_sweep_persistent")
    cf, _ = per(tag, "FETCH_SIZE", "fsm_shear_slowness"); cw, _ = per(tag, "WRITE_SIZE", "fsm_shear_slowness")
    rec = {"source_hash": source_hash(), "size": 512, "sources": 64, "kernel": k, "fetch_kb_per_launch": f, "write_kb_per_launch": w,
           "calibration": {"kernel": "fsm_shear_slowness_lines", "bytes_read_per_call": 512 ** 3 * 4, "FETCH_SIZE_kb_reported": cf,
                           "bytes_written_per_call": 512 ** 3 * 4, "WRITE_SIZE_kb_reported": cw},
           "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
           "note": "gfx950: FETCH_SIZE counts 128-B requests at 64 B (x2 correction, see the calibration kernel of the same run); per launch = "
                   "average over the launches of the run (with exact skipping the launches of a solve differ: the first sweep-iteration evaluates "
                   "about two thirds of the node updates, the second a sixth)"}
    json.dump(rec, open(os.path.join(O, name), "w"), indent=1)
PY
# the other configurations: one line each, then kernel stats + traffic per configuration
python scripts/configs_run.py > $O/configs.txt 2>&1
for C in C2 C4 C5 W1 W8; do bash scripts/prof_cmd.sh r03_$C $O python $ROOT/scripts/config_one.py $C 2; done
# WENO stage: SQ counters of the lone 256^3 source
cd /tmp && export TMPDIR=/tmp
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  T=$(echo $G | cut -d' ' -f1)
  rocprofv3 --pmc $G --output-format csv -d $O/raw_w -o w1_$T -- python $ROOT/scripts/config_one.py W1 1 > /dev/null 2>&1
done
python - <<PY > $O/weno_counters.txt
import csv, glob, collections
tot=collections.defaultdict(collections.Counter); n=collections.Counter()
for f in glob.glob("$O/raw_w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","?")
        if "fsm_sweep_persistent" in k: tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in tot.items(): print(k); [print("   %-24s %.6g" % (c, x)) for c, x in sorted(v.items())]
PY
rm -rf $O/raw_w; cd $ROOT
# exact skipping off / on over batch sizes; the floor experiments of the lone source (timing builds under variants/)
(python scripts/skip_sweep.py 512 1,2,4,8,16,32,64; python scripts/skip_sweep.py 256 1,8,16,64; python scripts/skip_sweep.py 256 1,8 1) > $O/skip_sweep.txt 2>&1
(for L in nowait noexch nowait_noexch; do [ -f variants/$L.so ] || continue; echo "== $L (TIMING build, wrong results)"; TTCR_AMD_LIB=$ROOT/variants/$L.so python scripts/solve_time.py 512 1 2 3 | tail -1;
   TTCR_FSM_XS_LDS=0 TTCR_AMD_LIB=$ROOT/variants/$L.so python scripts/solve_time.py 512 1 2 3 | tail -1; done; echo "== the library"; python scripts/solve_time.py 512 1 2 3 | tail -1) > $O/lone_source_floor.txt 2>&1
python scripts/set_slowness_time.py 512 > $O/set_slowness.txt 2>&1
ls $O
