#!/bin/bash
# GPU box: SQ / TCC counter passes of the bench command (one rocprofv3 --pmc pass per group, no trace domains),
# sums per sweep kernel printed by scripts/pmc_generic.sh.  usage: scripts/counters_run.sh [bench args]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$ROOT}
ARGS="--steps 1 --warmup 0 --no-single-source $*"
bash $ROOT/scripts/pmc_generic.sh q1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" $ARGS
bash $ROOT/scripts/pmc_generic.sh q2 "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" $ARGS
bash $ROOT/scripts/pmc_generic.sh q3 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $ARGS
bash $ROOT/scripts/pmc_generic.sh t1 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $ARGS
bash $ROOT/scripts/pmc_generic.sh t3 "TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $ARGS
