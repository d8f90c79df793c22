mkdir -p gpurun_out/r5o
O=gpurun_out/r5o
{
echo "== the 64 sources of the bench, two sweep-iterations (scripts/lone_skip.py 512 64, ITERS=2): host-side tuning switches of the final library, one box"
ITERS=2 python scripts/lone_skip.py 512 64 | sed "s/^/default: /"
TTCR_FSM_PRE_MIN=99 ITERS=2 python scripts/lone_skip.py 512 64 | sed "s/^/no sampled counters (PRE): /"
TTCR_FSM_WGS=768 ITERS=2 python scripts/lone_skip.py 512 64 | sed "s/^/768 workgroups: /"
TTCR_FSM_WGS=1024 ITERS=2 python scripts/lone_skip.py 512 64 | sed "s/^/1024 workgroups: /"
TTCR_FSM_TIME_ORDER_BELOW=64 ITERS=2 python scripts/lone_skip.py 512 64 | sed "s/^/units in start-time order: /"
ITERS=2 python scripts/lone_skip.py 512 64 | sed "s/^/default again: /"
} > $O/headline_switches.txt 2>&1
cat $O/headline_switches.txt
