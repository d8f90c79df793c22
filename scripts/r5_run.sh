mkdir -p gpurun_out/r5g
{
for lib in "" variants/skip1w4.so; do
  if [ -n "$lib" ]; then export TTCR_AMD_LIB=$PWD/$lib; fi
  echo "== ${lib:-library}"
  TTCR_FSM_PAIR=0 python scripts/lone_time.py 512 3 8
  TTCR_FSM_PAIR=0 python scripts/lone_time.py 512 3 4
  TTCR_FSM_PAIR=0 python scripts/lone_time.py 512 3 2
  TTCR_FSM_SKIP=1 python scripts/lone_time.py 512 3 1
  python scripts/config_one.py C4 2 | grep "^C4"
done
} > gpurun_out/r5g/skip1w4.txt 2>&1
sed 's/ lib=[a-z0-9_.]*//' gpurun_out/r5g/skip1w4.txt
