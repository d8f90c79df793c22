mkdir -p gpurun_out/r5g
{
for lib in "" variants/late5.so variants/late7.so; do
  if [ -n "$lib" ]; then export TTCR_AMD_LIB=$PWD/$lib; fi
  echo "== ${lib:-library}"
  python scripts/lone_time.py 512 3 1
  python scripts/lone_time.py 256 3 1
  TTCR_FSM_SKIP=0 python scripts/lone_time.py 512 2 8
done
} > gpurun_out/r5g/late.txt 2>&1
sed 's/ lib=[a-z0-9_.]*//; s/ pair=default//' gpurun_out/r5g/late.txt
