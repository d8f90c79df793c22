mkdir -p gpurun_out/r5b
export TTCR_FSM_SLAB=1
{
python scripts/slab_time.py 512 3
TTCR_AMD_LIB=$PWD/variants/pub0.so python scripts/slab_time.py 512 3
export TTCR_FSM_PROF=1
for v in prof prof_pub0; do TTCR_AMD_LIB=$PWD/variants/$v.so timeout 120 python scripts/slab_time.py 512 2; done
} > gpurun_out/r5b/pub.txt 2>&1
cat gpurun_out/r5b/pub.txt
