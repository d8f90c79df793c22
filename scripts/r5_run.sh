mkdir -p gpurun_out/r5p
O=$PWD/gpurun_out/r5p
timeout 600 bash scripts/sq_counters.sh $O/sq_counters_512x1_chunks16.txt python $PWD/scripts/lone_time.py 512 1 1 > /dev/null 2>&1
cat $O/sq_counters_512x1_chunks16.txt
timeout 600 bash scripts/sq_counters.sh $O/sq_counters_weno256x1_chunks16.txt python $PWD/scripts/config_one.py W1 1 > /dev/null 2>&1
cat $O/sq_counters_weno256x1_chunks16.txt
