rm -rf gpurun_out/r05; bash scripts/r5_run_final.sh
