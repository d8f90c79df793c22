mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05/pytest_gpu_full.txt 2>&1; tail -3 gpurun_out/r05/pytest_gpu_full.txt
bash scripts/round5_evidence.sh gpurun_out/r05 > gpurun_out/r05/evidence_log.txt 2>&1; tail -25 gpurun_out/r05/evidence_log.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.txt 2>&1; tail -2 gpurun_out/r05/smoke.txt
