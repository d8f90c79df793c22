mkdir -p gpurun_out/r5f
python -m pytest tests -m gpu -x -q > gpurun_out/r5f/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r5f/pytest_gpu.txt
