mkdir -p gpurun_out/r05
O=gpurun_out/r05
python -c "import ttcr_amd.build as b; print('library', b.source_hash())" > $O/hash.txt 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/pytest_gpu.txt
bash scripts/round5_evidence.sh $O 2>&1 | tail -40
