cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05d
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05d/smoke.txt 2>&1; echo "smoke rc $?" >> gpurun_out/r05d/smoke.txt
( time python bench.py ) > gpurun_out/r05d/bench_default.txt 2>&1
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r05d/gpu_suite.txt 2>&1
tail -3 gpurun_out/r05d/smoke.txt; tail -5 gpurun_out/r05d/bench_default.txt | cut -c1-400; tail -8 gpurun_out/r05d/gpu_suite.txt
