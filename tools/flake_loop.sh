cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/flake
for i in 1 2 3 4 5 6; do
  python -m pytest tests/test_dist_gpu.py -m gpu -q -x -k 'not full and not gcnmain' > gpurun_out/flake/run$i.txt 2>&1
  grep -E "passed|failed" gpurun_out/flake/run$i.txt | tail -1
done
