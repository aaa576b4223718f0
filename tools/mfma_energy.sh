cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r06_mfma_energy.txt; : > $O
for rep in 1 2; do
for v in 0 1 2 3; do
  python tools/clock_watch.py -- tools/micro/bin/mfma_energy $v 1500 >> $O 2>&1
done
done
cat $O
