cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_x3_energy.txt; : > $O
for rep in 1 2; do
for ord in 44 80 48 51 55 49 50 81 82 83 54; do
  python tools/clock_watch.py -- tools/micro/bin/x3_rows loop $ord 2>&1 | grep -v "^N = " >> $O
done
done
cat $O
