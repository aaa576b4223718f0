cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_x3_energy2.txt; : > $O
for rep in 1 2; do
for ord in 53 89 97 101 102 105 90 98 103 104; do
  python tools/clock_watch.py -- tools/micro/bin/x3_rows loop $ord 2>&1 | grep -v "^N = " | grep -v "power W median [2-7][0-9][0-9] max [2-9][0-9][0-9] " >> $O
done; done
cat $O
