# the split-bf16 A^T.B micro-benchmark's variants in `loop` mode under clock_watch.py (ordinals of tools/micro/x3_tn.hip, N = 600 section)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_x3_tn_energy.txt; : > $O
for rep in 1 2; do
for ord in 0 1 7 8 9; do
  python tools/clock_watch.py -- tools/micro/bin/x3_tn loop $ord 2>&1 | grep -v "^M = " >> $O
done
done
cat $O
