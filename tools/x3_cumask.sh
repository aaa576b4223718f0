# is the split-bf16 whole-rows kernel bound by the socket's power cap?  the same launch on all 256 CUs and on 128 / 64 of them (CU-masked stream, evenly
# over the XCDs, grid scaled): if the cap is what limits the full chip, the part-chip runs keep a higher clock and take LESS than 2x / 4x the time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_x3_cumask.txt; : > $O
for rep in 1 2; do
for ord in 44 49 50; do
for ncu in 256 128 64; do
  python tools/clock_watch.py -- tools/micro/bin/x3_rows loop $ord $ncu 2>&1 | grep -v "^N = " >> $O
done; done; done
cat $O
