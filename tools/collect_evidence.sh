#!/bin/bash
# Round evidence on the GPU box (run through gpurun from the repo root): the default bench line, its kernel trace, and the
# PMC passes of the dominant kernel (separate passes, csv) -> gpurun_out/evidence/.  Summaries are made afterwards with
# tools/rocprof_summary.py and tools/pmc_summary.py and copied to profiles/.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/evidence
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample none --steps 10 --warmup 3 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o pmc_$n -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample none --steps 3 --warmup 1 > /dev/null 2>&1
done
ls -R $OUT | head -40
