# same-box A/B of several builds of the library on ONE kernel's in-step time (run through gpurun from the repo root):
#   bash tools/ab_kernel.sh <out-name> <reps> "<lib1> <lib2> ..." <kernel-name regex> [bench args...]
# per run: ms per step of bench.py (under the tracer) + the rocprofv3 kernel-trace rows whose name matches
cd $GRAFT_REPO_ROOT
NAME=$1; REPS=$2; LIBS=$3; PAT=$4; shift 4
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out/$NAME.txt
: > $O
cp geographconv_amd/libgeogcn.so /tmp/intree.so
cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for lib in $LIBS; do
    if [ $lib != geographconv_amd/libgeogcn.so ]; then cp $GRAFT_REPO_ROOT/$lib $GRAFT_REPO_ROOT/geographconv_amd/libgeogcn.so; else cp /tmp/intree.so $GRAFT_REPO_ROOT/geographconv_amd/libgeogcn.so; fi
    rm -rf /tmp/tr
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o b -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample none --no-extras --steps 10 --warmup 3 "$@" > /tmp/b.json 2>/dev/null
    echo "== $lib" >> $O
    python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('  ms_per_step (traced) %.3f' % d['ms_per_step'])" >> $O
    python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/tr/b_kernel_trace.csv | grep -E "$PAT" | cut -c1-130 >> $O
  done
done
cp /tmp/intree.so $GRAFT_REPO_ROOT/geographconv_amd/libgeogcn.so
cat $O
