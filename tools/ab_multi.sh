# same-box A/B of several builds of the library (run through gpurun from the repo root):
#   bash tools/ab_multi.sh <out-name> <reps> "<lib1> <lib2> ..." <bench args...>
# every lib is a path to a libgeogcn.so (the in-tree one: geographconv_amd/libgeogcn.so); per run: ms per step + the in-step kernel medians
cd $GRAFT_REPO_ROOT
NAME=$1; REPS=$2; LIBS=$3; shift 3
O=$GRAFT_REPO_ROOT/gpurun_out/$NAME.txt
: > $O
cp geographconv_amd/libgeogcn.so /tmp/intree.so
for rep in $(seq 1 $REPS); do
  for lib in $LIBS; do
    if [ $lib != geographconv_amd/libgeogcn.so ]; then cp $lib geographconv_amd/libgeogcn.so; else cp /tmp/intree.so geographconv_amd/libgeogcn.so; fi
    echo "== $lib: bench.py $*" >> $O
    timeout 600 python bench.py "$@" --cpu-sample none --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('  ms_per_step %.3f  median %.3f' % (d['ms_per_step'], d['step_ms']['median']))
for o in d['roofline'].get('others',[]): print('    %-70s %.3f' % (o['kernel'][:70], o['ms']))
print('    %-70s %.3f' % ('plain graph product (roofline kernel)', d['roofline']['avg_launch_ms']))
" >> $O
  done
done
cp /tmp/intree.so geographconv_amd/libgeogcn.so
cat $O
