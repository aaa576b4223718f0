cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_fuse_gemms_ab.txt; : > $O
for rep in 1 2 3; do
for v in "" "--set FUSE_GEMMS=0"; do
  echo "== bench.py $v" >> $O
  python bench.py --cpu-sample none --no-extras $v 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('  ms_per_step %.3f  median %.3f' % (d['ms_per_step'], d['step_ms']['median']))
for o in d['roofline'].get('others',[]): print('    %-90s %.3f x%s' % (o['kernel'][:90], o['ms'], o.get('calls_per_step')))
" >> $O
done; done
cat $O
