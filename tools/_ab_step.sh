cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_kernels_gpu.py -x -q -m gpu -k "spmm" 2>&1 | tail -3
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_g_trace_on -o bench -- python $R/bench.py --cpu-sample none --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_g_trace_off -o bench -- python $R/bench.py --cpu-sample none --steps 10 --warmup 3 --set FUSE_CARRY=0 --set FUSE_DROPOUT=0 > /dev/null 2>&1
cd $R
for s in "" "--set FUSE_CARRY=0 --set FUSE_DROPOUT=0" "" "--set FUSE_CARRY=0 --set FUSE_DROPOUT=0"; do
  python bench.py --cpu-sample none --steps 30 --warmup 5 $s 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$s]', 'ms/step %.3f median %.3f' % (d['ms_per_step'], d['step_ms']['median']))"
done
