# generic same-box A/B of two builds of the library (run through gpurun from the repo root):
#   new = geographconv_amd/libgeogcn.so, old = tools/micro/bin/libgeogcn_ab.so (built beforehand from the commit to compare with)
#   bash tools/ab_lib.sh <out-name> <pytest -k expression> <bench args...>
cd $GRAFT_REPO_ROOT
NAME=$1; KEXPR=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/$NAME.txt
: > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -x -k "$KEXPR" 2>&1 | tail -2 >> $O
for rep in 1 2 3; do
  for lib in new old; do
    if [ $lib = old ]; then cp geographconv_amd/libgeogcn.so /tmp/new.so; cp tools/micro/bin/libgeogcn_ab.so geographconv_amd/libgeogcn.so; fi
    echo "== $lib: bench.py $*" >> $O
    timeout 600 python bench.py "$@" --cpu-sample none 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])" >> $O
    if [ $lib = old ]; then cp /tmp/new.so geographconv_amd/libgeogcn.so; fi
  done
done
cat $O
