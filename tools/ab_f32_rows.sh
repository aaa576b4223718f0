# same-box A/B of the whole-rows fp32 GEMM kernel (run through gpurun from the repo root)
# the 'old' side is the same library built without the kernel, made beforehand (it travels with the snapshot):
#   GEOGCN_BUILD_DEFINES=GEOGCN_F32_NO_ROWS_KERNEL python -m geographconv_amd.build --force && cp geographconv_amd/libgeogcn.so tools/micro/bin/libgeogcn_norows_f32.so
#   python -m geographconv_amd.build --force
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_u_f32_rows_ab.txt
: > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or whole_rows" 2>&1 | tail -3 >> $O
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x 2>&1 | tail -3 >> $O
for rep in 1 2 3; do
  for lib in new old; do
    if [ $lib = old ]; then cp geographconv_amd/libgeogcn.so /tmp/new.so; cp tools/micro/bin/libgeogcn_norows_f32.so geographconv_amd/libgeogcn.so; fi
    echo "== $lib TWUS 3x300 fp32" >> $O
    timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], [ (o['name'], round(o['ms'],3)) for o in d['roofline'].get('others', []) if 'gemm' in o.get('name','')])" >> $O
    if [ $lib = old ]; then cp /tmp/new.so geographconv_amd/libgeogcn.so; fi
  done
done
cat $O
