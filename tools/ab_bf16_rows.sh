# same-box A/B of the whole-rows bf16 GEMM kernel (run through gpurun from the repo root)
# the 'old' side is the same library built without the kernel, made beforehand (it travels with the snapshot):
#   GEOGCN_BUILD_DEFINES=GEOGCN_BF16_NO_ROWS_KERNEL python -m geographconv_amd.build --force && cp geographconv_amd/libgeogcn.so tools/micro/bin/libgeogcn_norows.so
#   python -m geographconv_amd.build --force
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_r_bf16_rows_ab.txt
: > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "bf16" 2>&1 | tail -3 >> $O
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then cp geographconv_amd/libgeogcn.so /tmp/new.so; cp tools/micro/bin/libgeogcn_norows.so geographconv_amd/libgeogcn.so; fi
    echo "== $lib 6x600 bf16" >> $O
    timeout 600 python bench.py --hid 600 600 600 600 600 600 --gemm-precision bf16 --steps 6 --warmup 2 --cpu-sample none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'])" >> $O
    echo "== $lib 3x300 bf16" >> $O
    timeout 600 python bench.py --gemm-precision bf16 --steps 10 --warmup 3 --cpu-sample none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'])" >> $O
    if [ $lib = old ]; then cp /tmp/new.so geographconv_amd/libgeogcn.so; fi
  done
done
cat $O
