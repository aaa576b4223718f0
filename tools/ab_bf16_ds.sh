cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_s_bf16_ds_ab.txt
: > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -x -k "highway or bf16 or config5" 2>&1 | tail -3 >> $O
for rep in 1 2; do
  for v in 1 0; do
    echo "== FUSE_BF16_DS=$v 6x600 bf16" >> $O
    timeout 600 python bench.py --hid 600 600 600 600 600 600 --gemm-precision bf16 --steps 6 --warmup 2 --cpu-sample none --set FUSE_BF16_DS=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])" >> $O
    echo "== FUSE_BF16_DS=$v 3x300 bf16" >> $O
    timeout 600 python bench.py --gemm-precision bf16 --steps 10 --warmup 3 --cpu-sample none --set FUSE_BF16_DS=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])" >> $O
  done
done
cat $O
