cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_w_hw_ab.txt
: > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -x -k "highway or fused or reproducible" 2>&1 | tail -2 >> $O
cd /tmp && export TMPDIR=/tmp
for lib in new old new old; do
  if [ $lib = old ]; then cp $GRAFT_REPO_ROOT/geographconv_amd/libgeogcn.so /tmp/new.so; cp $GRAFT_REPO_ROOT/tools/micro/bin/libgeogcn_ab.so $GRAFT_REPO_ROOT/geographconv_amd/libgeogcn.so; fi
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o b -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample none --steps 10 --warmup 3 > /tmp/b.json 2>/dev/null
  echo "== $lib" >> $O
  python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $O
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/tr/b_kernel_trace.csv | grep -E "highway_bwd" | cut -c1-110 >> $O
  if [ $lib = old ]; then cp /tmp/new.so $GRAFT_REPO_ROOT/geographconv_amd/libgeogcn.so; fi
done
cat $O
