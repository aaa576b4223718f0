# what the driver runs at round end, in one gpurun call (round 6): smoke(), the default bench.py, the whole -m gpu suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
( time python bench.py ) > $O/bench_default.txt 2>&1
( time python -m pytest tests -m gpu -x -q ) > $O/gpu_suite.txt 2>&1
tail -3 $O/smoke.txt; tail -5 $O/bench_default.txt | cut -c1-400; tail -8 $O/gpu_suite.txt
for m in 32768 1 4096; do GEOGCN_X3_ROWS_MIN_M=$m python bench.py --shape cmu --no-extras --traffic none --cpu-sample none --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cmu x3 min M $m', round(d['ms_per_step'],4))"; done
