# round 6: the default line's evidence (tools/collect_evidence.sh) + the other shapes' lines, one lease
cd $GRAFT_REPO_ROOT
bash tools/collect_evidence.sh > /dev/null 2>&1
O=gpurun_out/r06g; mkdir -p $O
python bench.py --shape cmu --no-extras --traffic none --cpu-sample none > $O/bench_cmu.json 2>/dev/null
python bench.py --gemm-precision bf16 --no-extras --traffic none --cpu-sample none > $O/bench_bf16.json 2>/dev/null
python bench.py --gemm-precision bf16 --hid 600 600 600 600 600 600 --no-extras --traffic none --cpu-sample none --steps 10 --warmup 3 > $O/bench_cfg5_6x600_bf16.json 2>/dev/null
python bench.py --shape twus_sbm --reorder lpa --no-extras --traffic none --cpu-sample none > $O/bench_twus_sbm_lpa.json 2>/dev/null
for f in $O/*.json gpurun_out/evidence/bench.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), d['roofline'].get('frac'))"; done
