# MFMA-pipe utilisation of the GEMM kernels inside the training step (run through gpurun): one --pmc pass over 3 steps
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_rows
rm -rf $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o g -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample none --steps 3 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import collections, csv, glob, os
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_gemm_rows'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('geogcn::', '').replace('(anonymous namespace)::', '')
        if ('gemm' not in k and 'x3_' not in k) or 'prep' in k or 'splitk' in k:
            continue
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        agg[k.split('(')[0][:60]][r['Counter_Name']].append((float(r['Counter_Value']), us))
print('| kernel | launches | us (profiled) | SQ_VALU_MFMA_BUSY_CYCLES | GRBM_GUI_ACTIVE | MFMA pipe busy | shader clock |')
print('|---|---|---|---|---|---|---|')
for k, c in sorted(agg.items()):
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
        continue
    m = sum(v for v, _ in c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
    g = sum(v for v, _ in c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE'])
    us = sum(u for _, u in c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE'])
    print('| `%s` | %d | %.1f | %.4g | %.4g | %.1f %% | %.2f GHz |' % (k, len(c['GRBM_GUI_ACTIVE']), us, m, g, 100 * m / (g / 8 * 1024), g / 8 / us / 1e3))
PY
