cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/micro/bin/x3_rows
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_x3
rm -rf $OUT; mkdir -p $OUT
for v in 35 39 42 40; do   # N=600: full, no B loads, MFMAs only, no MFMAs  (ordinals: see the listing)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/a$v -o p -- $B pmc $v > $OUT/run_a$v.txt 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/b$v -o p -- $B pmc $v > $OUT/run_b$v.txt 2>&1
done
grep -h "^\[" $OUT/run_a*.txt
python - <<'PY'
import csv, glob, collections, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_x3'
for d in sorted(glob.glob(out+'/[ab]*')):
    agg=collections.defaultdict(list)
    for f in glob.glob(d+'/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'x3_rows_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(d), {k: '%.4g' % (sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
