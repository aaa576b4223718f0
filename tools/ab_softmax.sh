cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in 1 0; do
rm -rf /tmp/tr_$v; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample none --steps 20 --warmup 3 --set FUSE_SOFTMAX=$v > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/tr_$v/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda:[0,0.0])
adam=0
for r in rows:
    k=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if "adam_kernel" in k: adam+=1
    for key in ("spmm_rows_kernel<4, 0, 16, 0, 2","spmm_rows_kernel<4, 0, 16, 0, 0","softmax_rows_reg","ce_rows4","ce_partial","spmm_long_reduce_kernel<0, 0"):
        if key in k: agg[key][0]+=1; agg[key][1]+=d
tot=sum((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows)
print("softmax fused $v: kernels per step %.1f us (%d steps)"%(tot/adam, adam), " ".join("%s %.1f x%d"%(k[-12:],t/n,n//adam) for k,(n,t) in agg.items()))
PY
done; done
