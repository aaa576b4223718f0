cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05b
rm -rf /tmp/tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o b -- python $GRAFT_REPO_ROOT/tools/sim_rank.py --world 8 --exchange a2a --steps 5 --gemm-precision bf16x3 > /tmp/sim.txt 2>&1
tail -2 /tmp/sim.txt
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/tr/b_kernel_trace.csv > $GRAFT_REPO_ROOT/gpurun_out/r05b/sim_rank_w8_a2a_kernel_stats.md
head -45 $GRAFT_REPO_ROOT/gpurun_out/r05b/sim_rank_w8_a2a_kernel_stats.md | cut -c1-140
