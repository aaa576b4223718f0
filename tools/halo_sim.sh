cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_q_halo_sim.txt
: > $O
python -m pytest tests/test_dist_gpu.py -q -k "comm_entry_points" 2>&1 | tail -2 >> $O
for ex in halo allgather a2a; do
  timeout 600 python tools/sim_rank.py --world 8 --shape twus_sbm --reorder lpa --exchange $ex 2>&1 | grep -E "^world|^halo" >> $O
done
timeout 600 python tools/sim_rank.py --world 8 --rank 5 --shape twus_sbm --reorder lpa --exchange halo 2>&1 | grep -E "^world|^halo" >> $O
timeout 600 python tools/sim_rank.py --world 4 --shape twus_sbm --reorder lpa --exchange halo 2>&1 | grep -E "^world|^halo" >> $O
timeout 600 python tools/sim_rank.py --world 2 --shape twus_sbm --reorder lpa --exchange halo 2>&1 | grep -E "^world|^halo" >> $O
timeout 600 python tools/sim_rank.py --world 8 --shape twus --exchange halo 2>&1 | grep -E "^world|^halo" >> $O
timeout 600 python bench.py --shape twus_sbm --reorder lpa --cpu-sample none 2>/dev/null | cut -c1-200 >> $O
cat $O
