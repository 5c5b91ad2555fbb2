O=gpurun_out/${1:-r5_f0dbg}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 5 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for v in ${2:-0 1 2 3}; do
  MPSE_F0_DBG=$v rocprofv3 --kernel-trace --stats -d $O/p_$v -o q -- python bench.py --cpu-updates 0 --steps 1 --warmup 0 --state-file /tmp/state.npz > $O/log_$v.txt 2>&1
  python tools/rocpd_summary.py $O/p_$v/q_results.db $O/k_$v.md > /dev/null; rm -rf $O/p_$v; echo "== dbg $v"; grep "heff0_fused" $O/k_$v.md | cut -c1-100
done
