O=gpurun_out/${1:-r5_choldbg}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1 2 4 3 7; do
  MPSE_CHOLQR=2 MPSE_CQ_DBG=$v rocprofv3 --kernel-trace --stats -d $O/p_$v -o q -- python tools/_qr_one.py 256 2 256 2 > $O/log_$v.txt 2>&1
  python tools/rocpd_summary.py $O/p_$v/q_results.db $O/k_$v.md > /dev/null; rm -rf $O/p_$v; echo "== dbg $v"; grep "chol" $O/k_$v.md
done
