O=gpurun_out/${1:-r5_chol12}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MPSE_CHOLQR=2 timeout 300 python tools/cholqr_check.py $O/check.md > $O/check.out 2>&1; echo "exit $?" >> $O/check.out; cat $O/check.out
for v in 0 1 2; do
  MPSE_CHOLQR=2 MPSE_CQ_CHOL12=$v rocprofv3 --kernel-trace --stats -d $O/p_$v -o q -- python tools/cholqr_check.py > $O/log_$v.txt 2>&1
  python tools/rocpd_summary.py $O/p_$v/q_results.db $O/k_$v.md > /dev/null; rm -rf $O/p_$v; echo "== chol12 var $v"; grep "chol\|trsm\|gram\|reduce" $O/k_$v.md
done
