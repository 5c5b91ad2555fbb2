# development: the TRSM variants of the Cholesky-QR path, per-kernel times at the two headline shapes
O=gpurun_out/${1:-r5_trsm_var}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1 2; do for shape in "256 16 256" "256 2 256"; do
  tag=v${v}_$(echo $shape | tr ' ' 'x')
  MPSE_CQ_TRSM=$v rocprofv3 --kernel-trace --stats -d $O/p_$tag -o q -- python tools/_qr_one.py $shape 2 > $O/log_$tag.txt 2>&1
  python tools/rocpd_summary.py $O/p_$tag/q_results.db $O/k_$tag.md > /dev/null; rm -rf $O/p_$tag
  echo "== $tag"; grep "trsm\|chol\|gram\|reduce" $O/k_$tag.md
done; done
