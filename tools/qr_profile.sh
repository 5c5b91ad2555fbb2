# rocprofv3 kernel trace of the block QR alone at the two headline shapes (GPU box): bash tools/qr_profile.sh <outdir>
O=${1:-gpurun_out/qrprof}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for shape in "256 16 256" "256 2 256"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --kernel-trace --stats -d $O/p_$tag -o q -- python tools/_qr_one.py $shape 2 > $O/log_$tag.txt 2>&1
  python tools/rocpd_summary.py $O/p_$tag/q_results.db $O/qr_kernels_$tag.md > /dev/null
  python tools/rocpd_gaps.py $O/p_$tag/q_results.db > $O/qr_gaps_$tag.md
  rm -rf $O/p_$tag
done
