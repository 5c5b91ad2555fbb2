# kernel traces of a launch-bound configuration with and without the one-launch matvec: bash tools/runs/small_trace.sh [sbm|fmo|holstein]
W=${1:-sbm}; O=gpurun_out/small_trace_$W; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 on; do
  if [ "$v" = 0 ]; then export MPSE_SMALL=0; else unset MPSE_SMALL; fi
  rocprofv3 --kernel-trace --stats -d $O/prof_$v -o b -- python tools/small_ab.py $W > $O/line_$v.json 2> $O/err_$v.log
  python tools/rocpd_summary.py $O/prof_$v/b_results.db $O/kernel_stats_$v.md > /dev/null
  python tools/rocpd_gaps.py $O/prof_$v/b_results.db > $O/gaps_$v.md
  rm -rf $O/prof_$v
done
head -30 $O/kernel_stats_0.md; head -30 $O/kernel_stats_on.md; head -12 $O/gaps_on.md; cat $O/line_*.json
