# kernel traces of one configuration under two builds of the library: bash tools/runs/lib_trace.sh <other .so> [sbm|fmo|holstein] [kernel name filter]
OTHER=$GRAFT_REPO_ROOT/$1; W=${2:-sbm}; K=${3:-k_heff_small}; O=gpurun_out/lib_trace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in other this; do
  if [ $lib = other ]; then export RENO_MPSENGINE=$OTHER; else unset RENO_MPSENGINE; fi
  rocprofv3 --kernel-trace --stats -d $O/prof_$lib -o b -- python tools/small_ab.py $W > $O/line_$lib.json 2> $O/err_$lib.log
  python tools/rocpd_summary.py $O/prof_$lib/b_results.db $O/kernel_stats_$lib.md > /dev/null
  rm -rf $O/prof_$lib
  echo "== $lib"; grep -E "$K|total kernel" $O/kernel_stats_$lib.md; cut -c1-140 $O/line_$lib.json
done
