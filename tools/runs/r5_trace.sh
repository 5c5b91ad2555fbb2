# kernel trace of two headline steps (timed region only) + summary files
T=${1:-r5_trace}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 5 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions -d $O/prof -o b -- python bench.py --cpu-updates 0 --steps 2 --warmup 5 --state-file /tmp/state.npz > $O/bench_under_rocprof.json 2> $O/err.log
python tools/rocpd_summary.py $O/prof/b_results.db $O/kernel_stats.md > /dev/null
python tools/rocpd_gaps.py $O/prof/b_results.db > $O/gaps.md
python tools/rocpd_by_grid.py $O/prof/b_results.db k_gemm $O/gemm_by_grid.md > /dev/null
python tools/rocpd_kernel_time.py $O/prof/b_results.db $O/kernel_time.json > /dev/null
rm -rf $O/prof
head -40 $O/kernel_stats.md | cut -c1-120; tail -4 $O/kernel_stats.md; head -3 $O/gaps.md
