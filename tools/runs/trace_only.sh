# the kernel-trace part of final.sh alone (when the tracer's host threads were starved in the full pass): bash tools/runs/trace_only.sh <tag>
T=${1:-trace}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
python bench.py --cpu-updates 0 --steps 2 --state-file /tmp/state.npz > $O/bench_untraced.json 2>/dev/null
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions -d $O/prof -o b -- python bench.py --cpu-updates 0 --steps 2 --state-file /tmp/state.npz > $O/bench_under_rocprof.json 2> $O/err.log
python tools/rocpd_summary.py $O/prof/b_results.db $O/kernel_stats.md > /dev/null
python tools/rocpd_gaps.py $O/prof/b_results.db > $O/gaps.md
python tools/rocpd_by_grid.py $O/prof/b_results.db k_gemm $O/gemm_by_grid.md > /dev/null
cp $O/prof/b_kernel_stats.csv $O/rocprofv3_kernel_stats.csv 2>/dev/null; rm -rf $O/prof
cut -c1-160 $O/bench_untraced.json $O/bench_under_rocprof.json; head -3 $O/gaps.md; uptime
