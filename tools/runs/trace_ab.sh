# kernel traces of two variants of one switch: bash tools/runs/trace_ab.sh ENVVAR tag
V=$1; T=${2:-trace}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for f in 1 0; do
  env $V=$f rocprofv3 --kernel-trace --marker-trace --stats --selected-regions -d $O/prof$f -o b -- python bench.py --cpu-updates 0 --steps 2 --state-file /tmp/state.npz > $O/bench_$f.json 2> $O/err$f.log
  python tools/rocpd_summary.py $O/prof$f/b_results.db $O/kernel_stats_$f.md > /dev/null
  python tools/rocpd_by_grid.py $O/prof$f/b_results.db k_gemm $O/gemm_by_grid_$f.md > /dev/null
  python tools/rocpd_gaps.py $O/prof$f/b_results.db > $O/gaps_$f.md
  rm -rf $O/prof$f
done
