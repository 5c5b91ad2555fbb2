# Block-SVD evidence at the headline bond size: bench.py --scheme tdvp_ps2 (two-site TDVP on the 50-site chain, D = 256),
# its bench line, kernel trace and PMC traffic.  Usage (GPU box, through gpurun): bash tools/runs/svd_profile.sh <tag>
T=${1:-svd}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 --state-file /tmp/state.npz > $O/bench_20steps.json 2> $O/bench20.err
cut -c1-160 $O/bench_20steps.json
timeout 900 python bench.py --scheme tdvp_ps2 --steps 2 --warmup 1 --cpu-updates 0 --state-file /tmp/state.npz > $O/bench_ps2.json 2> $O/bench_ps2.err
cut -c1-200 $O/bench_ps2.json; tail -3 $O/bench_ps2.err
timeout 900 rocprofv3 --kernel-trace --marker-trace --stats --selected-regions -d $O/prof -o b -- python bench.py --scheme tdvp_ps2 --cpu-updates 0 --steps 1 --warmup 1 --state-file /tmp/state.npz > $O/bench_ps2_under_rocprof.json 2> $O/err.log
python tools/rocpd_summary.py $O/prof/b_results.db $O/svd_kernel_stats.md > /dev/null
python tools/rocpd_gaps.py $O/prof/b_results.db > $O/svd_gaps.md
rm -rf $O/prof
head -30 $O/svd_kernel_stats.md
