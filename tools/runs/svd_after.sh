T=${1:-svd2}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py tests/test_tdvp_gpu.py tests/test_dmrg_gpu.py tests/test_observables_gpu.py tests/test_mpdm_gpu.py -m gpu -q -x 2>&1 | tail -5
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
timeout 900 python bench.py --scheme tdvp_ps2 --steps 2 --warmup 1 --cpu-updates 0 --state-file /tmp/state.npz > $O/bench_ps2.json 2> $O/bench_ps2.err
cut -c1-200 $O/bench_ps2.json; tail -3 $O/bench_ps2.err
timeout 900 rocprofv3 --kernel-trace --marker-trace --stats --selected-regions -d $O/prof -o b -- python bench.py --scheme tdvp_ps2 --cpu-updates 0 --steps 1 --warmup 1 --state-file /tmp/state.npz > $O/bench_ps2_under_rocprof.json 2> $O/err.log
python tools/rocpd_summary.py $O/prof/b_results.db $O/svd_kernel_stats.md > /dev/null
python tools/rocpd_gaps.py $O/prof/b_results.db > $O/svd_gaps.md
rm -rf $O/prof
head -16 $O/svd_kernel_stats.md
