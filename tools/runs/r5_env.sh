# environment-tensor contraction on its own: wall time + MFMA-busy of exactly its launches
T=${1:-r5_env}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/env_bench.py 5 $O/env_bench.json > $O/env_bench.out 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python tools/env_bench.py 2 > $O/trace.log 2>&1
python tools/rocpd_summary.py $O/prof/b_results.db $O/kernel_stats.md > /dev/null; rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o p -- python tools/env_bench.py 1 > $O/pmc.log 2>&1
python tools/pmc_mfma_util.py $O/pmc/p_results.db $O/pmc_mfma_util.md > /dev/null 2> $O/pmc.err; rm -rf $O/pmc
cat $O/env_bench.out | cut -c1-400; head -14 $O/kernel_stats.md | cut -c1-130; cat $O/pmc_mfma_util.md | cut -c1-200
