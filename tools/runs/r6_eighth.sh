O=gpurun_out/r6_eighth; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py -q -x -k "heff or fused or matvec or expm" > $O/pytest_engine.txt 2>&1; tail -3 $O/pytest_engine.txt
bash tools/runs/r6_ab.sh r6_eighth/ab 2 "plain:MPSE_F0_ORDER=0" "planned:MPSE_F0_ORDER=1" 2>&1 | tee $O/ab.txt
bash tools/runs/r6_final.sh r6_eighth/final trace > /dev/null 2>&1
head -14 $O/final/kernel_stats.md; grep "k_f0" $O/final/kernel_stats.md
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_h2o -o b -- python examples/h2o_dmrg.py > $O/h2o.log 2>&1
python tools/rocpd_summary.py $O/prof_h2o/b_results.db $O/h2o_kernel_stats.md > /dev/null; python tools/rocpd_gaps.py $O/prof_h2o/b_results.db > $O/h2o_gaps.md; rm -rf $O/prof_h2o
head -30 $O/h2o_kernel_stats.md; tail -2 $O/h2o_kernel_stats.md; head -3 $O/h2o_gaps.md; tail -4 $O/h2o.log
