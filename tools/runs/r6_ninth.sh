O=gpurun_out/r6_ninth; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py -q -x -k "heff or fused or matvec or expm" > $O/pytest_engine.txt 2>&1; tail -3 $O/pytest_engine.txt
bash tools/runs/r6_ab.sh r6_ninth/ab 2 "plain:MPSE_F0_ORDER=0" "planned:MPSE_F0_ORDER=1" 2>&1 | tee $O/ab.txt
bash tools/runs/r6_final.sh r6_ninth/final trace > /dev/null 2>&1
head -8 $O/final/kernel_stats.md; grep "k_f0" $O/final/kernel_stats.md; tail -1 $O/final/kernel_stats.md
