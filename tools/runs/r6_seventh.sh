O=gpurun_out/r6_seventh; mkdir -p $O; cd $GRAFT_REPO_ROOT
bash tools/runs/r6_final.sh r6_seventh/final trace > /dev/null 2>&1
head -40 $O/final/kernel_stats.md
python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest_gpu.txt 2>&1; tail -40 $O/pytest_gpu.txt
