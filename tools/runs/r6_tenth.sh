O=gpurun_out/r6_tenth; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py tests/test_tdvp_gpu.py -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
bash tools/runs/r6_ab.sh r6_tenth/ab 2 "per8:MPSE_RED_PER_THREAD=8" "per4:MPSE_RED_PER_THREAD=4" "per2:MPSE_RED_PER_THREAD=2" 2>&1 | tee $O/ab.txt
bash tools/runs/r6_final.sh r6_tenth/final trace > /dev/null 2>&1
head -8 $O/final/kernel_stats.md; grep "lanczos\|lincomb\|dot_partial" $O/final/kernel_stats.md; tail -1 $O/final/kernel_stats.md
