O=gpurun_out/r6_12; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.txt 2>&1; tail -22 $O/pytest_gpu.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; cut -c1-300 $O/bench20.json
