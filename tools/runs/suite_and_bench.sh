# full GPU suite (timed) + the driver's bench command: bash tools/runs/suite_and_bench.sh <tag>
T=${1:-suite}; O=gpurun_out/$T; mkdir -p $O; cd $GRAFT_REPO_ROOT
(time python -m pytest tests -m gpu -q --durations=15) > $O/pytest.log 2>&1; tail -25 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench20.err; cut -c1-220 $O/bench_20steps.json
