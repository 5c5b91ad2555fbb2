# A/B of one environment switch on the headline: bash tools/runs/ab.sh ENVVAR [pytest -k expression]
V=$1; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py tests/test_tdvp_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
for f in 1 0 1 0 1 0; do echo "$V=$f"; env $V=$f python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/dev/null | python -c "$P"; done
