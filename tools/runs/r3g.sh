mkdir -p gpurun_out/r3g; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3g
python -m pytest tests/test_engine_gpu.py tests/test_tdvp_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1), round(d["config"]["mean_krylov_dim"],3), d["roofline"]["traffic"])'
for f in 1 2 3; do python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/dev/null | python -c "$P"; done
