mkdir -p gpurun_out/r2x; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2x
python -m pytest tests/test_edge_cases_gpu.py tests/test_tdvp_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -15 $O/pytest.log
python bench.py --steps 6 --warmup 2 --cpu-updates 0 > $O/bench_defer.json 2> $O/err1.log
MPSE_DEFER=0 python bench.py --steps 6 --warmup 2 --cpu-updates 0 > $O/bench_nodefer.json 2>/dev/null
python bench.py --steps 6 --warmup 2 --cpu-updates 0 > $O/bench_defer2.json 2>/dev/null
for f in $O/bench_*.json; do echo $f; cut -c90-130 $f; done; tail -3 $O/err1.log
