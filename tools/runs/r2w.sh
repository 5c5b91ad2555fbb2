mkdir -p gpurun_out/r2w; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2w
python -m pytest tests/test_tdvp_gpu.py tests/test_rk45_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --steps 6 --warmup 2 --cpu-updates 0 > $O/bench_carry.json 2> $O/err1.log
MPSE_ENV_CARRY=0 python bench.py --steps 6 --warmup 2 --cpu-updates 0 > $O/bench_nocarry.json 2>/dev/null
python bench.py --steps 6 --warmup 2 --cpu-updates 0 > $O/bench_carry2.json 2>/dev/null
for f in $O/bench_*.json; do echo $f; cut -c90-130 $f; done
