O=gpurun_out/r6_probe2; mkdir -p $O; cd $GRAFT_REPO_ROOT
python tools/krylov_margin_probe.py 1 > $O/margin.txt 2>&1 &
python tools/qr_bench.py > $O/qr_bench.txt 2>&1
MPSE_CHOLQR_TAU=0 MPSE_CHOLQR_THETA=0 python tools/qr_bench.py > $O/qr_bench_r5scheme.txt 2>&1
python tools/cholqr_check.py > $O/cholqr_check.txt 2>&1
python -m pytest tests/test_engine_gpu.py -q -x -k "block_qr" > $O/pytest_qr.txt 2>&1; tail -3 $O/pytest_qr.txt
wait
cat $O/qr_bench.txt $O/qr_bench_r5scheme.txt; tail -25 $O/cholqr_check.txt; tail -30 $O/margin.txt
