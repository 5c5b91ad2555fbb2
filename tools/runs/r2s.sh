mkdir -p gpurun_out/r2s; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2s
(time python -m pytest tests -m gpu -q -x) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/qr_bench.py > $O/qr_default.txt 2>&1
MPSE_QR_LOOKAHEAD=0 python tools/qr_bench.py > $O/qr_nolook.txt 2>&1
MPSE_FORMQ_PAIR=0 python tools/qr_bench.py > $O/qr_nopair.txt 2>&1
python bench.py --steps 5 --warmup 2 --cpu-updates 0 > $O/bench_default.json 2> $O/bench_default.err
MPSE_QR_LOOKAHEAD=0 MPSE_FORMQ_PAIR=0 python bench.py --steps 5 --warmup 2 --cpu-updates 0 > $O/bench_oldqr.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --cpu-updates 0 > $O/bench_default2.json 2>/dev/null
for f in $O/qr_*.txt; do echo $f; cat $f; done
for f in $O/bench_*.json; do echo $f; cut -c90-130 $f; done
# PMC probe on a small chain first
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_small -o p -- python bench.py --nmol 4 --bond-dim 64 --cpu-updates 0 --steps 1 --warmup 0 > $O/pmc_small.log 2>&1; echo "pmc small exit $?"
ls -la $O/pmc_small 2>/dev/null | head
timeout 240 env MPSE_LANCZOS_ASYNC=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_small_sync -o p -- python bench.py --nmol 4 --bond-dim 64 --cpu-updates 0 --steps 1 --warmup 0 > $O/pmc_small_sync.log 2>&1; echo "pmc small sync exit $?"
rm -rf $O/pmc_small $O/pmc_small_sync
