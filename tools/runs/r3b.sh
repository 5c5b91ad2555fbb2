mkdir -p gpurun_out/r3b; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3b
for n in 0 1 4 5; do MPSE_QR_NARROW=$n python tools/qr_bench.py > $O/qr_narrow$n.txt 2>&1; echo "narrow=$n"; head -4 $O/qr_narrow$n.txt; done
MPSE_QR_NARROW=5 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "qr" 2>&1 | tail -3
for n in 0 5 1 0 5; do MPSE_QR_NARROW=$n python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/dev/null | cut -c90-125; done
