mkdir -p gpurun_out/r3i; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3i
for n in 1 0; do echo "short=$n"; MPSE_QR_FIT=$n python tools/qr_bench.py 2>&1 | head -8; done
python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "qr or svd" 2>&1 | grep -E "passed|failed"
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
for f in 1 0 1 0; do echo "short=$f"; MPSE_QR_FIT=$f python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/dev/null | python -c "$P"; done
