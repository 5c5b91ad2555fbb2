# one-launch matvec of small centres: parity tests, then A/B on the launch-bound configurations
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --durations=8 -k "${SMALL_AB_K:-small or heff or expm or contractions or lanczos}" 2>&1 | tail -16
O=gpurun_out/small_ab.jsonl; : > $O
for v in 0 "" 0 ""; do
  if [ -n "$v" ]; then export MPSE_SMALL=$v; else unset MPSE_SMALL; fi
  timeout 600 python tools/small_ab.py ${SMALL_AB_WHICH:-sbm fmo holstein} 2>/tmp/err.log >> $O || tail -5 /tmp/err.log
done
cat $O
