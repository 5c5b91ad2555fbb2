cd $GRAFT_REPO_ROOT
for t in "MPSE_LANCZOS_ASYNC=0" "MPSE_HEFF0=2"; do
  echo "== $t"; env $t python -m pytest tests/test_tdvp_gpu.py tests/test_engine_gpu.py tests/test_dmrg_gpu.py -m gpu -q -x 2>&1 | grep -E "^(FAILED|E  )|passed|failed" | head -12
done
