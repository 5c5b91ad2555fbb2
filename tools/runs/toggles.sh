# Every remaining engine switch through the TDVP / engine / DMRG suites: bash tools/runs/toggles.sh (GPU box, via gpurun)
cd $GRAFT_REPO_ROOT
for t in "MPSE_DEFER=0" "MPSE_ENV_CARRY=0" "MPSE_LANCZOS_ASYNC=0" "MPSE_CENTRE_MASK=0" "MPSE_WFOLD=0" "MPSE_SPLIT2=0" "MPSE_SMALL=0" "MPSE_HEFF0=0" "MPSE_HEFF0=2" "MPSE_CHOLQR=0" "MPSE_CHOLQR=2" "MPSE_QR_OPTIMISTIC=0" "MPSE_VEC_MASK=0" "MPSE_ENV_WFOLD=0"; do
  echo "== $t"; env $t python -m pytest tests/test_tdvp_gpu.py tests/test_engine_gpu.py tests/test_dmrg_gpu.py -m gpu -q -x 2>&1 | tail -1
done
