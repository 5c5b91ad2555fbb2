# every engine switch that selects an alternative path, under the TDVP / engine / DMRG suites (GPU box; ~1.5 min each)
cd $GRAFT_REPO_ROOT
for t in "MPSE_DEFER=0" "MPSE_QR_FIT=0" "MPSE_DOT_FUSED=0" "MPSE_ENV_CARRY=0" "MPSE_LANCZOS_ASYNC=0" "MPSE_QR_WY=0" "MPSE_BETA_SOURCE=0" \
         "MPSE_WSMALL=0" "MPSE_STAGE_KERNEL=0" "MPSE_CENTRE_MASK=0" "MPSE_LZ_DEFER_FIRST=0" "MPSE_QR_LOOKAHEAD=0" "MPSE_QR_CAQR=1" \
         "MPSE_QR_GRAPH=1" "MPSE_GEMM_SKEW=0" "MPSE_GEMM_ORDER=0" "MPSE_GEMM_ORDER=2" "MPSE_GEMM_WIDE=0" "MPSE_GEMM_SLICEFAST=0" \
         "MPSE_GEMM_DIEGROUP=0" "MPSE_SPLITK_BAL=2" "MPSE_MASKED_CHAIN=1" "MPSE_SMALL_TILES=1"; do
  echo "== $t"; env $t python -m pytest tests/test_tdvp_gpu.py tests/test_engine_gpu.py tests/test_dmrg_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -2
done
