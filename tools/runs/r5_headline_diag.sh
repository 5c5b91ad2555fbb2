#!/bin/bash
# which switch moves the per-solve Krylov dimensions of the headline parity test
mkdir -p gpurun_out
for cfg in "MPSE_HEFF0=0" "MPSE_CHOLQR=0" "MPSE_QR_OPTIMISTIC=0" "MPSE_HEFF0=0 MPSE_CHOLQR=0"; do
  echo "=== $cfg" >> gpurun_out/r5_headline_diag.txt
  env $cfg timeout 600 python -m pytest tests/test_headline_gpu.py -q -m gpu -k one_evolve 2>&1 | grep -E "AssertionError|passed|failed" >> gpurun_out/r5_headline_diag.txt
done
