#!/bin/bash
# Gram-matrix block step of the Jacobi SVD: the decomposition in isolation by kernel choice, and the suites that
# decompose (DMRG, compression, two-site TDVP, density operators) with the Gram step forced for every block size
set -x
mkdir -p gpurun_out/r6_svd
O=gpurun_out/r6_svd
MPSE_SVD_GRAM=0 timeout 600 python tools/svd_bench.py $O/svd_bench_columns.md > /dev/null 2>&1
MPSE_SVD_GRAM=2 timeout 600 python tools/svd_bench.py $O/svd_bench_gram.md > /dev/null 2>&1
timeout 600 python tools/svd_bench.py $O/svd_bench_default.md > /dev/null 2>&1
MPSE_SVD_GRAM=2 timeout 1500 python -m pytest tests/test_dmrg_gpu.py tests/test_observables_gpu.py tests/test_mpdm_gpu.py tests/test_tdvp_gpu.py -q -x -m gpu 2>&1 | tail -4 > $O/pytest_gram_forced.txt
cat $O/pytest_gram_forced.txt
