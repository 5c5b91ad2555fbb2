#!/bin/bash
set -x
mkdir -p gpurun_out/r6_svd
O=gpurun_out/r6_svd
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -k "svd or eigh" 2>&1 | tail -3
for rep in 1 2 3; do timeout 600 python tools/svd_bench.py /tmp/x.md "two-site centre 512 x 4096" 2>&1 | tail -4; done
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/svdprof -o p -- python $R/tools/svd_bench.py /tmp/x.md "two-site centre 512 x 4096, complex, 2" > /dev/null 2>&1
db=$(find /tmp/svdprof -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db $R/$O/kernel_stats_columns_fastrot.md | head -5
