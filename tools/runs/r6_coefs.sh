#!/bin/bash
set -x
mkdir -p gpurun_out/r6_coefs
O=gpurun_out/r6_coefs
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -k "expm or lanczos or krylov" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_tdvp_gpu.py -q -x 2>&1 | tail -3
timeout 1500 python tools/config_times.py $O/config_times.md > $O/config_times.log 2>&1; tail -9 $O/config_times.md
python bench.py --steps 5 --warmup 2 --cpu-updates 0 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('headline', d['value'], d['ms_per_step'])"
