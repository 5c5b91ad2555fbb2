#!/bin/bash
# Gram-matrix block step of the Jacobi SVD on whole runs: two-site TDVP at the headline size, the config table, alternating
set -x
mkdir -p gpurun_out/r6_svd
O=gpurun_out/r6_svd
for rep in 1 2; do
  for v in 0 1; do
    MPSE_SVD_GRAM=$v timeout 900 python bench.py --scheme tdvp_ps2 --steps 2 --warmup 1 --cpu-updates 0 2>/dev/null | tail -1 > $O/ps2_gram${v}_$rep.json
    python - <<PY
import json
d=json.load(open("$O/ps2_gram${v}_$rep.json"))
print("ps2 gram=$v rep=$rep:", d["value"], d["ms_per_step"], [ (c.get("name"), c.get("ms_per_call"), c.get("sweeps_per_call")) for c in d["roofline"].get("classes",[]) if "svd" in str(c.get("name","")).lower() or "jacobi" in str(c.get("name","")).lower()])
PY
  done
done
for v in 0 1; do MPSE_SVD_GRAM=$v timeout 1500 python tools/config_times.py $O/config_times_gram$v.md > $O/config_times_gram$v.log 2>&1; tail -12 $O/config_times_gram$v.md; done
