# one environment variable on the launch-bound configurations and the headline, alternating: bash tools/runs/env_small.sh VAR "v1 v2" [rounds]
V=$1; cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for r in $(seq 1 ${3:-2}); do for f in $2; do
  if [ "$f" = unset ]; then unset $V; else export $V=$f; fi
  echo "== $V=$f"
  timeout 300 python tools/small_ab.py sbm fmo 2>/dev/null | python -c 'import sys,json
for l in sys.stdin: d=json.loads(l); print(d["config"], round(d["site_updates_per_s"],1))'
  echo -n "headline: "; python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/dev/null | python -c "$P"
done; done
