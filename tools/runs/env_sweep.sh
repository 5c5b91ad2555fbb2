# headline under values of one environment variable, alternating: bash tools/runs/env_sweep.sh VAR "v1 v2 .." [rounds]
V=$1; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for r in $(seq 1 ${3:-3}); do for f in $2; do
  if [ "$f" = unset ]; then unset $V; else export $V=$f; fi
  echo -n "$V=$f: "; python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/tmp/err.log | python -c "$P" || tail -5 /tmp/err.log
done; done
