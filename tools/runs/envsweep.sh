# headline bench under a list of environment settings (each argument: "VAR=val VAR2=val"), interleaved twice (GPU box)
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
python bench.py --steps 2 --warmup 1 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for rep in 1 2; do
  for setting in "$@"; do
    echo -n "$setting : "; env $setting python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/dev/null | python -c "$P"
  done
done
