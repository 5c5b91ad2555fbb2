# A/B of a switch that is ON when the variable is unset: bash tools/runs/ab2.sh ENVVAR [pytest args]
V=$1; cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then python -m pytest $2 -m gpu -q -x 2>&1 | tail -5; fi
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for i in 1 2 3; do echo "default"; python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/tmp/err.log | python -c "$P" || tail -5 /tmp/err.log; echo "$V=1"; env $V=1 python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/tmp/err.log | python -c "$P" || tail -5 /tmp/err.log; done
