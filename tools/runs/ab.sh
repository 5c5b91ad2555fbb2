# A/B of one environment switch on the headline: bash tools/runs/ab.sh ENVVAR "pytest files / -k expression" [rounds]
V=$1; cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then python -m pytest $2 -m gpu -q -x 2>&1 | tail -15; fi
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for f in ${3:-1 0 1 0 1 0}; do echo "$V=$f"; env $V=$f python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/tmp/err.log | python -c "$P" || tail -5 /tmp/err.log; done
