# headline A/B of one environment switch, alternating runs on one box: bash tools/runs/r5_headline_ab.sh <tag> <VAR> <a> <b> [rounds]
T=${1:-ab}; V=$2; A=$3; B=$4; R=${5:-2}; O=gpurun_out/$T; mkdir -p $O; cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), round(d["ms_per_step"],1), d["config"]["mean_krylov_dim"], d["config"].get("block_qr"))'
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for i in $(seq $R); do for x in $A $B; do
  echo -n "$V=$x: "; env $V=$x python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-updates 0 --state-file /tmp/state.npz 2>>$O/err.log | tee -a $O/${V}_$x.jsonl | python -c "$P"
done; done
