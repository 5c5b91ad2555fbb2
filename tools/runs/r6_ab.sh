# round 6: A/B of named environment settings on the headline, alternating on one box from one state file
# usage: bash tools/runs/r6_ab.sh <tag> <rounds> "<name>:<VAR=val VAR=val ...>" ...
T=$1; R=$2; shift 2; O=gpurun_out/$T; mkdir -p $O; cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); q=d["config"]["block_qr"]; print(round(d["value"],1), round(d["ms_per_step"],1), d["config"]["mean_krylov_dim"], q.get("cholesky_qr_blocks"), q.get("cholesky_qr_blocks_two_passes"), q.get("steps_repeated_in_timed_region"))'
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>>$O/err.log
for i in $(seq $R); do for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  echo -n "$name [$envs]: "; env $envs python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-updates 0 --state-file /tmp/state.npz 2>>$O/err.log | tee -a $O/$name.jsonl | python -c "$P"
done; done
