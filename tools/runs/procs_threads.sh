# trajectories at the launch-bound size as P processes x T threads on one GPU (two processes are not time-sliced against
# each other, more are: profiles/r04_traj_procs_queues.jsonl): bash tools/runs/procs_threads.sh P T [steps]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; P=${1:-2}; T=${2:-4}; S=${3:-4}
for p in $(seq 1 $P); do (timeout 900 python tools/traj_scaling.py threads $T $S 2>/dev/null | tail -1 > /tmp/pt_$p.json) & done; wait
python - <<PY
import json, glob
rows=[json.load(open(f)) for f in sorted(glob.glob("/tmp/pt_*.json"))]
print(json.dumps({"workload": rows[0]["workload"], "mode": "%d processes x %d threads" % ($P, $T), "steps": $S,
                  "site_updates_per_s_each": [round(r["site_updates_per_s"], 1) for r in rows],
                  "site_updates_per_s_sum": round(sum(r["site_updates_per_s"] for r in rows), 1),
                  "wall_s_each": [round(r["wall_s"], 2) for r in rows]}))
PY
