# 2 x 2 on one box: {round-3 engine, round-4 engine} x {start state prepared by round 3, by round 4} (driver's 20 steps)
O=gpurun_out/${1:-vs_r3s}; mkdir -p $O; cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1), "mean Krylov dim", round(d["config"]["mean_krylov_dim"],3))'
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state_r4.npz > /dev/null 2>&1
(cd gpurun_tmp/r3 && python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state_r3.npz > /dev/null 2>&1)
for st in r3 r4; do
  echo "state prepared by $st, engine round 4"; python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-updates 0 --state-file /tmp/state_$st.npz 2>/dev/null | tee $O/eng4_state_$st.json | python -c "$P"
  echo "state prepared by $st, engine round 3"; (cd gpurun_tmp/r3 && python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-updates 0 --state-file /tmp/state_$st.npz 2>/dev/null) | tee $O/eng3_state_$st.json | python -c "$P"
done
