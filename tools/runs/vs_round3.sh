# Same-box comparison with the round-3 tree (gpurun_tmp/r3, not committed): the driver's 20-step command, alternating
O=gpurun_out/${1:-vs_r3}; mkdir -p $O; cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
for i in 1 2; do
  echo "round 4"; python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-updates 0 2>/dev/null | tee $O/r4_$i.json | python -c "$P"
  echo "round 3"; (cd gpurun_tmp/r3 && python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-updates 0 2>/dev/null) | tee $O/r3_$i.json | python -c "$P"
done
