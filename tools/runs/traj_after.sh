# trajectories sharing one GPU at the launch-bound size after the one-launch matvec (threads + streams)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/traj_scaling_small.jsonl; : > $O
for t in 1 2 4 8; do timeout 900 python tools/traj_scaling.py threads $t 2 2>/tmp/err.log | tail -1 >> $O || tail -3 /tmp/err.log >> $O; done
MPSE_SMALL=0 timeout 900 python tools/traj_scaling.py threads 1 2 2>/tmp/err.log | tail -1 >> $O
cat $O
