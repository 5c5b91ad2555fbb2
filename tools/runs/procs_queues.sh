# Trajectories as separate processes on ONE GPU with the number of hardware queues per process bounded
# (GPU_MAX_HW_QUEUES): does the collapse beyond two processes come from queue oversubscription?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/procs_queues.jsonl; : > $O
run() { echo "# $1 $2 $3 q=$4" >> $O; if [ -n "$4" ]; then export GPU_MAX_HW_QUEUES=$4; else unset GPU_MAX_HW_QUEUES; fi
        timeout 600 python tools/traj_scaling.py $1 $2 $3 2>/tmp/err.log | tail -1 >> $O || tail -3 /tmp/err.log >> $O; }
run threads 1 2 ""
run procs 4 2 ""
run procs 4 2 2
run procs 4 2 1
run procs 8 2 2
run procs 8 2 1
run threads 8 2 1
cat $O
