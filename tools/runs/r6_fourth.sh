O=gpurun_out/r6_fourth; mkdir -p $O; cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>>$O/err.log
MPSE_GEMM_TRACE=$O/f0_trace.bin MPSE_GEMM_TRACE_ONLY=f0 python bench.py --steps 1 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz > $O/bench_traced.json 2>>$O/err.log
python tools/f0_trace.py $O/f0_trace.bin $O/f0_trace.md; rm -f $O/f0_trace.bin
for i in 1 2; do for c in 96 48; do echo "MINCOLS=$c"; MPSE_CHOLQR_MINCOLS=$c python tools/small_ab.py sbm holstein fmo77 2>>$O/err.log | tee -a $O/small_mincols_$c.jsonl | cut -c1-200; done; done
for q in 4 8 16; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python tools/traj_scaling.py threads 8 2 2>>$O/err.log | tee -a $O/traj_queues.jsonl; done
python tools/traj_scaling.py threads 1 2 2>>$O/err.log | tee -a $O/traj_queues.jsonl
