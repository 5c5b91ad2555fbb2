O=gpurun_out/r6_fifth; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py -q -x -k "heff or fused or matvec or expm" > $O/pytest_engine.txt 2>&1; tail -3 $O/pytest_engine.txt
bash tools/runs/r6_ab.sh r6_fifth/ab 2 "plain:MPSE_F0_ORDER=0" "ordered:MPSE_F0_ORDER=1" 2>&1 | tee $O/ab.txt
MPSE_GEMM_TRACE=$O/f0_trace.bin MPSE_GEMM_TRACE_ONLY=f0 python bench.py --steps 1 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz > $O/bench_traced.json 2>>$O/err.log
python tools/f0_trace.py $O/f0_trace.bin $O/f0_trace_ordered.md; rm -f $O/f0_trace.bin
