O=gpurun_out/r6_third; mkdir -p $O; cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>>$O/err.log
MPSE_GEMM_TRACE=$O/f0_trace.bin MPSE_GEMM_TRACE_ONLY=f0 python bench.py --steps 1 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz > $O/bench_traced.json 2>>$O/err.log
python tools/f0_trace.py $O/f0_trace.bin $O/f0_trace.md; rm -f $O/f0_trace.bin
bash tools/runs/r6_ab.sh r6_third/ab 2 "nofuse:MPSE_LZ_FUSE=0" "fuse:MPSE_LZ_FUSE=1" 2>&1 | tee $O/ab.txt
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
