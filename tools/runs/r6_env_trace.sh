O=gpurun_out/r6_env; mkdir -p $O; cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
python tools/env_bench.py 3 > /dev/null 2>&1
MPSE_GEMM_TRACE=$O/trace.bin python tools/env_bench.py 1 > $O/env_traced.json 2>$O/err.log
python tools/gemm_trace.py $O/trace.bin $O/env_gemm_trace.md; rm -f $O/trace.bin
