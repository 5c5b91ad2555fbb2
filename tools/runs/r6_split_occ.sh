O=gpurun_out/r6_split_occ; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py tests/test_tdvp_gpu.py -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
python tools/env_bench.py 2 > /dev/null 2>&1
for i in 1 2; do for v in 0 1; do echo -n "MPSE_SPLIT_OCC=$v env_bench: "; MPSE_SPLIT_OCC=$v python tools/env_bench.py 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['L']['ms'],2), round(d['R']['ms'],2), round(d['L']['tflops'],1), round(d['R']['tflops'],1))"; done; done | tee $O/env_ab.txt
bash tools/runs/r6_ab.sh r6_split_occ/ab 2 "ranges:MPSE_SPLIT_OCC=0" "occupied:MPSE_SPLIT_OCC=1" 2>&1 | tee $O/ab.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o p -- python tools/env_bench.py 1 > $O/pmc.log 2>&1
python tools/pmc_mfma_util.py $O/pmc/p_results.db $O/env_pmc_mfma_util.md > /dev/null 2> $O/pmc.err; rm -rf $O/pmc
head -9 $O/env_pmc_mfma_util.md | cut -c1-200
MPSE_GEMM_TRACE=$O/trace.bin python tools/env_bench.py 1 > /dev/null 2>&1; python tools/gemm_trace.py $O/trace.bin $O/env_gemm_trace.md | head -9; rm -f $O/trace.bin
