# Measurement pass of a round: the driver's bench line, kernel trace, PMC passes, side measurements (no pytest: run
# tools/runs/suite_and_bench.sh for that).  Usage (GPU box, through gpurun): bash tools/runs/final.sh <tag>
T=${1:-final}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 --state-file /tmp/state.npz > $O/bench_20steps.json 2> $O/bench20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions -d $O/prof -o b -- python bench.py --cpu-updates 0 --steps 2 --state-file /tmp/state.npz > $O/bench_under_rocprof.json 2> $O/err.log
python tools/rocpd_summary.py $O/prof/b_results.db $O/kernel_stats.md > /dev/null
python tools/rocpd_gaps.py $O/prof/b_results.db > $O/gaps.md
python tools/rocpd_by_grid.py $O/prof/b_results.db k_gemm $O/gemm_by_grid.md > /dev/null
cp $O/prof/b_kernel_stats.csv $O/rocprofv3_kernel_stats.csv 2>/dev/null; rm -rf $O/prof
for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python bench.py --cpu-updates 0 --steps 1 --warmup 0 --state-file /tmp/state.npz > $O/pmc_$c.log 2>&1; done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/p_results.db $O/pmc_WRITE_SIZE/p_results.db $O/pmc_traffic.json $O/pmc_traffic.md > /dev/null 2> $O/pmc_traffic.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o p -- python bench.py --cpu-updates 0 --steps 1 --warmup 0 --state-file /tmp/state.npz > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma_util.py $O/pmc_mfma/p_results.db $O/pmc_mfma_util.md > /dev/null 2> $O/pmc_mfma.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
MPSE_GEMM_TRACE=/tmp/t.bin python bench.py --steps 1 --warmup 1 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1; python tools/gemm_trace.py /tmp/t.bin $O/gemm_trace.md > /dev/null; python tools/gemm_balance.py /tmp/t.bin $O/gemm_balance.md > /dev/null
for t in 2 4; do python bench.py --cpu-updates 0 --steps 3 --traj-per-gpu $t 2>/dev/null >> $O/multi_traj.jsonl; done
MPSE_ENV_CARRY=0 python bench.py --cpu-updates 0 --steps 5 --warmup 2 --state-file /tmp/state.npz > $O/bench_nocarry.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 1 --warmup 0 --cpu-updates 0 > $O/bench_2rank_shared.json 2> $O/bench_2rank.err
MPSE_RCCL_TIMEOUT=40 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --share-gpu --steps 1 --warmup 0 --cpu-updates 0 > $O/bench_2rank_strict.json 2> $O/bench_2rank_strict.err; echo "strict 2-rank exit code $?" >> $O/bench_2rank_strict.err
MPSE_COLLECTIVE=file python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --share-gpu --steps 1 --warmup 0 --cpu-updates 0 > $O/bench_2rank_file.json 2> $O/bench_2rank_file.err
(timeout 900 python tools/config_times.py $O/config_times.md > /dev/null) 2> $O/config_times.err
cut -c1-200 $O/bench_20steps.json; ls -la $O
