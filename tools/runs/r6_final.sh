# Measurement pass of round 6 (no pytest).  Usage (GPU box, through gpurun): bash tools/runs/r6_final.sh <tag> [part]
# part: all (default) | bench | trace | pmc | side | svd
T=${1:-r6_final}; P=${2:-all}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S="--state-file /tmp/state.npz"
if [ $P = all ] || [ $P = bench ]; then
python bench.py --gpus 1 --steps 20 --warmup 5 $S > $O/bench_20steps.json 2> $O/bench20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
fi
if [ $P = all ] || [ $P = trace ]; then
python bench.py --steps 1 --warmup 1 --cpu-updates 0 $S > /dev/null 2>&1
rocprofv3 --kernel-trace --marker-trace --stats --selected-regions -d $O/prof -o b -- python bench.py --cpu-updates 0 --steps 2 --warmup 5 $S > $O/bench_under_rocprof.json 2> $O/err.log
python tools/rocpd_summary.py $O/prof/b_results.db $O/kernel_stats.md > /dev/null
python tools/rocpd_gaps.py $O/prof/b_results.db > $O/gaps.md
python tools/rocpd_by_grid.py $O/prof/b_results.db k_gemm $O/gemm_by_grid.md > /dev/null
python tools/rocpd_kernel_time.py $O/prof/b_results.db $O/kernel_time.json > /dev/null
cp $O/prof/b_kernel_stats.csv $O/rocprofv3_kernel_stats.csv 2>/dev/null; rm -rf $O/prof
fi
if [ $P = all ] || [ $P = pmc ]; then
python bench.py --steps 1 --warmup 1 --cpu-updates 0 $S > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python bench.py --cpu-updates 0 --steps 1 --warmup 1 $S > $O/pmc_$c.log 2>&1; done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/p_results.db $O/pmc_WRITE_SIZE/p_results.db $O/pmc_traffic.json $O/pmc_traffic.md > /dev/null 2> $O/pmc_traffic.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o p -- python bench.py --cpu-updates 0 --steps 1 --warmup 1 $S > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma_util.py $O/pmc_mfma/p_results.db $O/pmc_mfma_util.md k_gemm $O/pmc_mfma_busy.json mpse_gemm.hip,mpse_plans.h,mpse_contract.hip > /dev/null 2> $O/pmc_mfma.err
python tools/pmc_mfma_util.py $O/pmc_mfma/p_results.db $O/pmc_mfma_util_fused.md k_heff0 $O/pmc_mfma_busy.json mpse_heff0.hip > /dev/null 2>> $O/pmc_mfma.err
python tools/pmc_mfma_util.py $O/pmc_mfma/p_results.db $O/pmc_mfma_util_cholqr.md k_cq_ $O/pmc_mfma_busy.json mpse_cholqr.hip > /dev/null 2>> $O/pmc_mfma.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
fi
if [ $P = all ] || [ $P = side ]; then
python bench.py --steps 1 --warmup 1 --cpu-updates 0 $S > /dev/null 2>&1
(timeout 300 python tools/cholqr_check.py $O/cholqr_check.md > $O/cholqr_check.out 2>&1; echo "exit $?" >> $O/cholqr_check.out)
(timeout 300 python tools/qr_bench.py > $O/qr_bench_chol.txt 2>&1); (MPSE_CHOLQR_THETA=0 MPSE_CHOLQR_TAU=0 timeout 300 python tools/qr_bench.py > $O/qr_bench_chol_r5scheme.txt 2>&1); (MPSE_CHOLQR=0 timeout 300 python tools/qr_bench.py > $O/qr_bench_hh.txt 2>&1)
(timeout 300 python tools/qr_trip_pattern.py > $O/qr_trip_pattern.txt 2>&1)
for v in "MPSE_CHOLQR_TAU=0" "MPSE_CHOLQR_TAU=0.1" "MPSE_F0_ORDER=0" "MPSE_F0_ORDER=1" "MPSE_CHOLQR=0" "MPSE_CHOLQR=1" "MPSE_HEFF0=0" "MPSE_HEFF0=1" "MPSE_CHOLQR_TAU=0" "MPSE_CHOLQR_TAU=0.1" "MPSE_F0_ORDER=0" "MPSE_F0_ORDER=1" "MPSE_CHOLQR=0" "MPSE_CHOLQR=1" "MPSE_HEFF0=0" "MPSE_HEFF0=1"; do
  env $v python bench.py --steps 5 --warmup 3 --cpu-updates 0 $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'switch': '$v', 'value': round(d['value'],1), 'ms_per_step': round(d['ms_per_step'],2), 'block_qr': d['config']['block_qr']}))" >> $O/ab_switches.jsonl
done
bash tools/runs/r5_env.sh $T/env > /dev/null 2>&1
for t in 2 4; do python bench.py --cpu-updates 0 --steps 3 --traj-per-gpu $t 2>/dev/null >> $O/multi_traj.jsonl; done
MPSE_ENV_CARRY=0 python bench.py --cpu-updates 0 --steps 5 --warmup 2 $S > $O/bench_nocarry.json 2>/dev/null
python bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 1 --warmup 1 --cpu-updates 0 > $O/bench_2rank_spawned_shared.json 2> $O/bench_2rank_spawned.err; echo "spawned 2-rank (gloo, shared GPU) exit code $?" >> $O/bench_2rank_spawned.err
MPSE_RCCL_TIMEOUT=40 python bench.py --gpus 2 --share-gpu --steps 1 --warmup 1 --cpu-updates 0 > $O/bench_2rank_spawned_strict.json 2> $O/bench_2rank_spawned_strict.err; echo "spawned strict 2-rank exit code $?" >> $O/bench_2rank_spawned_strict.err
(timeout 900 python tools/config_times.py $O/config_times.md > /dev/null) 2> $O/config_times.err
fi
if [ $P = all ] || [ $P = svd ]; then
python bench.py --steps 1 --warmup 0 --cpu-updates 0 $S > /dev/null 2>&1
timeout 900 python bench.py --scheme tdvp_ps2 --steps 2 --warmup 1 --cpu-updates 0 $S > $O/bench_ps2.json 2> $O/bench_ps2.err
timeout 900 rocprofv3 --kernel-trace --marker-trace --stats --selected-regions -d $O/prof -o b -- python bench.py --scheme tdvp_ps2 --cpu-updates 0 --steps 1 --warmup 1 $S > $O/bench_ps2_under_rocprof.json 2> $O/err_svd.log
python tools/rocpd_summary.py $O/prof/b_results.db $O/svd_kernel_stats.md > /dev/null
python tools/rocpd_gaps.py $O/prof/b_results.db > $O/svd_gaps.md
rm -rf $O/prof
fi
cut -c1-200 $O/bench_20steps.json 2>/dev/null; ls -la $O
