# Round 5, first pass: this round's baseline on this box, conditioning of the block-QR inputs, calibration of the MFMA counter.
T=${1:-r5_first}; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench20.err
(timeout 900 python tools/qr_cond_study.py $O/qr_cond_step2 1 > $O/qr_cond_step2.out 2>&1)
(timeout 900 python tools/qr_cond_study.py $O/qr_cond_step13 12 > $O/qr_cond_step13.out 2>&1)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f64_peak.hip -o /tmp/mfma_f64_peak 2> $O/ubench_build.err
/tmp/mfma_f64_peak > $O/mfma_f64_peak.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_peak -o p -- /tmp/mfma_f64_peak > $O/pmc_peak.log 2>&1
python tools/pmc_mfma_util.py $O/pmc_peak/p_results.db $O/pmc_mfma_util_calibration_peak.md k_peak > /dev/null 2> $O/pmc_peak.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_gemm -o p -- python tools/gemm_bench.py > $O/pmc_gemm.log 2>&1
python tools/pmc_mfma_util.py $O/pmc_gemm/p_results.db $O/pmc_mfma_util_calibration_gemm.md > /dev/null 2> $O/pmc_gemm.err
rm -rf $O/pmc_peak $O/pmc_gemm
cut -c1-300 $O/bench_20steps.json; cat $O/qr_cond_step2.out | tail -3; cat $O/pmc_mfma_util_calibration_peak.md; ls -la $O
