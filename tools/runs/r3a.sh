mkdir -p gpurun_out/r3a; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3a
(time python bench.py --cpu-updates 0 --steps 1 --warmup 0 --state-file /tmp/state.npz) > $O/prep.log 2>&1; ls -la /tmp/state.npz
for c in FETCH_SIZE WRITE_SIZE; do
 (time timeout 700 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python bench.py --cpu-updates 0 --steps 1 --warmup 0 --state-file /tmp/state.npz) > $O/pmc_$c.log 2>&1; echo "pmc $c exit $?"; tail -4 $O/pmc_$c.log | grep real
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/p_results.db $O/pmc_WRITE_SIZE/p_results.db $O/pmc_traffic.json $O/pmc_traffic.md > /dev/null 2> $O/pmc_traffic.err
(time timeout 700 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o p -- python bench.py --cpu-updates 0 --steps 1 --warmup 0 --state-file /tmp/state.npz) > $O/pmc_mfma.log 2>&1; echo "pmc mfma exit $?"
python tools/pmc_mfma_util.py $O/pmc_mfma/p_results.db $O/pmc_mfma_util.md > /dev/null 2> $O/pmc_mfma.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
head -12 $O/pmc_traffic.md; head -12 $O/pmc_mfma_util.md; tail -2 $O/pmc_traffic.err $O/pmc_mfma.err
