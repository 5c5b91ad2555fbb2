mkdir -p gpurun_out/r2z; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2z
MPSE_DEFER=1 python tools/defer_probe.py > $O/defer1.txt 2>&1
MPSE_DEFER=0 python tools/defer_probe.py > $O/defer0.txt 2>&1
grep -E "evolve|pool" $O/defer1.txt | tail -16; echo; grep -E "evolve|pool" $O/defer0.txt | tail -16
