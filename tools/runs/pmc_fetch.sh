# FETCH_SIZE / WRITE_SIZE passes of the bench under the environment given as arguments: bash tools/runs/pmc_fetch.sh <tag> [VAR=val ...]
T=$1; shift; O=gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
[ -f /tmp/state.npz ] || python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do env "$@" timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python bench.py --cpu-updates 0 --steps 1 --warmup 0 --state-file /tmp/state.npz > $O/pmc_$c.log 2>&1; done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/p_results.db $O/pmc_WRITE_SIZE/p_results.db $O/pmc_traffic.json $O/pmc_traffic.md > /dev/null 2> $O/pmc_traffic.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE; head -5 $O/pmc_traffic.md
