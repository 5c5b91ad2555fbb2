mkdir -p gpurun_out/r2y; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2y
for d in 1 0; do
MPSE_DEFER=$d rocprofv3 --kernel-trace --marker-trace --selected-regions -d $O/prof$d -o b -- python bench.py --cpu-updates 0 --steps 2 > $O/bench_$d.json 2> $O/err$d.log
python tools/rocpd_gaps.py $O/prof$d/b_results.db > $O/gaps_$d.md
rm -rf $O/prof$d
done
head -30 $O/gaps_1.md; head -30 $O/gaps_0.md
