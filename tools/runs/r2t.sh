mkdir -p gpurun_out/r2u; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2u
rocprofv3 --kernel-trace --hip-trace -d $O/prof -o b -- python bench.py --cpu-updates 0 --steps 1 --warmup 1 > $O/bench.json 2> $O/err.log


python tools/rocpd_host_gaps.py $O/prof/b_results.db $O/host_gaps.md 8 k_lanczos_update_u > /dev/null 2> $O/host_gaps.err
ls -la $O/prof; rm -rf $O/prof; cat $O/host_gaps.md | head -80; tail -3 $O/host_gaps.err
