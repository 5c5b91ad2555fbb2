# side configurations: this tree against the round-4 tree (worktree .r4tree, built in place) on one box, alternating
O=gpurun_out/r5_vs_r4; mkdir -p $O; cd $GRAFT_REPO_ROOT
for r in 1 2; do
  (cd .r4tree && timeout 600 python tools/config_times.py ../$O/r4_$r.md > /dev/null 2>&1)
  (timeout 600 python tools/config_times.py $O/r5_$r.md > /dev/null 2>&1)
done
for f in r4_1 r5_1 r4_2 r5_2; do echo $f; cut -d'|' -f2-4 $O/$f.md | tail -8; done
