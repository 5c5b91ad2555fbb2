# side configurations, round-5 tree (_r5tree/, built from git bed6be3, not committed) against the round-6 tree, alternating on one box
O=gpurun_out/r6_vs_r5; mkdir -p $O; cd $GRAFT_REPO_ROOT
for i in 1 2; do
  (cd _r5tree && python tools/config_times.py 2>/dev/null | grep "^| #" | sed "s/^/r5 /") | tee -a $O/configs.txt
  (python tools/config_times.py 2>/dev/null | grep "^| #" | sed "s/^/r6 /") | tee -a $O/configs.txt
done
