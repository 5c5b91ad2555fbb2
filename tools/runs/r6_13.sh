cd $GRAFT_REPO_ROOT
bash tools/runs/r6_final.sh r6_final all > gpurun_out/r6_final_stdout.txt 2>&1
tail -30 gpurun_out/r6_final_stdout.txt
python tools/oracle_threads_probe.py 2>&1 | grep threads | tee gpurun_out/r6_final/oracle_threads.txt
