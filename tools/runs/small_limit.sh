# element limit of the one-launch matvec: MPSE_SMALL sweep on the launch-bound configurations and the headline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/small_limit.txt; : > $O
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for r in 1 2; do for v in 0 32768 65536 131072 262144; do
  export MPSE_SMALL=$v; echo "# MPSE_SMALL=$v" >> $O
  timeout 300 python tools/small_ab.py sbm holstein holstein128 2>/tmp/err.log | python -c 'import sys,json
for l in sys.stdin: d=json.loads(l); print(d["config"], round(d["site_updates_per_s"],1))' >> $O || tail -3 /tmp/err.log >> $O
  echo -n "headline: " >> $O; python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/tmp/err.log | python -c "$P" >> $O || tail -3 /tmp/err.log >> $O
done; done
cat $O
