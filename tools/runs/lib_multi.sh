# headline A/B of several builds of the library on one box, alternating: bash tools/runs/lib_multi.sh lib1.so lib2.so ...   (paths relative to the repo)
cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
python bench.py --steps 1 --warmup 1 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for r in 1 2 3; do for l in "$@"; do echo -n "$l: "; RENO_MPSENGINE=$GRAFT_REPO_ROOT/$l python bench.py --steps 5 --warmup 3 --cpu-updates 0 --state-file /tmp/state.npz 2>/tmp/err.log | python -c "$P" || tail -3 /tmp/err.log; done; done
