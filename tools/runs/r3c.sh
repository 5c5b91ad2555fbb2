mkdir -p gpurun_out/r3c; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3c
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1), round(d["config"]["mean_krylov_dim"],3), round(d["roofline"]["visited_ktile_share"],3), d["roofline"]["timed_launches"])'
python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/dev/null | python -c "$P"
python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/dev/null | python -c "$P"
python bench.py --steps 5 --warmup 2 --cpu-updates 0 2>/dev/null | python -c "$P"
python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/dev/null | python -c "$P"
python bench.py --steps 5 --warmup 2 --cpu-updates 0 2>/dev/null | python -c "$P"
