# same-box A/B of two builds of the library: bash tools/runs/lib_ab.sh <other .so relative to the repo> [which configs]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OTHER=$GRAFT_REPO_ROOT/$1; W=${2:-sbm holstein}
O=gpurun_out/lib_ab.jsonl; : > $O
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],1))'
python bench.py --steps 1 --warmup 0 --cpu-updates 0 --state-file /tmp/state.npz > /dev/null 2>&1
for r in 1 2 3; do
  for lib in other this; do
    if [ $lib = other ]; then export RENO_MPSENGINE=$OTHER; else unset RENO_MPSENGINE; fi
    echo "# $lib" >> $O
    timeout 300 python tools/small_ab.py $W 2>/tmp/err.log >> $O || tail -3 /tmp/err.log >> $O
    echo -n "# $lib headline: " >> $O
    python bench.py --steps 5 --warmup 2 --cpu-updates 0 --state-file /tmp/state.npz 2>/tmp/err.log | python -c "$P" >> $O || tail -3 /tmp/err.log >> $O
  done
done
grep -v energy $O; python - <<'PY'
import json
rows={}
lib=None
for ln in open("gpurun_out/lib_ab.jsonl"):
    if ln.startswith("# "):
        p=ln.split()
        lib=p[1]
        if "headline:" in ln and len(p)>=4: rows.setdefault((lib,"headline"),[]).append(float(p[3]))
        continue
    d=json.loads(ln); rows.setdefault((lib,d["config"]),[]).append(d["site_updates_per_s"])
for k in sorted(rows): print(k, [round(x,1) for x in rows[k]])
PY
