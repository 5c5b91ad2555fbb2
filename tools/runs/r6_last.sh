# last pass of round 6: trace + PMC on the final sources, their summaries put in place under profiles/, THEN the driver's bench command (its line quotes those files)
O=gpurun_out/r6_last; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/runs/r6_final.sh r6_last/final trace > /dev/null 2>&1
bash tools/runs/r6_final.sh r6_last/final pmc > /dev/null 2>&1
F=$O/final
cp $F/kernel_time.json profiles/r06_kernel_time.json; cp $F/pmc_traffic.json profiles/r06_pmc_traffic.json; cp $F/pmc_mfma_busy.json profiles/r06_pmc_mfma_busy.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --state-file /tmp/s.npz > $O/bench_20steps_b.json 2>> $O/bench20.err
python - <<'PY'
import json
for f in ("bench_20steps.json", "bench_20steps_b.json", "bench_default.json"):
    d = json.loads(open("gpurun_out/r6_last/" + f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, round(d["value"], 1), round(d["ms_per_step"], 1), "frac", round(r["frac"], 3), "frac_kernel_time", r.get("frac_kernel_time"), "traffic", r.get("traffic"))
    for c in r["classes"]:
        if "heff0" in c["kernel"]:
            print("   fused:", c.get("dense_equiv_frac"), c.get("mfma_busy"), c.get("mfma_busy_gui_active"))
PY
python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
