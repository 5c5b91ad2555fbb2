# last pass of round 6 on the final tree: r6_last.sh (trace + PMC -> profiles/, bench lines, full GPU suite) and the two-site part
bash tools/runs/r6_last.sh
bash tools/runs/r6_final.sh r6_last/final svd > /dev/null 2>&1
cut -c1-160 gpurun_out/r6_last/final/bench_ps2.json
