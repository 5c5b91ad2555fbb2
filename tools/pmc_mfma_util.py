#!/usr/bin/env python3
"""MFMA utilisation per contraction shape from one rocprofv3 PMC pass over tools/gemm_bench.py.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d out -o p -- python tools/gemm_bench.py
    python tools/pmc_mfma_util.py out/.../p_results.db [out.md]

util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the busy counter sums over all SIMDs, the
GUI-active counter sums the 8 XCDs' active cycles.  Launches are grouped by (kernel, grid size) = one GEMM shape."""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    per = defaultdict(dict)
    # one row per hardware instance (XCD / shader engine): sum them per dispatch and counter
    q = ("select kernel_name, grid_size, dispatch_id, counter_name, sum(value), max(duration) from counters_collection "
         "group by dispatch_id, counter_name")
    for name, grid, disp, cname, val, dur in db.execute(q):
        if "k_gemm" not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        d = per[(short, grid, disp)]
        d[cname] = val
        d["dur"] = dur
    groups = defaultdict(list)
    for (short, grid, _), d in per.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
            groups[(short, grid)].append(d)
    lines = ["| kernel | grid (threads) | launches | dur us (under PMC) | MFMA busy Mcyc | MFMA util |", "|---|---|---|---|---|---|"]
    for (short, grid), ds in sorted(groups.items(), key=lambda kv: -sum(x["dur"] for x in kv[1])):
        busy = sorted(x["SQ_VALU_MFMA_BUSY_CYCLES"] for x in ds)[len(ds) // 2]
        act = sorted(x["GRBM_GUI_ACTIVE"] for x in ds)[len(ds) // 2]
        dur = sorted(x["dur"] for x in ds)[len(ds) // 2]
        lines.append(f"| {short} | {grid} | {len(ds)} | {dur / 1e3:.1f} | {busy / 1e6:.1f} | {busy / (1024 * act / 8):.3f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
