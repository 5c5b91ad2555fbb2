#!/usr/bin/env python3
"""What the host was doing while the GPU sat idle: for every gap between consecutive kernels longer than a
threshold, the HIP API calls (rocprofv3 --hip-trace, rocpd sqlite) that overlap the gap are collected, grouped by
(previous kernel -> next kernel) and by API name; the part of a gap not covered by any HIP call is host code outside
the runtime (Python, the engine's own C++).
Usage: tools/rocpd_host_gaps.py results.db [out.md] [threshold_us=8] [first_kernel_regex]
(only the part of the trace from the first kernel matching the regex on is analysed - e.g. the timed evolves)"""
import bisect
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name.split("(")[0])
    return name[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    thr = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 8e3
    objs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    reg = "regions" if "regions" in objs else None
    if reg is None:
        print("no regions view; objects:", objs)
        return
    cols = [r[1] for r in cur.execute(f"pragma table_info({reg})")]
    apis = cur.execute(f"select name, start, end from {reg} where end > start order by start").fetchall()
    apis = [a for a in apis if a[0].startswith("hip")]
    starts = [a[1] for a in apis]
    kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in kcols else [c for c in kcols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    if len(sys.argv) > 4:
        rx = re.compile(sys.argv[4])
        first = next((i for i, r in enumerate(rows) if rx.search(r[0])), 0)
        rows = rows[first:]
    by_pair = defaultdict(lambda: [0, 0.0, defaultdict(lambda: [0, 0.0]), 0.0])
    prev_end, prev_name = None, None
    for name, s, e in rows:
        if prev_end is not None and thr < s - prev_end < 2e6:
            g0, g1 = prev_end, s
            rec = by_pair[(short(prev_name), short(name))]
            rec[0] += 1
            rec[1] += g1 - g0
            i = bisect.bisect_left(starts, g0 - 5e5)
            covered = []
            while i < len(apis) and apis[i][1] < g1:
                n, a0, a1 = apis[i]
                lo, hi = max(a0, g0), min(a1, g1)
                if hi > lo:
                    r = rec[2][n]
                    r[0] += 1
                    r[1] += hi - lo
                    covered.append((lo, hi))
                i += 1
            covered.sort()
            tot, cur_hi = 0.0, g0
            for lo, hi in covered:
                lo = max(lo, cur_hi)
                if hi > lo:
                    tot += hi - lo
                    cur_hi = hi
            rec[3] += (g1 - g0) - tot
        prev_end, prev_name = max(e, prev_end or 0), name
    lines = [f"gaps longer than {thr / 1e3:.0f} us, by (previous -> next) kernel; per gap class the HIP calls overlapping the "
             "gap (count, time inside the gap) and the time not inside any HIP call", ""]
    for (a, b), (n, t, calls, unc) in sorted(by_pair.items(), key=lambda x: -x[1][1])[:14]:
        lines.append(f"### {a} -> {b}: {n} gaps, {t / 1e6:.2f} ms total, {t / n / 1e3:.1f} us avg; outside HIP calls "
                     f"{unc / n / 1e3:.1f} us avg")
        for cn, (c, ct) in sorted(calls.items(), key=lambda x: -x[1][1])[:8]:
            lines.append(f"* {cn}: {c / n:.1f} calls / gap, {ct / n / 1e3:.1f} us / gap")
        lines.append("")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
