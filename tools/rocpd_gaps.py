#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 (rocpd sqlite) trace, grouped by (previous -> next)
kernel: where a dependent chain of small launches leaves the GPU waiting.
Usage: tools/rocpd_gaps.py results.db [out.md]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name.split("(")[0])
    return name[:48]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    pair = defaultdict(lambda: [0, 0.0])
    after = defaultdict(lambda: [0, 0.0])
    busy = 0.0
    idle = 0.0
    big = 0.0
    prev_end, prev_name = None, None
    for name, s, e in rows:
        busy += e - s
        if prev_end is not None:
            g = max(0, s - prev_end)
            if g > 2e6:                      # > 2 ms: between timed steps / host phases, not the launch chain
                big += g
            else:
                idle += g
                p = pair[(short(prev_name), short(name))]
                p[0] += 1
                p[1] += g
                a = after[short(prev_name)]
                a[0] += 1
                a[1] += g
        prev_end, prev_name = max(e, prev_end or 0), name
    lines = [f"{len(rows)} dispatches: busy {busy / 1e6:.1f} ms, idle between kernels {idle / 1e6:.1f} ms "
             f"(avg {idle / max(1, len(rows)) / 1e3:.2f} us per dispatch), gaps > 2 ms excluded {big / 1e6:.1f} ms", "",
             "| previous kernel | gaps | total ms | avg us |", "|---|---|---|---|"]
    for k, (n, t) in sorted(after.items(), key=lambda x: -x[1][1])[:16]:
        lines.append(f"| {k} | {n} | {t / 1e6:.2f} | {t / n / 1e3:.2f} |")
    lines += ["", "| previous -> next | gaps | total ms | avg us |", "|---|---|---|---|"]
    for (a, b), (n, t) in sorted(pair.items(), key=lambda x: -x[1][1])[:28]:
        lines.append(f"| {a} -> {b} | {n} | {t / 1e6:.2f} | {t / n / 1e3:.2f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
