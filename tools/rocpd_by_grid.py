#!/usr/bin/env python3
"""Per (kernel, grid size) breakdown of a rocprofv3 (rocpd sqlite) kernel trace: one line per launch shape.
Usage: tools/rocpd_by_grid.py results.db [substring of the kernel name] [out.md]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_gemm"
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcols = [c for c in cols if c.lower() in ("grid_size", "grid_x", "grid_size_x", "grid")]
    gx = gcols[0] if gcols else None
    if gx is None:
        print("columns:", cols)
        return
    rows = cur.execute(f"select {name_col}, {gx}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels where {name_col} like ? group by {name_col}, {gx} order by 4 desc", (f"%{pat}%",)).fetchall()
    lines = ["| kernel | grid (threads) | calls | total ms | avg us | min us | max us |", "|---|---|---|---|---|---|---|"]
    for name, g, n, tot, avg, mn, mx in rows[:40]:
        short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0][:60]
        lines.append(f"| {short} | {g} | {n} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out + "\n")


if __name__ == "__main__":
    main()
