#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
Usage: tools/rocpd_summary.py results.db [out.md]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = short.split("(")[0][:80]
        lines.append(f"| {short} | {n} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.1f} |")
    lines.append(f"\ntotal kernel time {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
    # bench.py's roofline times one c128 x c128 contraction call = k_gemm<true, true, *> plus, for split-K calls,
    # its k_splitk_reduce<true>; the matching trace figure is (their summed time) / (number of k_gemm launches)
    g = [r for r in rows if re.search(r"k_gemm<true, true, (true|false)>", r[0])]
    red = [r for r in rows if "k_splitk_reduce<true>" in r[0]]
    occ = [r for r in rows if "k_tile_occ" in r[0]]
    if g:
        ng = sum(r[1] for r in g)
        tg = sum(r[2] for r in g)
        tr = sum(r[2] for r in red)
        to = sum(r[2] for r in occ)
        lines.append(f"\nc128 x c128 contraction calls: {ng}; k_gemm alone {tg / ng / 1e3:.2f} us/call; "
                     f"with split-K reduce {(tg + tr) / ng / 1e3:.2f} us/call; with reduce and occupancy scan "
                     f"{(tg + tr + to) / ng / 1e3:.2f} us/call (bench.py roofline.avg_launch_ms brackets all three with "
                     f"HIP events, which add ~2 us per side)")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
