#!/usr/bin/env python3
"""Average duration of the dominant kernel from a rocprofv3 kernel trace (rocpd sqlite) of bench.py's timed region,
as a small JSON that bench.py quotes for `roofline.frac_kernel_time` (the same way `roofline.traffic` is sourced: a
committed measurement with the fingerprint of the kernel sources it was taken on).

    tools/rocpd_kernel_time.py results.db out.json

c128 x c128 contraction: every instantiation of k_gemm<true, true, ...>; `avg_us` = their summed duration / launches
(the kernel alone), `avg_us_with_reduce` adds the k_splitk_reduce<true> launches that complete split products - the
quantity bench.py's HIP-event bracket spans."""
import json
import re
import sqlite3
import sys

from pmc_traffic import kernel_source_sha


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start) from kernels group by {name_col}").fetchall()
    g = [r for r in rows if re.search(r"k_gemm<true, true, (true|false)", r[0])]
    red = [r for r in rows if "k_splitk_reduce<true>" in r[0]]
    ng, tg, tr = sum(r[1] for r in g), sum(r[2] for r in g), sum(r[2] for r in red)
    total = sum(r[2] for r in rows)
    out = dict(kernel="k_gemm<c128,c128> (all instantiations)", launches=ng, total_ms=tg / 1e6, avg_us=tg / ng / 1e3,
               avg_us_with_reduce=(tg + tr) / ng / 1e3, share_of_kernel_time=tg / total, all_kernels_ms=total / 1e6,
               dispatches=sum(r[1] for r in rows), kernel_source_sha=kernel_source_sha(),
               source="rocprofv3 --kernel-trace --marker-trace --stats --selected-regions -- python bench.py --cpu-updates 0 "
                      "--steps 2 (the timed region only)")
    with open(sys.argv[2], "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
