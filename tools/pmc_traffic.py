#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (rocpd sqlite): FETCH_SIZE and WRITE_SIZE.

Usage: tools/pmc_traffic.py fetch.db write.db out.json [out.md]

Both counters are reported in KB by rocprofv3.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced reads, so the fetch figure is doubled; WRITE_SIZE is taken as is.
The passes are separate runs of the same command (FETCH_SIZE and WRITE_SIZE do not fit one pass), launches are
matched by kernel name and averaged."""
import hashlib
import json
import os
import re
import sqlite3
import sys


def kernel_source_sha():
    """Fingerprint of the contraction kernel's sources at the time of the counter passes (bench.py quotes the figure
    as `roofline.traffic` only while the sources still match)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "renormalizer_amd", "csrc")
    h = hashlib.sha256()
    for name in ("mpse_gemm.hip", "mpse_plans.h", "mpse_contract.hip"):
        with open(os.path.join(root, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    out = {}
    q = "select name, count(*), avg(counter_value), avg(duration) from pmc_events where counter_name = ? group by name"
    for name, n, avg, dur in db.execute(q, (counter,)):
        short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        out[short] = dict(launches=n, avg_kb=avg, avg_us_under_pmc=dur / 1e3)
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k), write.get(k)
        if not f or not w:
            continue
        res[k] = dict(launches=f["launches"], fetch_raw_bytes=f["avg_kb"] * 1024.0, write_bytes=w["avg_kb"] * 1024.0,
                      hbm_bytes_per_launch=2.0 * f["avg_kb"] * 1024.0 + w["avg_kb"] * 1024.0)
    json.dump(dict(note="avg per launch; hbm_bytes = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE",
                   kernel_source_sha=kernel_source_sha(), kernels=res),
              open(sys.argv[3], "w"), indent=1)
    lines = ["| kernel | launches | FETCH_SIZE raw MB | WRITE_SIZE MB | HBM MB / launch (2 x fetch + write) |", "|---|---|---|---|---|"]
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
        lines.append(f"| {k[:70]} | {v['launches']} | {v['fetch_raw_bytes'] / 1e6:.2f} | {v['write_bytes'] / 1e6:.2f} | "
                     f"{v['hbm_bytes_per_launch'] / 1e6:.2f} |")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
