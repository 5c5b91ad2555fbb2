#!/usr/bin/env python3
"""Load balance of the contraction launches from the per-workgroup timeline (MPSE_GEMM_TRACE=<file>): the record
stream is cut into launches (an in-order stream: the records of one launch are contiguous, one per workgroup), and per
launch class the makespan is set against the work: how evenly the non-empty workgroups and their K tiles land on the
compute units (HW_ID / XCC_ID of each workgroup).
Usage: tools/gemm_balance.py trace.bin [out.md]"""
import sys
from collections import defaultdict

import numpy as np

r = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 10)
grid = (r[:, 0] >> np.uint64(32)).astype(np.int64)
ca = ((r[:, 1] >> np.uint64(62)) & np.uint64(1)).astype(int)
cb = ((r[:, 1] >> np.uint64(61)) & np.uint64(1)).astype(int)
ks = ((r[:, 1] >> np.uint64(40)) & np.uint64(0xFFFF)).astype(int)
K = ((r[:, 1] >> np.uint64(8)) & np.uint64(0xFFFFFFFF)).astype(np.int64)
kt = (r[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
hw = (r[:, 2] >> np.uint64(32)).astype(np.int64)
xcc = (hw >> 16) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu          # unique per compute unit
t0, t4 = r[:, 8].astype(np.int64) * 24, r[:, 9].astype(np.int64) * 24   # 100 MHz device clock -> ~shader cycles (2.4 GHz)
stats = defaultdict(list)
i, n = 0, len(r)
while i < n:
    g = int(grid[i])
    j = i + g
    if j > n or not np.all(grid[i:j] == g):
        i += 1            # (a launch cut by the capacity of the trace buffer)
        continue
    sl = slice(i, j)
    key = (g, ca[i], cb[i], int(K[i]), ks[i])
    span = t4[sl].max() - t0[sl].min()
    busy = (t4[sl] - t0[sl]).sum()
    ne = kt[sl] > 0
    tiles_cu = np.bincount(cuid[sl], weights=kt[sl], minlength=1)
    wg_cu = np.bincount(cuid[sl][ne], minlength=1) if ne.any() else np.zeros(1)
    tiles_x = np.bincount(xcc[sl], weights=kt[sl], minlength=8)
    stats[key].append((span, busy, kt[sl].sum(), ne.sum(), tiles_cu.max(), (tiles_cu > 0).sum(), wg_cu.max(),
                       tiles_x.max(), tiles_x.min(), len(set(cuid[sl].tolist()))))
    i = j
lines = ["| grid (WGs) | types | K | ksplit | launches | makespan (cycles) | sum WG time / (CUs used x makespan) | K tiles | non-empty WGs | max K tiles on one CU | mean K tiles per used CU | max non-empty WGs on one CU | K tiles on fullest / emptiest XCD | CUs seen |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for key, v in sorted(stats.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:16]:
    a = np.array(v, dtype=float)
    m = a.mean(axis=0)
    g, A, B, k, s = key
    lines.append(f"| {g} | {'c' if A else 'r'}x{'c' if B else 'r'} | {k} | {s} | {len(v)} | {m[0]:.0f} | {m[1] / (m[9] * m[0]):.2f} | {m[2]:.0f} | "
                 f"{m[3]:.0f} | {m[4]:.1f} | {m[2] / max(m[5], 1):.1f} | {m[6]:.1f} | {m[7]:.0f} / {m[8]:.0f} | {m[9]:.0f} |")
bid = (r[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
big = grid >= 256
rr = 100.0 * np.mean(xcc[big] == (bid[big] % 8)) if big.any() else 0.0
out = "\n".join(lines) + f"\n\n{len(r)} workgroup records; die of a workgroup == blockIdx mod 8 for {rr:.1f} % of the workgroups of launches with >= 256 workgroups\n"
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
