#!/usr/bin/env python3
"""Per-workgroup timeline of the contraction kernel (MPSE_GEMM_TRACE=<file> python bench.py ...): phases of a
workgroup's life in shader cycles, by launch class (grid size, operand types, K).
   setup = entry -> first K tile chosen (index arithmetic, tile masks into LDS)
   first = -> first operand tile staged (global load latency + LDS write + barrier)
   loop  = -> last MFMA of the K loop
   epi   = -> exit (3M combine, alpha/beta, strided stores, dot partials)
Usage: tools/gemm_trace.py trace.bin [out.md]"""
import sys

import numpy as np

r = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 10)
grid = (r[:, 0] >> np.uint64(32)).astype(np.int64)
ca = ((r[:, 1] >> np.uint64(62)) & np.uint64(1)).astype(int)
cb = ((r[:, 1] >> np.uint64(61)) & np.uint64(1)).astype(int)
ks = ((r[:, 1] >> np.uint64(40)) & np.uint64(0xFFFF)).astype(int)
K = ((r[:, 1] >> np.uint64(8)) & np.uint64(0xFFFFFFFF)).astype(np.int64)
kt = (r[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
t0, t1, t2, t3, t4 = (r[:, i].astype(np.int64) for i in range(3, 8))
lines = ["| grid (WGs) | types | K | ksplit | workgroups | empty % | K tiles / WG (non-empty) | setup | first | loop | per K tile | epi | total (non-empty) | total (empty) |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
keys = sorted(set(zip(grid.tolist(), ca.tolist(), cb.tolist(), K.tolist(), ks.tolist())), key=lambda k: -np.sum((grid == k[0]) & (K == k[3])))
for g, a, b, k, s in keys[:24]:
    m = (grid == g) & (ca == a) & (cb == b) & (K == k) & (ks == s)
    ne = m & (kt > 0)
    em = m & (kt == 0)
    if ne.sum() == 0:
        continue
    setup = np.median(t1[ne] - t0[ne])
    first = np.median(t2[ne] - t1[ne])
    loop = np.median(t3[ne] - t2[ne])
    epi = np.median(t4[ne] - t3[ne])
    per = np.median((t3[ne] - t2[ne]) / kt[ne])
    tot = np.median(t4[ne] - t0[ne])
    tote = np.median(t4[em] - t0[em]) if em.sum() else 0
    lines.append(f"| {g} | {'c' if a else 'r'}x{'c' if b else 'r'} | {k} | {s} | {m.sum()} | {100 * em.sum() / m.sum():.0f} | {kt[ne].mean():.1f} | "
                 f"{setup:.0f} | {first:.0f} | {loop:.0f} | {per:.0f} | {epi:.0f} | {tot:.0f} | {tote:.0f} |")
out = "\n".join(lines) + f"\n\n{len(r)} workgroup records; cycles = s_memtime (shader clock)\n"
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
