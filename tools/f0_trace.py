#!/usr/bin/env python3
"""Per-workgroup timeline of k_heff0_fused (fused bond / two-level-site matvec), from the debug trace of the engine:

    MPSE_GEMM_TRACE=trace.bin MPSE_GEMM_TRACE_ONLY=f0 python bench.py --steps 1 --warmup 2 --cpu-updates 0
    python tools/f0_trace.py trace.bin [out.md]

A record per workgroup: entry, flags in hand (first operand loads can go out), end of step 1 (last MFMA of T issued), end of
step 2, exit - shader cycles - plus the device-wide 100 MHz clock at entry and exit, the number of c tiles of step 1 and
of l tiles of step 2 the workgroup had, and where it ran.  Launches are told apart by the 100 MHz clock (a gap of more
than 2 us between the entry of one workgroup and the next, sorted).  Per launch class (grid, d):
  makespan            first entry -> last exit (100 MHz clock)
  mean / max life     of the workgroups that had work
  phase medians       prelude (entry -> flags), step 1, barrier + step 2, epilogue - for the HEAVIEST decile by life
  slots               how many workgroups were alive at once (max), how late the last workgroup ENTERED"""
import sys

import numpy as np

r = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 10)
r = r[(r[:, 1] >> np.uint64(63)) == 1]
grid = (r[:, 0] >> np.uint64(32)).astype(np.int64)
d = ((r[:, 1] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
nct = (r[:, 2] & np.uint64(0xFFFF)).astype(np.int64)
nlt = ((r[:, 2] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
t = r[:, 3:8].astype(np.int64)
rt0, rt1 = r[:, 8].astype(np.int64), r[:, 9].astype(np.int64)
order = np.argsort(rt0, kind="stable")
launch = np.zeros(len(r), dtype=np.int64)
launch[order] = np.cumsum(np.concatenate([[0], (np.diff(rt0[order]) > 200).astype(np.int64)]))   # 200 ticks = 2 us
lines = ["| grid x d | launches | WGs / launch | with work % | makespan us | mean life us (work) | max life us | last entry after us | "
         "max alive | heaviest decile: prelude | step 1 | step 2 | epilogue (cycles) | c tiles | l tiles | clock GHz |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for key in sorted(set(zip(grid.tolist(), d.tolist()))):
    m = (grid == key[0]) & (d == key[1])
    ids = np.unique(launch[m])
    rows = []
    for L in ids:
        k = m & (launch == L)
        if k.sum() < key[0] * key[1] // 2:
            continue                                   # (a launch cut by the start / end of the traced region)
        work = k & (nlt > 0)
        life = (rt1 - rt0) / 100.0
        ms = (rt1[k].max() - rt0[k].min()) / 100.0
        ev = np.concatenate([np.stack([rt0[k], np.ones(k.sum())], 1), np.stack([rt1[k], -np.ones(k.sum())], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
        alive = np.cumsum(ev[:, 1]).max()
        heavy = work & (life >= np.quantile(life[work], 0.9)) if work.sum() else work
        ph = [np.median(t[heavy, i + 1] - t[heavy, i]) if heavy.sum() else 0 for i in range(4)]
        ghz = np.median((t[work, 4] - t[work, 0]) / np.maximum(1, (rt1[work] - rt0[work]) * 10.0)) if work.sum() else 0
        rows.append((k.sum(), 100.0 * work.sum() / k.sum(), ms, life[work].mean() if work.sum() else 0, life[k].max(),
                     (rt0[k].max() - rt0[k].min()) / 100.0, alive, *ph, nct[heavy].mean() if heavy.sum() else 0,
                     nlt[heavy].mean() if heavy.sum() else 0, ghz))
    if not rows:
        continue
    a = np.median(np.array(rows), axis=0)
    lines.append(f"| {key[0]} x {key[1]} | {len(rows)} | {a[0]:.0f} | {a[1]:.0f} | {a[2]:.1f} | {a[3]:.1f} | {a[4]:.1f} | {a[5]:.1f} | "
                 f"{a[6]:.0f} | {a[7]:.0f} | {a[8]:.0f} | {a[9]:.0f} | {a[10]:.0f} | {a[11]:.1f} | {a[12]:.1f} | {a[13]:.2f} |")
# Does a heavy workgroup run slower when another WORKING workgroup shares its compute unit?  Heavy = top quartile of
# (c tiles + 4 l tiles) within its launch; "shared" = another workgroup with work on the same (die, HW_ID[15:8]) whose life
# overlaps more than half of this one's.
hwid = (r[:, 2] >> np.uint64(32)).astype(np.int64)
cu = ((hwid >> 16) & 0xF) * 256 + ((hwid >> 8) & 0xFF)
lines += ["", "| grid x d | heavy workgroups | alone on their CU: n, median life us | sharing it with a working workgroup: n, median life us | CUs with 0 / 1 / 2+ working workgroups (median per launch) |",
          "|---|---|---|---|---|"]
for key in sorted(set(zip(grid.tolist(), d.tolist()))):
    m = (grid == key[0]) & (d == key[1])
    alone, shared, occ = [], [], []
    for L in np.unique(launch[m])[:60]:
        k = np.nonzero(m & (launch == L))[0]
        if len(k) < key[0] * key[1] // 2:
            continue
        w = nct[k] + 4 * nlt[k]
        work = nlt[k] > 0
        if work.sum() < 4:
            continue
        heavy = work & (w >= np.quantile(w[work], 0.75))
        cnt = np.bincount(cu[k][work], minlength=4096)
        used = np.unique(cu[k])
        occ.append((int((cnt[used] == 0).sum()), int((cnt[used] == 1).sum()), int((cnt[used] >= 2).sum())))
        for i in np.nonzero(heavy)[0]:
            same = (cu[k] == cu[k][i]) & work
            same[i] = False
            ov = np.minimum(rt1[k][same], rt1[k][i]) - np.maximum(rt0[k][same], rt0[k][i])
            life_i = (rt1[k][i] - rt0[k][i]) / 100.0
            (shared if (ov > 0.5 * (rt1[k][i] - rt0[k][i])).any() else alone).append(life_i)
    if alone or shared:
        o = np.median(np.array(occ), axis=0) if occ else (0, 0, 0)
        lines.append(f"| {key[0]} x {key[1]} | {len(alone) + len(shared)} | {len(alone)}, {np.median(alone) if alone else 0:.1f} | "
                     f"{len(shared)}, {np.median(shared) if shared else 0:.1f} | {o[0]:.0f} / {o[1]:.0f} / {o[2]:.0f} |")
# What a workgroup's life is made of: least squares  life = fixed + a (c tiles of step 1) + b (l tiles of step 2)  over the
# workgroups with work (a c tile = 12 MFMAs per wave = 768 MFMA cycles; an l tile, dealt to four waves, 48 MFMAs of one
# wave = 3072 cycles, i.e. 768 per l tile of the workgroup)
lines += ["", "| grid x d | workgroups fitted | fixed us | per c tile us | per l tile us | residual rms us |", "|---|---|---|---|---|---|"]
for key in sorted(set(zip(grid.tolist(), d.tolist()))):
    m = (grid == key[0]) & (d == key[1]) & (nlt > 0)
    if m.sum() < 100:
        continue
    A = np.stack([np.ones(m.sum()), nct[m], nlt[m]], 1).astype(float)
    y = (rt1[m] - rt0[m]) / 100.0
    coef, *_ = np.linalg.lstsq(A, y, rcond=None)
    res = np.sqrt(np.mean((A @ coef - y) ** 2))
    lines.append(f"| {key[0]} x {key[1]} | {m.sum()} | {coef[0]:.2f} | {coef[1]:.3f} | {coef[2]:.3f} | {res:.2f} |")
out = "\n".join(lines) + f"\n\n{len(r)} workgroup records of k_heff0_fused; medians over the launches of a class\n"
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
