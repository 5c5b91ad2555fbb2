import sys, struct, numpy as np
raw = open(sys.argv[1], "rb").read()
off = 0
k = 0
while off < len(raw):
    hdr = np.frombuffer(raw, dtype=np.int64, count=8, offset=off); off += 64
    nwg, d, Dl, Dr, wl, wr, nsite, nrec = [int(x) for x in hdr]
    rec = np.frombuffer(raw, dtype=np.uint64, count=nrec, offset=off).reshape(-1, 8).astype(np.int64); off += nrec * 8
    ok = (rec[:, 4] > rec[:, 0]) & (rec[:, 4] - rec[:, 0] < 10_000_000)
    r = rec[ok]
    if len(r) == 0:
        print(k, "no records", nwg, d, nsite); k += 1; continue
    ph = np.stack([r[:, 1] - r[:, 0], r[:, 2] - r[:, 1], r[:, 3] - r[:, 2], r[:, 4] - r[:, 3], r[:, 4] - r[:, 0]], 1)
    wall = r[:, 6]
    print(f"launch {k}: nsite {nsite} Dl {Dl} Dr {Dr} w {wl}/{wr} grid {nwg} x {d}; waves recorded {len(r)} of {nwg * d * 4}; wall span of ends {(wall.max() - wall.min()) * 10} ns")
    names = ["flags", "step1", "barrier", "step2", "total"]
    for i, nm in enumerate(names):
        print(f"   {nm:8s} mean {ph[:, i].mean():8.0f}  p50 {np.median(ph[:, i]):8.0f}  p90 {np.percentile(ph[:, i], 90):8.0f}  max {ph[:, i].max():8.0f} cycles")
    for pc in sorted(set(r[:, 5])):
        m = r[:, 5] == pc
        print(f"   l tiles {pc:2d}: waves {m.sum():5d}  total mean {ph[m, 4].mean():8.0f} max {ph[m, 4].max():8.0f}")
    k += 1
