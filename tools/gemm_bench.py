#!/usr/bin/env python3
"""Per-shape timing of the FP64-MFMA contraction kernel on the headline matvec shapes (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renormalizer_amd import engine as E  # noqa: E402


def run(eng, name, A, B, C, desc_args, reps=10, **kw):
    for _ in range(2):
        eng.gemm(A, B, C, *desc_args, **kw)
    eng.prof_reset()
    eng.prof_enable(True)
    for _ in range(reps):
        eng.gemm(A, B, C, *desc_args, **kw)
    eng.prof_enable(False)
    p = eng.prof_get()
    for k, v in p.items():
        if v["launches"]:
            ms = v["ms"] / v["launches"]
            print(f"{name:34s} {k:10s} {ms*1e3:9.1f} us  {v['flops']/v['launches']/ms/1e9:8.2f} TF/s  "
                  f"{v['bytes']/v['launches']/ms/1e9*1e3/1e3:8.2f} TB/s(compulsory)")


def main():
    eng = E.get_engine()
    rng = np.random.default_rng(0)
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 256

    def dev(shape, cplx=True):
        a = rng.standard_normal(shape)
        if cplx:
            a = a + 1j * rng.standard_normal(shape)
        return eng.asdevice(a)
    i1, i2 = E.idx1, E.idx2
    for (d, wl, wr) in ((16, 5, 4), (2, 4, 5)):
        L, Cc, R = dev((D, wl, D)), dev((D, d, D)), dev((D, wr, D))
        W = dev((wl, d, d, wr), False)
        T1 = eng.empty((D, wl, d, D), np.complex128)
        T2 = eng.empty((D, d, wr, D), np.complex128)
        out = eng.empty((D, d, D), np.complex128)
        N = d * D
        run(eng, f"A d={d}: ({D*wl}x{D})x({D}x{N})", L, Cc, T1,
            (i1(D * wl, D), i1(D, 1), i1(D, N), i1(N, 1), i1(D * wl, N), i1(N, 1)))
        run(eng, f"B d={d}: {D}x[({d*wr}x{wl*d})x({wl*d}x{D})]", W, T1, T2,
            (i2(d, wr, d * wr, 1), i2(wl, d, d * d * wr, wr), i1(wl * d, D), i1(D, 1), i1(d * wr, D), i1(D, 1)),
            batch=D, sb_a=0, sb_b=wl * d * D, sb_c=d * wr * D)
        run(eng, f"C d={d}: ({D*d}x{wr*D})x({wr*D}x{D})", T2, R, out,
            (i1(D * d, wr * D), i1(wr * D, 1), i1(wr * D, 1), i1(D, wr * D), i1(D * d, D), i1(D, 1)))
    # 0-site
    wl = 5
    L, S, R = dev((D, wl, D)), dev((D, D)), dev((D, wl, D))
    T = eng.empty((D, wl, D), np.complex128)
    out = eng.empty((D, D), np.complex128)
    run(eng, f"0-site A: ({D*wl}x{D})x({D}x{D})", L, S, T, (i1(D * wl, D), i1(D, 1), i1(D, D), i1(D, 1), i1(D * wl, D), i1(D, 1)))
    run(eng, f"0-site C: ({D}x{wl*D})x({wl*D}x{D})", T, R, out,
        (i1(D, wl * D), i1(wl * D, 1), i1(wl * D, 1), i1(D, wl * D), i1(D, D), i1(D, 1)))
    # square reference shapes
    for n in (1024, 2048, 4096):
        A, B = dev((n, n)), dev((n, n))
        Cm = eng.empty((n, n), np.complex128)
        run(eng, f"square zgemm {n}", A, B, Cm, (i1(n, n), i1(n, 1), i1(n, n), i1(n, 1), i1(n, n), i1(n, 1)), reps=3)
        A, B = dev((n, n), False), dev((n, n), False)
        Cm = eng.empty((n, n), np.float64)
        run(eng, f"square dgemm {n}", A, B, Cm, (i1(n, n), i1(n, 1), i1(n, n), i1(n, 1), i1(n, n), i1(n, 1)), reps=3)


if __name__ == "__main__":
    main()
