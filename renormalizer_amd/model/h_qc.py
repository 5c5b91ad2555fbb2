"""Ab-initio electronic Hamiltonians: FCIDUMP reader and Jordan-Wigner model (kept API of
renormalizer/model/h_qc.py: ``read_fcidump`` :14-47, ``qc_model`` :127-195).

Spin orbital j = 2*spatial + spin lives on site j (a ``BasisHalfSpin`` with two conserved particle numbers,
sigmaqn [[0,0],[1,0]] for alpha / [[0,0],[0,1]] for beta sites; state 0 = empty).  With that convention
a_j = prod_{l<j} Z_l sigma^+_j and a_j^+ = prod_{l<j} Z_l sigma^-_j.  Products of the strings are handed to the
MPO builder as they are: it merges local operators that are proportional as matrices ("Z Z" = I,
"Z +" = -"+" ...), which is what the reference's symbolic ``simplify_op`` achieves."""
import itertools

import numpy as np

from .basis import BasisHalfSpin
from .op import Op


def read_fcidump(fname, norb):
    """Returns (spin-orbital one-electron matrix, antisymmetrised two-electron tensor, nuclear repulsion) in the
    convention H = sum_pq sh[p,q] a+_p a_q + sum_{p<q, r<s} aseri[p,q,r,s] a+_p a+_q a_r a_s."""
    eri = np.zeros((norb,) * 4)
    h = np.zeros((norb, norb))
    nuc = 0.0
    with open(fname) as f:
        lines = f.readlines()
    body = False
    for line in lines:
        if not body:
            if line.strip().upper().startswith("&END") or line.strip() == "/":
                body = True
            continue
        t = line.split()
        if len(t) != 5:
            continue
        val, p, q, r, s = float(t[0]), int(t[1]), int(t[2]), int(t[3]), int(t[4])
        if r != 0:
            for (a, b) in ((p, q), (q, p)):
                for (c, d) in ((r, s), (s, r)):
                    eri[a - 1, b - 1, c - 1, d - 1] = val
        elif p != 0:
            h[p - 1, q - 1] = h[q - 1, p - 1] = val
        else:
            nuc = val
    sh, aseri = int_to_h(h, eri)
    return sh, aseri, nuc


def int_to_h(h, eri):
    """Spatial (chemists' notation, as stored in FCIDUMP) to spin-orbital integrals, h_qc.py:50-70."""
    n = 2 * len(h)
    idx = np.arange(n)
    sp, spin = idx // 2, idx % 2
    sh = h[np.ix_(sp, sp)] * (spin[:, None] == spin[None, :])
    # seri[p,q,r,s] = (ps|qr) delta(spin p, spin s) delta(spin q, spin r)   for a+_p a+_q a_r a_s
    seri = eri[np.ix_(sp, sp, sp, sp)].transpose(0, 2, 3, 1)
    seri = seri * (spin[:, None, None, None] == spin[None, None, None, :]) * (spin[None, :, None, None] == spin[None, None, :, None])
    aseri = np.zeros((n,) * 4)
    for q, s in itertools.product(range(n), repeat=2):
        for p, r in itertools.product(range(q), range(s)):
            aseri[p, q, r, s] = seri[p, q, r, s] - seri[p, q, s, r]
    return sh, aseri


def _ladder(j, create, conserve_qn):
    sym = "-" if create else "+"
    if conserve_qn:
        q = [1, 0] if j % 2 == 0 else [0, 1]
        qn = [[0, 0]] * j + [q if create else [-q[0], -q[1]]]
    else:
        qn = [0] * (j + 1)
    return Op(" ".join(["Z"] * j + [sym]), list(range(j)) + [j], 1.0, qn)


def qc_model(h1e, h2e, conserve_qn=True):
    """(basis list, Hamiltonian terms) of the Jordan-Wigner transformed electronic Hamiltonian."""
    n = h1e.shape[0]
    assert h1e.shape == (n, n) and h2e.shape == (n, n, n, n)
    cre = [_ladder(j, True, conserve_qn) for j in range(n)]
    ann = [_ladder(j, False, conserve_qn) for j in range(n)]
    terms = []
    for p, q in np.argwhere(h1e != 0):
        terms.append(Op.product([cre[p], ann[q]]) * h1e[p, q])
    for p, q, r, s in np.argwhere(h2e != 0):
        terms.append(Op.product([cre[p], cre[q], ann[r], ann[s]]) * h2e[p, q, r, s])
    basis = []
    for j in range(n):
        if conserve_qn:
            sigmaqn = np.array([[0, 0], [1, 0]]) if j % 2 == 0 else np.array([[0, 0], [0, 1]])
        else:
            sigmaqn = [0, 0]
        basis.append(BasisHalfSpin(j, sigmaqn=sigmaqn))
    return basis, terms
