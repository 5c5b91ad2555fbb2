"""Host-side integer / bookkeeping helpers for choosing which renormalised basis states to keep.

Counterpart of ``select_basis`` in renormalizer/mps/lib.py:253-322: an optional equal quota per
quantum-number block, then the remaining slots by descending singular value (stable on ties).
Only the index selection happens here; the column copies are done on the device
(mpse_gather_cols)."""
import numpy as np


def select_basis_indices(sset, qnlist, Mmax, percent=0.0):
    sset = np.asarray(sset, dtype=float)
    qn_t = [tuple(int(x) for x in np.atleast_1d(qn)) for qn in qnlist]
    remaining = list(range(len(qn_t)))
    nbasis = min(len(remaining), int(Mmax))
    picked = []
    if percent != 0:
        blocks = sorted(set(qn_t))
        per_block = int(nbasis * percent / len(blocks))
        for b in blocks:
            members = [i for i in remaining if qn_t[i] == b]
            members.sort(key=lambda i: -sset[i])
            take = members[: min(per_block, len(members))]
            picked += take
            taken = set(take)
            remaining = [i for i in remaining if i not in taken]
    rest = nbasis - len(picked)
    remaining.sort(key=lambda i: -sset[i])
    picked += remaining[:rest]
    assert len(set(picked)) == len(picked)
    return picked
