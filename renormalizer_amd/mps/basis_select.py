"""Which renormalised basis states to keep (counterpart of ``select_basis`` in renormalizer/mps/lib.py:253-322): an
optional equal quota per quantum-number block, then the remaining slots by descending singular value (stable on
ties).  The index selection is ``mpse_truncate_select`` of the engine library (host-side integer logic, no device
work); the column copies are done on the device (mpse_gather_cols)."""
import ctypes as C

import numpy as np

from ..engine import load_library

_LIB = None


def select_basis_indices(sset, qnlist, Mmax, percent=0.0):
    global _LIB
    if _LIB is None:
        _LIB = load_library()
    sset = np.ascontiguousarray(sset, dtype=np.float64)
    n = len(sset)
    ids = None
    if percent != 0:
        # block id of every state = rank of its quantum number among the distinct ones in lexicographic order (rows of an
        # integer array: np.unique sorts them that way)
        qn_a = np.asarray(qnlist, dtype=np.int64).reshape(n, -1)
        _, inv = np.unique(qn_a, axis=0, return_inverse=True)
        ids = np.ascontiguousarray(np.asarray(inv).reshape(-1), dtype=np.int64)
    picked = np.empty(max(n, 1), dtype=np.int64)
    npicked = C.c_int64(0)
    st = _LIB.mpse_truncate_select(sset.ctypes.data_as(C.POINTER(C.c_double)),
                                   None if ids is None else ids.ctypes.data_as(C.POINTER(C.c_int64)), n, int(Mmax),
                                   float(percent), picked.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(npicked))
    if st != 0:
        raise ValueError(f"mpse_truncate_select failed with status {st}")
    out = picked[:npicked.value].tolist()
    assert len(set(out)) == len(out)
    return out
