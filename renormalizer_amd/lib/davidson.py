"""Davidson eigensolver (counterpart of ``davidson`` in renormalizer/lib/davidson/davidson.py:73-441 as called from
mps/gs.py:533-538).  The whole iteration runs inside the engine (``mpse_davidson``, include/mpsengine.h): basis
vectors, images and residuals stay in HBM, the subspace matrix grows by one batched reduction per new vector."""
import ctypes as C

import numpy as np

from ..engine import get_engine
from ..mps.hop_expr import Hop


def _solve(hop, guesses, hdiag, nroots, mask, tol, max_cycle, max_space, lindep, shift):
    if not isinstance(hop, Hop):
        raise TypeError("davidson: the operator must be an effective-Hamiltonian closure from hop_expr")
    eng = get_engine()
    cplx = hop.operator_is_complex or any(g.is_complex for g in guesses)
    dt = np.complex128 if cplx else np.float64
    n = int(np.prod(hop.cshape))
    stack = eng.empty((len(guesses), n), dt)
    for i, g in enumerate(guesses):
        g = g.to_complex() if cplx else g
        assert g.size == n
        eng._check(eng.lib.mpse_memcpy_d2d(eng.ctx, stack.ptr + i * n * stack.dtype.itemsize, g.ptr, g.nbytes))
    out = eng.empty((nroots, n), dt)
    e = (C.c_double * nroots)()
    ncyc, nmv = C.c_int(), C.c_int()
    eng._check(eng.lib.mpse_davidson(eng.ctx, stack.code, C.byref(hop.heff), int(hop.twolayer), hdiag.ptr,
                                     None if mask is None else mask.ptr, nroots, len(guesses), stack.ptr, tol, max_cycle,
                                     0 if max_space is None else max_space, lindep, shift, e, out.ptr, C.byref(ncyc),
                                     C.byref(nmv)))
    es = [float(x) for x in e]
    xs = [out.row_block(r, r + 1).reshape(hop.cshape) for r in range(nroots) if not np.isnan(es[r])]
    return [x for x in es if not np.isnan(x)], xs, ncyc.value


def davidson(hop, x0, hdiag, mask=None, tol=1e-12, max_cycle=100, max_space=12, lindep=1e-14, shift=1e-4):
    """Lowest eigenpair of the effective Hamiltonian ``hop`` (a ``hop_expr`` closure).  ``hdiag`` (float64 device
    tensor) feeds the preconditioner x / (hdiag - e + shift); ``mask`` (float64 0/1) restricts the iteration to the
    symmetry-allowed entries.  Returns (e, x, ncycle)."""
    es, xs, ncyc = _solve(hop, [x0], hdiag, 1, mask, tol, max_cycle, max_space, lindep, shift)
    return es[0], xs[0], ncyc


def davidson_multi(hop, guesses, hdiag, nroots, mask=None, tol=1e-12, max_cycle=100, max_space=None, lindep=1e-14,
                   shift=1e-4):
    """``nroots`` lowest eigenpairs (block Davidson; davidson.py:73-441 with nroots > 1 as called at gs.py:533-538).
    ``guesses``: list of device tensors (at least one); missing / dependent guesses are not replaced here - the
    caller supplies random ones like the reference (gs.py:273-276).  Default ``max_space`` = 12 + 3 (nroots - 1).
    Returns (list of e, list of x, ncycle)."""
    return _solve(hop, list(guesses), hdiag, nroots, mask, tol, max_cycle, max_space, lindep, shift)
