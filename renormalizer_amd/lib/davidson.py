"""Davidson eigensolver on device-resident vectors.

Counterpart of ``davidson`` in renormalizer/lib/davidson/davidson.py:73-441 as called from
mps/gs.py:533-538 (single lowest root): diagonal preconditioner x / (hdiag - e + 1e-4), subspace restart
when ``max_space`` vectors are held, convergence on |de| < tol and |residual| < sqrt(tol), new directions
dropped when their norm after orthogonalisation falls under ``lindep``.  Subspace matrices (<= 12 x 12) are
diagonalised on the host; every vector operation and the matvec run on the GPU."""
import numpy as np

from ..engine import get_engine


def _lincomb(eng, vecs, coef, like):
    out = eng.zeros(like.shape, like.dtype)
    for c, v in zip(coef, vecs):
        c = complex(c)
        eng._check(eng.lib.mpse_axpy(eng.ctx, out.code, out.ptr, v.ptr, out.size, c.real, c.imag))
    return out


def davidson(aop, x0, hdiag, mask=None, tol=1e-12, max_cycle=100, max_space=12, lindep=1e-14, shift=1e-4):
    """Lowest eigenpair of the Hermitian operator ``aop`` (callable on a device tensor).
    ``hdiag`` (float64 device tensor) feeds the preconditioner; ``mask`` (float64 0/1) restricts the iteration
    to the symmetry-allowed entries.  Returns (e, x, ncycle)."""
    eng = get_engine()
    n = x0.size
    toloose = np.sqrt(tol)
    x0 = x0.copy()
    if mask is not None:
        eng._check(eng.lib.mpse_mul_real(eng.ctx, x0.code, x0.ptr, mask.ptr, n))
        nfree = None
    nrm = x0.norm()
    if not nrm > 0:
        raise ValueError("davidson: zero initial guess")
    x0.scale_(1.0 / nrm)
    V, W = [x0], []
    e_last = None
    e = x = hx = None
    ncyc = 0
    for ncyc in range(1, max_cycle + 1):
        while len(W) < len(V):
            w = aop(V[len(W)])
            if mask is not None:
                eng._check(eng.lib.mpse_mul_real(eng.ctx, w.code, w.ptr, mask.ptr, n))
            W.append(w)
        m = len(V)
        hsub = np.zeros((m, m), dtype=complex)
        for i in range(m):
            for j in range(i, m):
                hsub[i, j] = V[i].vdot(W[j])
                hsub[j, i] = np.conj(hsub[i, j])
        if not x0.is_complex:
            hsub = hsub.real
        ew, ev = np.linalg.eigh(hsub)
        e, c = float(ew[0]), ev[:, 0]
        x = _lincomb(eng, V, c, x0)
        hx = _lincomb(eng, W, c, x0)
        r = hx.copy()
        eng._check(eng.lib.mpse_axpy(eng.ctx, r.code, r.ptr, x.ptr, n, -e, 0.0))
        rnorm = r.norm()
        de = np.inf if e_last is None else e - e_last
        e_last = e
        if abs(de) < tol and rnorm < toloose:
            break
        if rnorm < 1e-14:                      # exact eigenvector (tiny spaces)
            break
        t = eng.empty(x.shape, x.dtype)
        eng._check(eng.lib.mpse_davidson_precond(eng.ctx, t.code, t.ptr, r.ptr, hdiag.ptr,
                                                 None if mask is None else mask.ptr, n, e, shift))
        if m >= max_space or m >= n:
            V, W = [x], [hx]                    # restart from the current Ritz vector
            nx = x.norm()
            x.scale_(1.0 / nx)
            hx.scale_(1.0 / nx)
        for _ in range(2):                      # Gram-Schmidt twice
            for v in V:
                ov = complex(v.vdot(t))
                eng._check(eng.lib.mpse_axpy(eng.ctx, t.code, t.ptr, v.ptr, n, -ov.real, -ov.imag))
        tn = t.norm()
        if tn ** 2 < lindep:
            break
        V.append(t.scale_(1.0 / tn))
    return e, x, ncyc


def davidson_multi(aop, guesses, hdiag, nroots, mask=None, tol=1e-12, max_cycle=100, max_space=None, lindep=1e-14,
                   shift=1e-4):
    """``nroots`` lowest eigenpairs (block Davidson; davidson.py:73-441 with nroots > 1 as called at gs.py:533-538).
    ``guesses``: list of device tensors (at least one); missing / dependent guesses are not replaced here - the
    caller supplies random ones like the reference (gs.py:273-276).  Default ``max_space`` = 12 + 3 (nroots - 1).
    Returns (list of e, list of x, ncycle); roots are converged when |de| < tol and |r| < sqrt(tol)."""
    eng = get_engine()
    if max_space is None:
        max_space = 12 + (nroots - 1) * 3
    toloose = np.sqrt(tol)
    like = guesses[0]
    n = like.size

    def masked(v):
        if mask is not None:
            eng._check(eng.lib.mpse_mul_real(eng.ctx, v.code, v.ptr, mask.ptr, n))
        return v

    def orth_append(V, t):
        """Gram-Schmidt twice against V; append when what is left is not negligible"""
        for _ in range(2):
            for v in V:
                ov = complex(v.vdot(t))
                eng._check(eng.lib.mpse_axpy(eng.ctx, t.code, t.ptr, v.ptr, n, -ov.real, -ov.imag))
        tn = t.norm()
        if tn ** 2 < lindep:
            return False
        V.append(t.scale_(1.0 / tn))
        return True

    V, W = [], []
    for g in guesses:
        g = masked(g.copy())
        nrm = g.norm()
        if nrm > 0:
            orth_append(V, g.scale_(1.0 / nrm))
    if not V:
        raise ValueError("davidson: zero initial guesses")
    e_last = None
    es, xs = None, None
    ncyc = 0
    for ncyc in range(1, max_cycle + 1):
        while len(W) < len(V):
            W.append(masked(aop(V[len(W)])))
        m = len(V)
        hsub = np.zeros((m, m), dtype=complex)
        for i in range(m):
            for j in range(i, m):
                hsub[i, j] = V[i].vdot(W[j])
                hsub[j, i] = np.conj(hsub[i, j])
        if not like.is_complex:
            hsub = hsub.real
        ew, ev = np.linalg.eigh(hsub)
        k = min(nroots, m)
        es = [float(x) for x in ew[:k]]
        xs = [_lincomb(eng, V, ev[:, r], like) for r in range(k)]
        hxs = [_lincomb(eng, W, ev[:, r], like) for r in range(k)]
        rs, conv = [], []
        for r in range(k):
            res = hxs[r].copy()
            eng._check(eng.lib.mpse_axpy(eng.ctx, res.code, res.ptr, xs[r].ptr, n, -es[r], 0.0))
            rn = res.norm()
            de = np.inf if (e_last is None or r >= len(e_last)) else es[r] - e_last[r]
            conv.append((abs(de) < tol and rn < toloose) or rn < 1e-14)
            rs.append(res)
        e_last = es
        if k == nroots and all(conv):
            break
        if m >= n:
            break
        todo = [r for r in range(k) if not conv[r]] or list(range(k))
        if m + len(todo) > max_space:
            # restart from the current Ritz vectors (orthonormal by construction up to rounding)
            V, W = [], []
            for r in range(k):
                x, hx = xs[r].copy(), hxs[r].copy()
                for _ in range(2):
                    for v, w in zip(list(V), list(W)):
                        ov = complex(v.vdot(x))
                        eng._check(eng.lib.mpse_axpy(eng.ctx, x.code, x.ptr, v.ptr, n, -ov.real, -ov.imag))
                        eng._check(eng.lib.mpse_axpy(eng.ctx, hx.code, hx.ptr, w.ptr, n, -ov.real, -ov.imag))
                nx = x.norm()
                V.append(x.scale_(1.0 / nx))
                W.append(hx.scale_(1.0 / nx))
        added = 0
        for r in todo:
            t = eng.empty(like.shape, like.dtype)
            eng._check(eng.lib.mpse_davidson_precond(eng.ctx, t.code, t.ptr, rs[r].ptr, hdiag.ptr,
                                                     None if mask is None else mask.ptr, n, es[r], shift))
            added += bool(orth_append(V, t))
        if added == 0:
            break
    return es, xs, ncyc
