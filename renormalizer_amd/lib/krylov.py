"""Lanczos exponential (counterpart of renormalizer/lib/krylov/krylov.py:27-82).

When ``Afunc`` is an effective-Hamiltonian closure from ``hop_expr`` the whole solve runs
inside the engine (mpse_expm_lanczos): Krylov vectors stay in HBM, only alpha/beta scalars
and the convergence flag cross to the host.  A generic callable on device tensors is also
accepted and driven from Python with the same recurrence and stopping rule."""
import ctypes as C

import numpy as np
import scipy.linalg

from ..engine import get_engine
from ..mps.hop_expr import Hop


def expm_krylov(Afunc, dt, vstart, block_size=50, rtol=1e-5, atol=1e-8, out=None):
    """``out``: (engine Hop only) an existing device tensor of the result's shape and dtype to receive it - calls
    recorded with ``Engine.recording`` refer to it before the solve has run."""
    eng = get_engine()
    dt = complex(dt)
    v = eng.asdevice(vstart)
    if isinstance(Afunc, Hop):
        if (dt.imag != 0 or Afunc.operator_is_complex) and not v.is_complex:
            v = v.to_complex()
        if out is None:
            out = eng.empty(v.shape, v.dtype)
        elif out.size != v.size or out.dtype != v.dtype:
            raise ValueError("expm_krylov: out does not match the start vector")
        nv = C.c_int()
        if Afunc.cmask is not None:
            eng._check(eng.lib.mpse_expm_centre_mask(eng.ctx, Afunc.cmask.ptr, Afunc.cmask.nbytes))
        eng._check(eng.lib.mpse_expm_lanczos(eng.ctx, v.code, C.byref(Afunc.heff), dt.real, dt.imag, v.ptr, out.ptr,
                                             rtol, atol, 0, C.byref(nv)))
        return out, nv.value
    return _expm_krylov_generic(eng, Afunc, dt, v, block_size, rtol, atol)


def _combine(eng, V, coef, dtype):
    out = eng.zeros(V[0].shape, dtype)
    for c, vec in zip(coef, V):
        c = complex(c)
        eng._check(eng.lib.mpse_axpy(eng.ctx, out.code, out.ptr, vec.ptr, out.size, c.real, c.imag))
    return out


def _expm_krylov_generic(eng, Afunc, dt, v, block_size, rtol, atol):
    if dt.imag != 0 and not v.is_complex:
        v = v.to_complex()
    n = v.size
    nrmv = v.norm()
    assert nrmv > 0
    V = [v.copy().scale_(1.0 / nrmv)]
    alpha, beta = [], []
    res = None
    dtx = dt if dt.imag != 0 else dt.real

    def small(m):
        w, u = scipy.linalg.eigh_tridiagonal(np.array(alpha[:m]), np.array(beta[:m - 1]))
        return u @ (nrmv * np.exp(dtx * w) * u[0])

    for j in range(n):
        w = Afunc(V[j]).reshape(V[j].shape)
        alpha.append(complex(w.vdot(V[j])).real)
        if j == n - 1:
            return _combine(eng, V, small(j + 1), v.dtype), j + 1
        eng._check(eng.lib.mpse_axpy(eng.ctx, w.code, w.ptr, V[j].ptr, n, -alpha[j], 0.0))
        if j > 0:
            eng._check(eng.lib.mpse_axpy(eng.ctx, w.code, w.ptr, V[j - 1].ptr, n, -beta[j - 1], 0.0))
        beta.append(w.norm())
        if beta[j] < 100 * n * np.finfo(float).eps:
            return _combine(eng, V, small(j + 1), v.dtype), j + 1
        if 3 < j and j % 2 == 0:
            new_res = _combine(eng, V, small(j + 1), v.dtype)
            if res is not None and np.allclose(res.to_host(), new_res.to_host(), rtol=rtol, atol=atol):
                return new_res, j + 1
            res = new_res
        V.append(w.scale_(1.0 / beta[j]))
    raise AssertionError("unreachable")
