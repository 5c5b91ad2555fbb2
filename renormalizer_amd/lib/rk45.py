"""Dormand-Prince 5(4) integrator on device-resident vectors.

Counterpart of ``scipy.integrate.solve_ivp(..., method="RK45")`` as the reference uses it for single-site problems
(``ivp_solver="RK45"`` local propagators, mps/mps.py:1299-1315, 1342-1360, and the per-site integrations of TDVP-CMF,
:1096-1265).  The state, the stages and the error estimate live in HBM (axpy kernels, one reduction per step for the
error norm, ``mpse_scaled_rms``); only the step-size control runs on the host, with exactly SciPy's rules (initial step
``select_initial_step``, safety 0.9, factors 0.2 ... 10, no growth after a rejection, rms norm), so that the sequence
of steps is the reference's."""
import ctypes as C

import numpy as np

from ..engine import get_engine

_C = [0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0]
_A = [[],
      [1 / 5],
      [3 / 40, 9 / 40],
      [44 / 45, -56 / 15, 32 / 9],
      [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
      [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]]
_B = [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]
_E = [-71 / 57600, 0.0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40]
_SAFETY, _MIN_FACTOR, _MAX_FACTOR, _EXPONENT = 0.9, 0.2, 10.0, -0.2


def _axpy(eng, y, x, a):
    if a != 0.0:
        eng._check(eng.lib.mpse_axpy(eng.ctx, y.code, y.ptr, x.ptr, y.size, float(a), 0.0))


def _combine(eng, base, ks, coefs, h):
    out = base.copy()
    for k, c in zip(ks, coefs):
        _axpy(eng, out, k, c * h)
    return out


def _rms(eng, x, y1, y2, rtol, atol):
    out = (C.c_double * 2)()
    eng._check(eng.lib.mpse_scaled_rms(eng.ctx, x.code, x.ptr, y1.ptr, y2.ptr, x.size, rtol, atol, out))
    return float(out[0])


def solve_rk45(fun, t_bound, y0, rtol=1e-3, atol=1e-6):
    """Integrate dy/dt = fun(t, y) from 0 to ``t_bound`` (either sign: a negative bound integrates backwards, as
    ``solve_ivp((0, t_bound))`` does).  ``fun`` maps a device tensor to a device tensor of the same shape and dtype.
    Returns (y(t_bound), number of evaluations of fun, number of accepted steps)."""
    eng = get_engine()
    t_bound = float(t_bound)
    t, y = 0.0, y0
    if y.size == 0 or t_bound == 0:
        return y, 0, 0
    direction = 1.0 if t_bound > 0 else -1.0
    span = abs(t_bound)
    f = fun(t, y)
    nfev = 1
    # scipy/integrate/_ivp/common.py select_initial_step (order of the error estimator: 4)
    d0, d1 = _rms(eng, y, y, y, rtol, atol), _rms(eng, f, y, y, rtol, atol)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = min(h0, span)
    f1 = fun(t + h0 * direction, _combine(eng, y, [f], [1.0], h0 * direction))
    nfev += 1
    diff = f1.copy()
    _axpy(eng, diff, f, -1.0)
    d2 = _rms(eng, diff, y, y, rtol, atol) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** 0.2
    h_abs = min(100 * h0, h1, span)
    nsteps = 0
    while direction * (t - t_bound) < 0:
        min_step = 10 * abs(np.nextafter(t, direction * np.inf) - t)
        h_abs = max(h_abs, min_step)
        rejected = False
        while True:
            if h_abs < min_step:
                raise RuntimeError("solve_rk45: required step size is less than spacing between numbers")
            t_new = t + h_abs * direction
            if direction * (t_new - t_bound) > 0:
                t_new = t_bound
            h = t_new - t
            h_abs = abs(h)
            ks = [f]
            for s in range(1, 6):
                ks.append(fun(t + _C[s] * h, _combine(eng, y, ks, _A[s], h)))
            y_new = _combine(eng, y, ks, _B, h)
            f_new = fun(t + h, y_new)
            ks.append(f_new)
            nfev += 6
            err = eng.zeros(y.shape, y.dtype)
            for k, e in zip(ks, _E):
                _axpy(eng, err, k, e * h)
            error_norm = _rms(eng, err, y, y_new, rtol, atol)
            if error_norm < 1:
                factor = _MAX_FACTOR if error_norm == 0 else min(_MAX_FACTOR, _SAFETY * error_norm ** _EXPONENT)
                if rejected:
                    factor = min(1.0, factor)
                h_abs *= factor
                break
            h_abs *= max(_MIN_FACTOR, _SAFETY * error_norm ** _EXPONENT)
            rejected = True
        t, y, f = t_new, y_new, f_new
        nsteps += 1
    return y, nfev, nsteps
