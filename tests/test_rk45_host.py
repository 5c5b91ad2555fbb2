"""Step control of the device Dormand-Prince driver (renormalizer_amd/lib/rk45.py) on a NumPy stand-in for the engine:
the controller is host logic - initial step, error norm, acceptance, growth limits - and must reproduce
scipy.integrate.solve_ivp(method="RK45") decision by decision (same number of derivative evaluations and accepted
steps).  The GPU test (test_rk45_gpu.py) runs the same comparison through the C ABI."""
import ctypes as C

import numpy as np
from scipy.integrate import solve_ivp

import renormalizer_amd.lib.rk45 as rk45


class _T:
    def __init__(self, eng, a):
        self.eng, self.a = eng, np.array(a)
        self.ptr = id(self)
        eng.reg[self.ptr] = self

    code = 0
    shape = property(lambda s: s.a.shape)
    dtype = property(lambda s: s.a.dtype)
    size = property(lambda s: s.a.size)

    def copy(self):
        return _T(self.eng, self.a.copy())


class _Lib:
    def __init__(self, eng):
        self.eng = eng

    def mpse_axpy(self, ctx, code, y, x, n, ar, ai):
        self.eng.reg[y].a = self.eng.reg[y].a + (ar + 1j * ai if ai else ar) * self.eng.reg[x].a
        return 0

    def mpse_scaled_rms(self, ctx, code, x, y1, y2, n, rtol, atol, out):
        r = self.eng.reg
        scale = atol + rtol * np.maximum(np.abs(r[y1].a), np.abs(r[y2].a))
        out[0] = float(np.sqrt(np.mean(np.abs(r[x].a / scale) ** 2)))
        return 0


class _Eng:
    ctx = None

    def __init__(self):
        self.reg = {}
        self.lib = _Lib(self)

    def _check(self, st):
        assert st == 0

    def zeros(self, shape, dtype):
        return _T(self, np.zeros(shape, dtype))


def _run(monkeypatch, fun, span, y0, **tol):
    eng = _Eng()
    monkeypatch.setattr(rk45, "get_engine", lambda: eng)
    y, nfev, nsteps = rk45.solve_rk45(lambda t, v: _T(eng, fun(t, v.a)), span, _T(eng, y0), **tol)
    sol = solve_ivp(fun, (0, span), y0, method="RK45", **tol)
    assert nfev == sol.nfev and nsteps + 1 == len(sol.t)
    assert np.abs(y.a - sol.y[:, -1]).max() < 1e-13 * max(1.0, span)


def test_rk45_controller_follows_scipy(monkeypatch):
    rng = np.random.default_rng(0)
    a = rng.normal(size=(12, 12))
    h = (a + a.T) / 4
    y0 = rng.normal(size=12) + 1j * rng.normal(size=12)
    _run(monkeypatch, lambda t, v: -1j * (h @ v), 3.0, y0)                                  # scipy's default tolerances
    _run(monkeypatch, lambda t, v: -1j * (h @ v), 25.0, y0, rtol=1e-9, atol=1e-12)           # many steps
    lam = -np.logspace(0, 3.5, 16)
    _run(monkeypatch, lambda t, v: lam * v, 1.0, np.ones(16), rtol=1e-6, atol=1e-9)          # rejections
    _run(monkeypatch, lambda t, v: 0 * v, 1.0, np.arange(1.0, 5.0))                          # zero derivative
    _run(monkeypatch, lambda t, v: np.cos(t) * v, 1e-7, np.ones(3))                          # span below the first step


def test_rk45_backward_integration(monkeypatch):
    """A negative bound integrates backwards like solve_ivp((0, t_bound)) - TDVP-CMF with a negative real evolve_dt
    (the reference supports it through scipy; a forward-only loop returned y0 unchanged)."""
    rng = np.random.default_rng(1)
    a = rng.normal(size=(10, 10))
    h = (a + a.T) / 4
    y0 = rng.normal(size=10) + 1j * rng.normal(size=10)
    _run(monkeypatch, lambda t, v: -1j * (h @ v), -3.0, y0)
    _run(monkeypatch, lambda t, v: -1j * (h @ v) * (1 + 0.1 * t), -12.0, y0, rtol=1e-8, atol=1e-11)
    _run(monkeypatch, lambda t, v: np.cos(t) * v, -1e-7, np.ones(3))
