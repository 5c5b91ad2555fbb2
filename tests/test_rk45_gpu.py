"""Device Dormand-Prince (renormalizer_amd/lib/rk45.py) against scipy.integrate.solve_ivp(method="RK45"), the integrator
the reference calls for ivp_solver="RK45" local propagators and the per-site TDVP-CMF problems (mps/mps.py:1241-1247,
1299-1315): same tolerances -> the same sequence of steps, the same number of derivative evaluations, the same result
to rounding."""
import numpy as np
import pytest
from scipy.integrate import solve_ivp

pytestmark = pytest.mark.gpu


def _problem(n, cplx, seed):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, n))
    h = (a + a.T) / np.sqrt(n)
    y0 = rng.normal(size=n) + (1j * rng.normal(size=n) if cplx else 0)
    return h, y0 / np.linalg.norm(y0)


@pytest.mark.parametrize("cplx,rtol,atol,span", [(True, 1e-3, 1e-6, 0.7), (True, 1e-8, 1e-10, 2.0), (False, 1e-5, 1e-8, 1.3),
                                                 (True, 1e-5, 1e-8, 40.0)])
def test_rk45_follows_scipy(cplx, rtol, atol, span):
    from renormalizer_amd.engine import get_engine
    from renormalizer_amd.lib.rk45 import solve_rk45
    eng = get_engine()
    h, y0 = _problem(96, cplx, 5)
    phase = -1j if cplx else -1.0
    sol = solve_ivp(lambda t, v: phase * (h @ v), (0, span), y0, method="RK45", rtol=rtol, atol=atol)
    hd = eng.asdevice(h.astype(complex) if cplx else h)

    def rhs(t, v):
        return eng.matmul(hd, v.reshape(96, 1)).reshape(96).scale_(phase)

    y, nfev, nsteps = solve_rk45(rhs, span, eng.asdevice(y0), rtol=rtol, atol=atol)
    assert nfev == sol.nfev and nsteps + 1 == len(sol.t)
    assert np.abs(y.to_host() - sol.y[:, -1]).max() < 1e-12 * max(1.0, span)


def test_rk45_zero_rhs_and_rejections():
    from renormalizer_amd.engine import get_engine
    from renormalizer_amd.lib.rk45 import solve_rk45
    eng = get_engine()
    y0 = np.arange(1.0, 9.0)
    y, nfev, _ = solve_rk45(lambda t, v: eng.zeros(v.shape, v.dtype), 1.0, eng.asdevice(y0))
    sol = solve_ivp(lambda t, v: 0 * v, (0, 1.0), y0, method="RK45")
    assert nfev == sol.nfev and np.array_equal(y.to_host(), y0)
    # a stiff-ish diagonal problem forces step rejections; the count of evaluations pins the controller
    lam = -np.logspace(0, 3.5, 64)
    ld = eng.asdevice(np.diag(lam))
    y0 = np.ones(64)
    sol = solve_ivp(lambda t, v: lam * v, (0, 1.0), y0, method="RK45", rtol=1e-6, atol=1e-9)
    y, nfev, nsteps = solve_rk45(lambda t, v: eng.matmul(ld, v.reshape(64, 1)).reshape(64), 1.0, eng.asdevice(y0), rtol=1e-6,
                                 atol=1e-9)
    assert nfev == sol.nfev and nsteps + 1 == len(sol.t)
    assert np.abs(y.to_host() - sol.y[:, -1]).max() < 1e-12
