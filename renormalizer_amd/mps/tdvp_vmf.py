"""TDVP with the whole state integrated as one ODE: variable mean field (VMF), with the density-matrix or the
matrix-unfolding (MU) regularisation (counterpart of ``Mps._evolve_tdvp_mu_vmf``, renormalizer/mps/mps.py:887-1094;
Z. Phys. D 42, 113 (1997); arXiv:1907.12044).

State: left-canonical sites L L ... L C.  For every site  i dA_i/dt = S_L,i^-1 (1 - P_i) H_eff,i A_i S_R,i^-1  with
the projector P_i = S_L,i A_i S_L,i+1^-1 A_i^+ (plain A A^+ when ``force_ovlp`` is off), the left overlaps S_L, and
the inverse of the right density matrix S_R: eigenvalues regularised as w + eps exp(-w / eps) (``tdvp_vmf``) or,
matrix unfolding, the singular values of the right block as s + sqrt(eps) exp(-s / sqrt(eps)) (``tdvp_mu_vmf``).
The time stepping of VMF is SciPy's adaptive RK45 over the concatenated symmetry-allowed entries of all sites, exactly
as in the reference (step control on the host); every derivative evaluation - environments, effective-Hamiltonian
products, projector and overlap products, the block SVDs of the MU scheme - runs on the device.  The per-site
integrations of CMF run Dormand-Prince on device-resident vectors (lib/rk45.py, scipy's step-size rules)."""
import logging

import numpy as np
import scipy.linalg
from scipy.integrate import solve_ivp

from ..lib.rk45 import solve_rk45

from ..engine import get_engine
from ..utils import EvolveMethod
from . import svd_qn
from .hop_expr import hop_expr
from .lib import Environ, contract_one_site
from .svd_qn import get_qn_mask

logger = logging.getLogger("renormalizer_amd")


def mu_regularize(s, epsilon=1e-10):
    """mps/mps.py:1923-1929"""
    epsilon = np.sqrt(epsilon)
    return s + epsilon * np.exp(-s / epsilon)


def _ident_w(eng, site, cache):
    pdims = tuple(site.shape[1:-1])
    d = pdims[0]
    if d not in cache:
        cache[d] = eng.asdevice(np.eye(d).reshape(1, d, d, 1))
    return cache[d]


def transfer_matrix(eng, site, domain, val, cache):
    """overlap matrix (bra index, ket index) pushed through one site (``transferMat``, mps/mps.py:1885-1920)"""
    env = eng.asdevice(np.ascontiguousarray(val).reshape(val.shape[0], 1, val.shape[1]))
    out = contract_one_site(env, site, _ident_w(eng, site, cache), domain)
    return out.to_host().reshape(out.shape[0], out.shape[2])


class SiteDerivative:
    """``integrand_func_factory`` (mps/mps.py:1848-1883) for the left-canonical layout: y -> S_L0^-1 (1 - P) H y S^-1 /
    coef on device tensors; ``pre`` (optional) is multiplied onto the right bond of y before H (the factor the MU
    scheme moves from the right environment into the site)."""

    def __init__(self, eng, shape, hop, islast, s_inv, coef, ovlp_inv1=None, ovlp_inv0=None, ovlp0=None, pre=None):
        self.eng, self.shape, self.hop, self.islast, self.coef = eng, tuple(shape), hop, islast, coef
        up = (lambda m: m if (m is None or hasattr(m, "ptr")) else eng.asdevice(np.ascontiguousarray(m)))
        self.s_inv, self.ovlp_inv1, self.ovlp_inv0, self.ovlp0, self.pre = map(up, (s_inv, ovlp_inv1, ovlp_inv0, ovlp0, pre))

    def __call__(self, y0):
        """dy/dt on the host (the ODE drivers keep their vectors there)"""
        return self.device(y0).to_host().reshape(self.shape) / self.coef

    def device(self, y0):
        """S_L0^-1 (1 - P) H y S^-1 as a device tensor, not yet divided by ``coef``"""
        eng = self.eng
        dl, dr = self.shape[0], self.shape[-1]
        rows = int(np.prod(self.shape[:-1]))
        x = y0 if self.pre is None else eng.matmul(y0.reshape(rows, dr), self.pre)
        k = x.shape[-1] if self.pre is not None else dr
        hc = self.hop(x.reshape(self.shape[:-1] + (k,))).reshape(rows, k)
        if not self.islast:
            ymat = y0.reshape(rows, dr)
            m = eng.matmul(ymat, hc, conj_a=True, trans_a=True)                    # A^+ (H A): (dr, k)
            b = ymat
            if self.ovlp_inv1 is not None:
                b = eng.matmul(self.ovlp0, y0.reshape(dl, rows // dl * dr)).reshape(rows, dr)
                b = eng.matmul(b, self.ovlp_inv1)
            bm = eng.matmul(b, m)
            if bm.is_complex and not hc.is_complex:
                hc = hc.to_complex()
            eng._check(eng.lib.mpse_axpy(eng.ctx, hc.code, hc.ptr, bm.ptr, hc.size, -1.0, 0.0))
        if self.ovlp_inv0 is not None:
            hc = eng.matmul(self.ovlp_inv0, hc.reshape(dl, rows // dl * k)).reshape(rows, k)
        return eng.matmul(hc, self.s_inv).reshape(self.shape)


def evolve_tdvp_mu_vmf(self, mpo, evolve_dt):
    from .mpo import Mpo
    eng = get_engine()
    config = self.evolve_config
    if callable(mpo) and not isinstance(mpo, Mpo):
        mpo_t = mpo
    elif isinstance(mpo, Mpo):
        mpo_t = (lambda t, *args, **kwargs: mpo)
    else:
        raise TypeError(f"unsupported mpo type: {mpo}")
    imag_time = bool(np.iscomplex(evolve_dt))
    if imag_time:
        evolve_dt = -np.imag(evolve_dt)       # solve_ivp needs a real time axis; the sign goes into coef
        coef = -1
    else:
        coef = 1j
    if not (config.force_ovlp and not self.to_right):
        self.ensure_left_canonical()
    mps = self.copy() if imag_time else self.to_complex()
    n = mps.site_num
    dtype = np.complex128 if mps.is_complex else np.float64
    masks, position = [], [0]
    for i in range(n):
        mps.move_qnidx(i)
        _, _, qnmat = mps._get_big_qn([i])
        masks.append(get_qn_mask(qnmat, mps.qntot))
        position.append(position[-1] + int(masks[-1].sum()))
    sw_min_list = []
    ident_cache = {}

    def put_sites(y):
        for i in range(n):
            full = np.zeros(masks[i].shape, dtype=y.dtype)
            full[masks[i]] = y[position[i]:position[i + 1]]
            mps[i] = eng.asdevice(full)

    def func_vmf(t, y):
        sw_min_list.clear()
        put_sites(y)
        ham = mpo_t(t, mps=mps)
        mu = config.method == EvolveMethod.tdvp_mu_vmf
        if mu:
            environ_mps = mps.copy()
        else:
            assert config.method == EvolveMethod.tdvp_vmf
            environ_mps = mps
            s_r = np.ones((1, 1), dtype=dtype)
        environ = Environ(environ_mps, ham, "L")
        if config.force_ovlp:
            s_l = [np.ones((1, 1), dtype=dtype)]
            for i in range(n):
                s_l.append(transfer_matrix(eng, mps[i], "L", s_l[i], ident_cache))
            s_l_inv = []
            for m in s_l:
                w, u = scipy.linalg.eigh(m)
                s_l_inv.append((u / w) @ u.T.conj())
        else:
            s_l = s_l_inv = [None] * (n + 1)
        hop_y = np.empty_like(y)
        for i in mps.iter_idx_list(full=True):
            shape = list(mps[i].shape)
            ltensor = environ.read("L", i - 1)
            if i == n - 1:
                hop = hop_expr(ltensor, environ.sentinel, [ham.device(i, eng)], shape)
                f = SiteDerivative(eng, shape, hop, True, np.ones((1, 1), dtype=dtype), coef, s_l_inv[i + 1], s_l_inv[i],
                                   s_l[i])
                hop_y[position[i]:position[i + 1]] = f(mps[i])[masks[i]]
                continue
            pre = None
            if mu:
                # orthogonalise the right neighbour of the environment state; its singular values are what gets
                # regularised, the factor u s is carried into site i
                qnbigl, qnbigr, _ = environ_mps._get_big_qn([i + 1])
                u, s, qnlset, v, s, qnrset = svd_qn.svd_qn(environ_mps[i + 1], qnbigl, qnbigr, environ_mps.qntot,
                                                           system="R", full_matrices=False)
                environ_mps[i + 1] = v.T.reshape((len(s),) + tuple(environ_mps[i + 1].shape[1:]))
                rtensor = environ.GetLR("R", i + 1, environ_mps, ham, itensor=None, method="System")
                sw_min_list.append(s.min())
                regular_s = mu_regularize(s, epsilon=config.reg_epsilon)
                us = eng.asdevice(u.to_host() * s)
                site = environ_mps[i]
                rows = site.size // site.shape[-1]
                environ_mps[i] = eng.matmul(site.reshape(rows, site.shape[-1]), us).reshape(tuple(site.shape[:-1]) + (len(s),))
                environ_mps.qn[i + 1] = qnrset
                environ_mps.qnidx = i
                s_inv = (u.to_host().conj() / regular_s).T
                pre = us
                hop_shape = shape[:-1] + [len(s)]
            else:
                rtensor = environ.GetLR("R", i + 1, environ_mps, ham, itensor=None, method="System")
                s_r = transfer_matrix(eng, environ_mps[i + 1], "R", s_r, ident_cache)
                w, u = scipy.linalg.eigh(s_r)
                w = np.where(w > 0, w, 0)               # negative values are rounding noise
                sw_min_list.append(w.min())
                eps = config.reg_epsilon
                w = w + eps * np.exp(-w / eps)
                s_inv = ((u / w) @ u.T.conj()).T
                hop_shape = shape
            hop = hop_expr(ltensor, rtensor, [ham.device(i, eng)], hop_shape)
            f = SiteDerivative(eng, shape, hop, False, s_inv, coef, s_l_inv[i + 1], s_l_inv[i], s_l[i], pre=pre)
            hop_y[position[i]:position[i + 1]] = f(mps[i])[masks[i]]
        return hop_y

    init_y = np.concatenate([mps[i].to_host()[masks[i]] for i in range(n)]).astype(dtype)
    sol = solve_ivp(func_vmf, (0, evolve_dt), init_y, method="RK45", rtol=config.ivp_rtol, atol=config.ivp_atol)
    put_sites(sol.y[:, -1])
    logger.info(f"{config.method} VMF func called: {sol.nfev}. RKF steps: {len(sol.t)}")
    mps.evolve_config.stat = {"nfev": sol.nfev, "nsteps": len(sol.t)}
    sw_min = min(sw_min_list) if sw_min_list else np.inf
    if getattr(config, "vmf_auto_switch", True):
        # MU costs an SVD per site and evaluation; it is only needed while the right density matrix is singular
        if sw_min > np.sqrt(config.reg_epsilon * 10.0) and mps.evolve_config.method == EvolveMethod.tdvp_mu_vmf:
            mps.evolve_config.method = EvolveMethod.tdvp_vmf
        elif sw_min < config.reg_epsilon and mps.evolve_config.method == EvolveMethod.tdvp_vmf:
            mps.evolve_config.method = EvolveMethod.tdvp_mu_vmf
    return mps.canonicalise()


def evolve_tdvp_mu_cmf(self, mpo, evolve_dt):
    """TDVP with constant mean field and matrix-unfolding regularisation (``Mps._evolve_tdvp_mu_cmf``,
    mps/mps.py:1096-1265): the environments are frozen over the step - taken from the state at t (first order) or
    at t + dt/2 from a first-order half step (``tdvp_cmf_midpoint``, default) - and every site is integrated on
    its own with RK45, right to left, the coefficient site with the local propagator of ``ivp_solver``;
    ``tdvp_cmf_c_trapz`` propagates the coefficient site in two halves around the update of the other sites.

    One deviation from the reference: for the coefficient site with ``ivp_solver="krylov"`` it hands the operator
    already divided by i to its Hermitian Lanczos routine, which overflows once dt |H| is of order one (measured in
    the dev container); here the Hermitian operator and the factor -i dt go to the Lanczos exponential (identical
    results where the reference's works).  Kept as in the reference: in imaginary time the midpoint environment comes
    from a half step taken with the real number substituted for the step (mps.py:1113-1117, 1137).  With constant
    regularised inverses (~1e5 on bond directions padded by ``expand_bond_dimension``) imaginary-time CMF is unstable
    on freshly padded states - its first-order variant collapses within a step in the reference as well."""
    from ..lib.krylov import expm_krylov
    eng = get_engine()
    config = self.evolve_config
    if config.tdvp_cmf_c_trapz:
        assert config.tdvp_cmf_midpoint
    imag_time = bool(np.iscomplex(evolve_dt))
    if imag_time:
        evolve_dt = -np.imag(evolve_dt)
        coef = -1
    else:
        coef = 1j
    self.ensure_left_canonical()
    mps = self.copy() if imag_time else self.to_complex()
    n = mps.site_num
    dtype = np.complex128 if mps.is_complex else np.float64
    if config.tdvp_cmf_midpoint:
        orig = self.evolve_config
        first_order = orig.copy()
        first_order.tdvp_cmf_midpoint = first_order.tdvp_cmf_c_trapz = first_order.adaptive = False
        self.evolve_config = first_order
        try:
            environ_mps = self.evolve(mpo, evolve_dt / 2)
        finally:
            self.evolve_config = orig
    else:
        environ_mps = mps.copy()
    if environ_mps.is_complex and not mps.is_complex:
        mps = mps.to_complex()               # imaginary time with the (real-time) midpoint environment
        dtype = np.complex128
    loop = 1
    if config.tdvp_cmf_c_trapz:
        loop = 2
        mps[n - 1] = environ_mps[n - 1]
    ident_cache = {}
    rk_steps = []
    while loop > 0:
        environ = Environ(environ_mps, mpo, "L")
        if config.force_ovlp:
            s_l = [np.ones((1, 1), dtype=dtype)]
            for i in range(n):
                s_l.append(transfer_matrix(eng, environ_mps[i], "L", s_l[i], ident_cache))
            s_l_inv = []
            for m in s_l:
                w, u = scipy.linalg.eigh(m)
                s_l_inv.append((u / w) @ u.T.conj())
        else:
            s_l = s_l_inv = [None] * (n + 1)
        for i in mps.iter_idx_list(full=True):
            shape = list(mps[i].shape)
            ltensor = environ.read("L", i - 1)
            if i == n - 1:
                if loop == 1:
                    hop = hop_expr(ltensor, environ.sentinel, [mpo.device(i, eng)], shape)
                    f = SiteDerivative(eng, shape, hop, True, np.ones((1, 1), dtype=dtype), coef, s_l_inv[i + 1],
                                       s_l_inv[i], s_l[i])
                    if config.ivp_solver == "krylov":
                        ms, nvec = expm_krylov(lambda v: f.device(v.reshape(shape)), evolve_dt / coef, mps[i])
                        mps[i] = ms.reshape(shape)
                    elif config.ivp_solver == "RK45":
                        y0 = mps[i].to_complex() if dtype == np.complex128 else mps[i]
                        mps[i], _, _ = solve_rk45(lambda t, y, f=f: f.device(y).scale_(1.0 / coef), evolve_dt, y0,
                                                  rtol=config.ivp_rtol, atol=config.ivp_atol)
                    else:
                        y0 = mps[i].to_host().astype(dtype)
                        sol = solve_ivp(lambda t, y: f(eng.asdevice(y.reshape(shape))).ravel(), (0, evolve_dt), y0.ravel(),
                                        method=config.ivp_solver, rtol=config.ivp_rtol, atol=config.ivp_atol)
                        mps[i] = sol.y[:, -1].reshape(shape)
                if loop == 1 and config.tdvp_cmf_c_trapz:
                    break
                continue
            qnbigl, qnbigr, _ = environ_mps._get_big_qn([i + 1])
            u, s, qnlset, v, s, qnrset = svd_qn.svd_qn(environ_mps[i + 1], qnbigl, qnbigr, environ_mps.qntot, system="R",
                                                       full_matrices=False)
            environ_mps[i + 1] = v.T.reshape((len(s),) + tuple(environ_mps[i + 1].shape[1:]))
            rtensor = environ.GetLR("R", i + 1, environ_mps, mpo, itensor=None, method="System")
            regular_s = mu_regularize(s, epsilon=config.reg_epsilon)
            us = eng.asdevice(u.to_host() * s)
            site = environ_mps[i]
            rows = site.size // site.shape[-1]
            environ_mps[i] = eng.matmul(site.reshape(rows, site.shape[-1]), us).reshape(tuple(site.shape[:-1]) + (len(s),))
            environ_mps.qn[i + 1] = qnrset
            environ_mps.qnidx = i
            s_inv = (u.to_host().conj() / regular_s).T
            hop = hop_expr(ltensor, rtensor, [mpo.device(i, eng)], shape[:-1] + [len(s)])
            f = SiteDerivative(eng, shape, hop, False, s_inv, coef, s_l_inv[i + 1], s_l_inv[i], s_l[i], pre=us)
            # per-site Dormand-Prince on the device at scipy's default tolerances (the reference calls solve_ivp
            # without rtol / atol here, mps.py:1241-1247); rk_steps counts the time points like len(sol.t)
            y0 = mps[i].to_complex() if dtype == np.complex128 else mps[i]
            mps[i], _, nst = solve_rk45(lambda t, y, f=f: f.device(y).scale_(1.0 / coef), evolve_dt, y0)
            rk_steps.append(nst + 1)
        if loop == 2:
            environ_mps = mps
            evolve_dt /= 2.0
        loop -= 1
    mps.evolve_config.stat = {"rk_steps": rk_steps}
    return mps
