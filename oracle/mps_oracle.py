"""NumPy/SciPy restatement of the reference's per-site sweep primitives.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference location (relative to /root/reference/renormalizer) it follows.  The
arithmetic is deliberately structured like the reference's CPU path (three
``tensordot`` calls per contraction, SciPy LAPACK per quantum-number block, the
same Lanczos stopping rule) because it doubles as the timed CPU baseline.

Index conventions (identical to the reference):
  environment  (bra bond, mpo bond, ket bond)
  mps site     (D_l, d, D_r)            [ancilla variant: (D_l, d, d_anc, D_r)]
  mpo site     (w_l, d_up, d_down, w_r)
Quantum numbers are int64 arrays whose last axis has length ``qn_size``.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.linalg


# ---------------------------------------------------------------------------
# integer quantum-number algebra  (bit-exact contract)
# ---------------------------------------------------------------------------

def add_outer(a, b):
    """Outer sum over all leading axes, element-wise on the trailing qn axis.
    mps/svd_qn.py:305-313."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape[-1] == b.shape[-1]
    a2 = a.reshape(a.shape[:-1] + (1,) * (b.ndim - 1) + a.shape[-1:])
    return a2 + b


def get_qn_mask(qnmat, qntot):
    """mps/svd_qn.py:316-317."""
    return np.all(np.asarray(qnmat) == np.asarray(qntot), axis=-1)


def get_big_qn(qnl, qnr, sigmaqn, to_right):
    """Super-block quantum numbers around the centre site(s).  mps/mp.py:308-352.

    ``sigmaqn`` is a list with one (d, qn_size) array for a 1-site centre or two
    for a 2-site centre."""
    qnl = np.asarray(qnl)
    qnr = np.asarray(qnr)
    sig = [np.asarray(s) for s in sigmaqn]
    if len(sig) == 1:
        if to_right:
            qnbigl, qnbigr = add_outer(qnl, sig[0]), qnr
        else:
            qnbigl, qnbigr = qnl, add_outer(sig[0], qnr)
    else:
        qnbigl, qnbigr = add_outer(qnl, sig[0]), add_outer(sig[1], qnr)
    return qnbigl, qnbigr, add_outer(qnbigl, qnbigr)


def qn_blocks(qnbigl, qnbigr, qntot):
    """Enumerate the symmetry blocks of the centre matrix: for every distinct
    left qn ``nl`` the rows with that qn and the columns with ``qntot - nl``
    (mps/svd_qn.py:177-182).  The reference walks a CPython ``set``; the block
    order is a gauge, here it is fixed to lexicographic order of ``nl``."""
    qntot = np.asarray(qntot)
    q = len(qntot)
    lq = np.asarray(qnbigl).reshape(-1, q)
    rq = np.asarray(qnbigr).reshape(-1, q)
    out = []
    for nl in sorted(set(map(tuple, lq.tolist()))):
        nr = qntot - np.array(nl)
        rset = np.nonzero(get_qn_mask(rq, nr))[0]
        if len(rset) == 0:
            continue
        lset = np.nonzero(get_qn_mask(lq, nl))[0]
        out.append((tuple(int(x) for x in nl), tuple(int(x) for x in nr), lset, rset))
    return out


# ---------------------------------------------------------------------------
# environment update and effective-Hamiltonian matvec
# ---------------------------------------------------------------------------

def contract_one_site(environ, ms, mo, domain, ms_conj=None):
    """One-site environment update, mps/lib.py:169-250.

    L: abc,adf->bcdf ; bcdf,bdeg->cfeg ; cfeg,ceh->fgh      (lib.py:200-205)
    R: fda,abc->fdbc ; fdbc,gdeb->fcge ; fcge,hec->fgh      (lib.py:233-237)
    ndim-4 (ancilla traced) variants lib.py:207-211 / 239-243."""
    ms = np.asarray(ms)
    mo = np.asarray(mo)
    environ = np.asarray(environ)
    if ms_conj is None:
        ms_conj = ms.conj()
    ms_conj = np.asarray(ms_conj)
    if domain == "L":
        if ms.ndim == 3:
            t = np.tensordot(environ, ms_conj, ([0], [0]))          # b c d f
            t = np.tensordot(t, mo, ([0, 2], [0, 1]))               # c f e g
            return np.tensordot(t, ms, ([0, 2], [0, 1]))            # f g h
        t = np.tensordot(environ, ms_conj, ([0], [0]))              # b c d l f
        t = np.tensordot(t, mo, ([0, 2], [0, 1]))                   # c l f e g
        return np.tensordot(t, ms, ([0, 3, 1], [0, 1, 2]))          # f g h
    if domain == "R":
        if ms.ndim == 3:
            t = np.tensordot(ms_conj, environ, ([2], [0]))          # f d b c
            t = np.tensordot(t, mo, ([1, 2], [1, 3]))               # f c g e
            return np.tensordot(t, ms, ([3, 1], [1, 2]))            # f g h
        t = np.tensordot(ms_conj, environ, ([3], [0]))              # f d l b c
        t = np.tensordot(t, mo, ([1, 3], [1, 3]))                   # f l c g e
        return np.tensordot(t, ms, ([4, 1, 2], [1, 2, 3]))          # f g h
    raise ValueError(domain)


def hop_apply(ltensor, rtensor, cmo, c):
    """Effective Hamiltonian applied to the centre tensor, order (L.C).W.R as
    chosen by opt_einsum's optimal path for D >> w,d.  mps/hop_expr.py:57-115.

    len(cmo)==0: abc,lbk,ck->al                 (hop_expr.py:63-67)
    len(cmo)==1: abc,bdef,lfk,cek->adl          (75-79)   [+ancilla cegk->adgl, 87-91]
    len(cmo)==2: abc,bdef,fghj,ljk,cehk->adgl   (99-103)  [+ancilla cemhnk->admgnl, 111-115]"""
    l = np.asarray(ltensor)
    r = np.asarray(rtensor)
    c = np.asarray(c)
    n = len(cmo)
    if n == 0:
        t = np.tensordot(l, c, ([2], [0]))                          # a b k
        return np.tensordot(t, r, ([1, 2], [1, 2]))                 # a l
    anc = (c.ndim == 2 * n + 2)
    if n == 1:
        w = np.asarray(cmo[0])
        t = np.tensordot(l, c, ([2], [0]))                          # a b e [g] k
        t = np.tensordot(t, w, ([1, 2], [0, 2]))                    # a [g] k d f
        if not anc:
            t = np.tensordot(t, r, ([3, 1], [1, 2]))                # a d l
            return t
        t = np.tensordot(t, r, ([4, 2], [1, 2]))                    # a g d l
        return t.transpose(0, 2, 1, 3)                              # a d g l
    if n == 2:
        w1 = np.asarray(cmo[0])
        w2 = np.asarray(cmo[1])
        t = np.tensordot(l, c, ([2], [0]))                          # a b e [m] h [n] k
        if not anc:
            t = np.tensordot(t, w1, ([1, 2], [0, 2]))               # a h k d f
            t = np.tensordot(t, w2, ([4, 1], [0, 2]))               # a k d g j
            return np.tensordot(t, r, ([4, 1], [1, 2]))             # a d g l
        t = np.tensordot(t, w1, ([1, 2], [0, 2]))                   # a m h n k d f
        t = np.tensordot(t, w2, ([6, 2], [0, 2]))                   # a m n k d g j
        t = np.tensordot(t, r, ([6, 3], [1, 2]))                    # a m n d g l
        return t.transpose(0, 3, 1, 4, 2, 5)                        # a d m g n l
    raise ValueError("at most two centre sites")


def hop_dense(ltensor, rtensor, cmo):
    """Dense effective Hamiltonian (row = out index, col = in index) for small
    centres, as used by the direct eigensolver mps/gs.py:383-407."""
    l = np.asarray(ltensor)
    r = np.asarray(rtensor)
    if len(cmo) == 1:
        h = np.einsum("abc,bdef,lfk->adlcek", l, cmo[0], r, optimize=True)
    elif len(cmo) == 2:
        h = np.einsum("abc,bdef,fghj,ljk->adglcehk", l, cmo[0], cmo[1], r, optimize=True)
    else:
        h = np.einsum("abc,lbk->alck", l, r, optimize=True)
    n = int(np.prod(h.shape[: h.ndim // 2]))
    return h.reshape(n, n)


# ---------------------------------------------------------------------------
# Lanczos exponential
# ---------------------------------------------------------------------------

def _expm_tridiag(alpha, beta, V, v_norm, dt):
    """lib/krylov/krylov.py:15-24; V has the Krylov vectors as ROWS here."""
    try:
        w, u = scipy.linalg.eigh_tridiagonal(alpha, beta)
    except np.linalg.LinAlgError:
        h = np.diag(alpha) + np.diag(beta, k=-1) + np.diag(beta, k=1)
        w, u = np.linalg.eigh(h)
    coef = u @ (v_norm * np.exp(dt * w) * u[0])
    return V.T @ coef, coef


def expm_krylov(Afunc, dt, vstart, block_size=50, rtol=1e-5, atol=1e-8, return_stat=False, margins=None):
    """Krylov approximation of expm(dt*A) v for Hermitian A.
    lib/krylov/krylov.py:27-82: no re-orthogonalisation, growth by blocks of
    ``block_size``, breakdown if beta < 100*n*eps, convergence test = successive
    approximations ``allclose`` on every even j > 3.

    ``margins`` (test instrumentation, not in the reference): a list that receives, per convergence test, the largest
    ``|res - new_res| / (atol + rtol |new_res|)`` - the test passes where it is <= 1, so values near 1 mark solves whose
    Krylov dimension hinges on rounding."""
    dt = complex(dt)
    if dt.imag == 0:
        dt = dt.real
    vstart = np.asarray(vstart)
    n = len(vstart)
    nrmv = float(np.linalg.norm(vstart))
    assert nrmv > 0
    alpha = np.zeros(block_size)
    beta = np.zeros(block_size - 1)
    V = np.empty((block_size, n), dtype=vstart.dtype)
    V[0] = vstart / nrmv
    res = None
    for j in range(n):
        w = Afunc(V[j])
        alpha[j] = np.vdot(w, V[j]).real
        if j == n - 1:
            return _expm_tridiag(alpha[: j + 1], beta[:j], V[: j + 1], nrmv, dt)[0], j + 1
        if len(V) == j + 1:
            V2 = np.empty((len(V) + block_size, n), dtype=V.dtype)
            V2[: len(V)] = V
            V = V2
            alpha = np.concatenate([alpha, np.zeros(block_size)])
            beta = np.concatenate([beta, np.zeros(block_size)])
        w = w - (alpha[j] * V[j] + (beta[j - 1] * V[j - 1] if j > 0 else 0))
        beta[j] = np.linalg.norm(w)
        if beta[j] < 100 * n * np.finfo(float).eps:
            return _expm_tridiag(alpha[: j + 1], beta[:j], V[: j + 1], nrmv, dt)[0], j + 1
        if 3 < j and j % 2 == 0:
            new_res = _expm_tridiag(alpha[: j + 1], beta[:j], V[: j + 1], nrmv, dt)[0]
            if margins is not None and res is not None:
                margins.append(float(np.max(np.abs(res - new_res) / (atol + rtol * np.abs(new_res)))))
            if res is not None and np.allclose(res, new_res, rtol=rtol, atol=atol):
                return new_res, j + 1
            res = new_res
        V[j + 1] = w / beta[j]
    raise AssertionError("unreachable")


# ---------------------------------------------------------------------------
# quantum-number blocked QR / RQ / SVD
# ---------------------------------------------------------------------------

def _scatter_rows(indices, block, nrow):
    """mps/svd_qn.py:89-96 (blockrecover)."""
    out = np.zeros((nrow, block.shape[1]), dtype=block.dtype)
    out[indices, :] = block
    return out


def _complete_basis(u, rng):
    """n extra orthonormal columns for a tall isometry, mps/svd_qn.py:52-63."""
    m, n = u.shape
    a = rng.random((m, n))
    a = a - u @ (u.conj().T @ a)
    q, _ = scipy.linalg.qr(a, mode="economic")
    return np.concatenate([u, q], axis=1)


def _block_svd(a, full_matrices, opt_full_matrices, rng):
    """mps/svd_qn.py:12-49 (gesdd with gesvd fallback; cheap 'full' completion for
    aspect ratios outside (1/3, 3))."""
    m, n = a.shape
    if not full_matrices:
        opt_full_matrices = False
    opt = opt_full_matrices and not (1 / 3 < m / n < 3)
    full = full_matrices and not opt
    try:
        U, S, Vt = scipy.linalg.svd(a, full_matrices=full, lapack_driver="gesdd")
    except scipy.linalg.LinAlgError:
        U, S, Vt = scipy.linalg.svd(a, full_matrices=full, lapack_driver="gesvd")
    if opt:
        if m < n:
            Vt = _complete_basis(Vt.T, rng).T
        else:
            U = _complete_basis(U, rng)
    return U, S, Vt


def svd_qn(coef_array, qnbigl, qnbigr, qntot, QR=False, system=None,
           full_matrices=True, opt_full_matrices=True, rng=None):
    """Block decomposition of the centre tensor by quantum number.
    mps/svd_qn.py:99-240.

    Returns ``(u, su, qnl, v, sv, qnr)`` for SVD and ``(u, qnl, v, qnr)`` for
    QR (system="L") / RQ (system="R"); ``coef == u @ diag(s) @ v.T`` resp.
    ``coef == u @ v.T`` restricted to the symmetry-allowed entries."""
    if rng is None:
        rng = np.random.default_rng(2019)
    qntot = np.asarray(qntot)
    q = len(qntot)
    nrow = int(np.prod(np.asarray(qnbigl).shape[:-1]))
    ncol = int(np.prod(np.asarray(qnbigr).shape[:-1]))
    mat = np.asarray(coef_array).reshape(nrow, ncol)
    u_nz, u_0, v_nz, v_0 = [], [], [], []
    s_nz, su_0, sv_0 = [], [], []
    qnl_nz, qnl_0, qnr_nz, qnr_0 = [], [], [], []
    for nl, nr, lset, rset in qn_blocks(qnbigl, qnbigr, qntot):
        block = mat[np.ix_(lset, rset)]
        dim = min(block.shape)
        if not QR:
            bu, bs, bvt = _block_svd(block, full_matrices, opt_full_matrices, rng)
            s_nz.append(bs)
        else:
            mode = "full" if full_matrices else "economic"
            if system == "R":
                bu, bvt = scipy.linalg.rq(block, mode=mode)
            elif system == "L":
                bu, bvt = scipy.linalg.qr(block, mode=mode)
            else:
                raise ValueError("system must be 'L' or 'R' for QR")
        bv = bvt.T
        u_nz.append(_scatter_rows(lset, bu[:, :dim], nrow))
        v_nz.append(_scatter_rows(rset, bv[:, :dim], ncol))
        qnl_nz += [list(nl)] * dim
        qnr_nz += [list(nr)] * dim
        if full_matrices:
            u_0.append(_scatter_rows(lset, bu[:, dim:], nrow))
            v_0.append(_scatter_rows(rset, bv[:, dim:], ncol))
            qnl_0 += [list(nl)] * (bu.shape[1] - dim)
            qnr_0 += [list(nr)] * (bv.shape[1] - dim)
            su_0.append(np.zeros(bu.shape[1] - dim))
            sv_0.append(np.zeros(bv.shape[1] - dim))
    if len(u_nz) + len(u_0) == 0:
        raise ValueError("Invalid quantum number")
    u = np.concatenate(u_nz + u_0, axis=1)
    v = np.concatenate(v_nz + v_0, axis=1)
    qnl = qnl_nz + qnl_0
    qnr = qnr_nz + qnr_0
    if QR:
        return u, qnl, v, qnr
    su = np.concatenate(s_nz + su_0)
    sv = np.concatenate(s_nz + sv_0)
    if not full_matrices:
        order = np.argsort(su)[::-1]
        u = u[:, order]
        v = v[:, order]
        su = sv = su[order]
        qnl = np.array(qnl)[order].tolist()
        qnr = np.array(qnr)[order].tolist()
    return u, su, qnl, v, sv, qnr


# ---------------------------------------------------------------------------
# truncation
# ---------------------------------------------------------------------------

def compute_m_trunc(sigma, criteria, threshold=1e-3, max_dim=None):
    """utils/configs.py:196-219.  criteria in {"threshold", "fixed", "both"}."""
    sigma = np.asarray(sigma)
    by_thr = int(np.sum(sigma / scipy.linalg.norm(sigma) > threshold)) if criteria != "fixed" else None
    by_dim = min(int(max_dim), len(sigma)) if criteria != "threshold" else None
    if criteria == "threshold":
        return by_thr
    if criteria == "fixed":
        return by_dim
    return min(by_thr, by_dim)


def select_basis_indices(sset, qnlist, Mmax, percent=0.0):
    """Index selection of mps/lib.py:253-300: an optional equal quota of
    ``percent`` of the kept states per qn block (largest sigma first, ties in
    original order), then the rest globally by sigma (stable)."""
    sset = np.asarray(sset)
    qn_t = [tuple(int(x) for x in np.atleast_1d(qn)) for qn in qnlist]
    remaining = list(range(len(qn_t)))
    nbasis = min(len(remaining), int(Mmax))
    picked = []
    if percent != 0:
        blocks = sorted(set(qn_t))
        per_block = int(nbasis * percent / len(blocks))
        for b in blocks:
            members = [i for i in remaining if qn_t[i] == b]
            members.sort(key=lambda i: -sset[i])            # stable: ties keep index order
            take = members[: min(per_block, len(members))]
            picked += take
            taken = set(take)
            remaining = [i for i in remaining if i not in taken]
    rest = nbasis - len(picked)
    remaining.sort(key=lambda i: -sset[i])
    picked += remaining[:rest]
    assert len(set(picked)) == len(picked)
    return picked


def select_basis(vset, sset, qnlist, compset, Mmax, percent=0.0):
    """mps/lib.py:253-322: returns (ms, mpsdim, mpsqn, compmps)."""
    sidx = select_basis_indices(sset, qnlist, Mmax, percent)
    sset = np.asarray(sset)
    ms = np.asarray(vset)[:, sidx].copy()
    comp = None
    if compset is not None:
        compset = np.asarray(compset)
        comp = np.zeros((compset.shape[0], len(sidx)), dtype=compset.dtype)
        for k, i in enumerate(sidx):
            if i < compset.shape[1]:
                comp[:, k] = compset[:, i] * sset[i]
    mpsqn = np.array([np.atleast_1d(qnlist[i]) for i in sidx]).reshape(len(sidx), -1)
    return ms, len(sidx), mpsqn, comp


# ---------------------------------------------------------------------------
# a minimal matrix-product-state container and the sweep drivers
# ---------------------------------------------------------------------------

@dataclass
class MpsState:
    """The subset of mps/mp.py:60-77 state that the sweeps read and write."""
    sites: List[np.ndarray]
    qn: List[np.ndarray]              # nsite+1 arrays (D_i, qn_size)
    qnidx: int
    qntot: np.ndarray
    to_right: bool
    sigmaqn: List[np.ndarray]         # per site (d, qn_size)
    coeff: complex = 1.0
    krylov_dims: List[int] = field(default_factory=list)
    krylov_margins: List[list] = field(default_factory=list)     # per solve: expm_krylov's ``margins``

    def copy(self):
        return MpsState([s.copy() for s in self.sites], [np.array(q).copy() for q in self.qn],
                        self.qnidx, np.array(self.qntot).copy(), self.to_right,
                        self.sigmaqn, self.coeff, [])

    @property
    def nsite(self):
        return len(self.sites)

    @property
    def bond_dims(self):
        return [s.shape[0] for s in self.sites] + [self.sites[-1].shape[-1]]

    def iter_idx_list(self, full):
        """mps/mp.py:230-243."""
        n = self.nsite
        if self.to_right:
            return range(self.qnidx, n if full else n - 1)
        return range(self.qnidx, -1 if full else 0, -1)

    def switch_direction(self):
        """mps/mp.py:297-306."""
        if self.to_right:
            self.qnidx, self.to_right = self.nsite - 1, False
        else:
            self.qnidx, self.to_right = 0, True

    def move_qnidx(self, dst):
        """mps/mp.py:159-172."""
        n = self.nsite
        for idx in range(self.qnidx + 1, n + 1):
            self.qn[idx] = self.qntot - self.qn[idx]
        for idx in range(n, dst, -1):
            self.qn[idx] = self.qntot - self.qn[idx]
        self.qnidx = dst


def build_environ(sites, mpo, domain, sites_conj=None):
    """mps/lib.py:28-54: dict {(domain, idx): tensor} with the all-ones sentinels."""
    n = len(sites)
    env = {("L", -1): np.ones((1, 1, 1)), ("R", n): np.ones((1, 1, 1))}
    doms = ["L", "R"] if domain is None else [domain]
    for dom in doms:
        t = np.ones((1, 1, 1))
        rng = range(0, n - 1) if dom == "L" else range(n - 1, 0, -1)
        for i in rng:
            cj = None if sites_conj is None else sites_conj[i]
            t = contract_one_site(t, sites[i], mpo[i], dom, ms_conj=cj)
            env[(dom, i)] = t
    return env


def mps_dot(bra_conj_sites, ket_sites):
    """<bra|ket> where ``bra_conj_sites`` are already conjugated (mps/mp.py:933-956)."""
    e = np.ones((1, 1))
    for b, k in zip(bra_conj_sites, ket_sites):
        t = np.tensordot(e, k, 1)
        nd = k.ndim - 1
        e = np.tensordot(t, b, (list(range(nd)), list(range(nd)))).T
    return complex(e[0, 0])


def mp_norm(sites):
    """mps/mp.py:354-372."""
    r = mps_dot([s.conj() for s in sites], sites).real
    return float(np.sqrt(max(r, 0.0)))


def expectation(sites, mpo, sites_conj=None):
    """<psi|O|psi> via the R environment and the closing contraction of
    mps/mps.py:450-466, 471-525."""
    if sites_conj is None:
        sites_conj = [s.conj() for s in sites]
    n = len(sites)
    r = np.ones((1, 1, 1))
    for i in range(n - 1, 0, -1):
        r = contract_one_site(r, sites[i], mpo[i], "R", ms_conj=sites_conj[i])
    l = np.ones((1, 1, 1))
    full = contract_one_site(l, sites[0], mpo[0], "L", ms_conj=sites_conj[0])
    val = np.tensordot(full, r, ([0, 1, 2], [0, 1, 2]))
    val = complex(val)
    return float(val.real) if np.isclose(val.imag, 0) else val


def normalize(state: MpsState, kind):
    """mps/mps.py:2025-2059."""
    nrm = mp_norm(state.sites)
    if kind == "mps_and_coeff":
        state.coeff = state.coeff / abs(state.coeff)
    elif kind == "mps_norm_to_coeff":
        state.coeff = state.coeff * nrm
    elif kind != "mps_only":
        raise ValueError(kind)
    # scale() multiplies the qn-centre site (mp.py scale)
    state.sites[state.qnidx] = state.sites[state.qnidx] * (1.0 / nrm)
    return state


def tdvp_ps_step(state: MpsState, mpo, dt, normalize_after=True, max_updates=None, timings=None,
                 env_domain=None) -> MpsState:
    """One ``Mps.evolve`` with EvolveMethod.tdvp_ps and the Krylov solver:
    mps/mps.py:1267-1404 followed by normalize (mps.py:644-662).

    ``max_updates`` / ``timings`` serve the bounded CPU-baseline sample of bench.py: stop after that
    many site updates and append (site, shape, seconds) per update.  ``env_domain`` restricts the
    initial environment construction to the direction the first half sweep needs."""
    import time as _time
    imag_time = (complex(dt).imag != 0)
    st = state.copy()
    if not imag_time:
        st.sites = [s.astype(complex) for s in st.sites]
    n = st.nsite
    env = build_environ(st.sites, mpo, env_domain)
    dims = []
    margins = []
    done = 0

    def _m():
        margins.append([])
        return margins[-1]

    for _ in range(2):
        for i in st.iter_idx_list(full=True):
            if max_updates is not None and done >= max_updates:
                st.krylov_dims, st.krylov_margins = dims, margins
                return st
            done += 1
            _t0 = _time.perf_counter()
            system = "L" if st.to_right else "R"
            l = env[("L", i - 1)]
            r = env[("R", i + 1)]
            shape = st.sites[i].shape
            w = mpo[i]
            c, k = expm_krylov(lambda y: hop_apply(l, r, [w], y.reshape(shape)).ravel(),
                               -1j * dt / 2, st.sites[i].ravel(), margins=_m())
            dims.append(k)
            c = c.reshape(shape)
            qnbigl, qnbigr, _ = get_big_qn(st.qn[i], st.qn[i + 1], [st.sigmaqn[i]], st.to_right)
            u, qnl, v, qnr = svd_qn(c, qnbigl, qnbigr, st.qntot, QR=True, system=system,
                                    full_matrices=False)
            vt = v.T
            if (not st.to_right) and i != 0:
                st.sites[i] = vt.reshape((-1,) + tuple(shape[1:]))
                st.qn[i] = np.array(qnr).reshape(-1, len(st.qntot))
                st.qnidx = i - 1
                r = contract_one_site(r, st.sites[i], w, "R")
                env[("R", i)] = r
                b, k = expm_krylov(lambda y: hop_apply(l, r, [], y.reshape(u.shape)).ravel(),
                                   1j * dt / 2, u.ravel(), margins=_m())
                dims.append(k)
                st.sites[i - 1] = np.tensordot(st.sites[i - 1], b.reshape(u.shape), axes=(-1, 0))
            elif st.to_right and i != n - 1:
                st.sites[i] = u.reshape(tuple(shape[:-1]) + (-1,))
                st.qn[i + 1] = np.array(qnl).reshape(-1, len(st.qntot))
                st.qnidx = i + 1
                l = contract_one_site(l, st.sites[i], w, "L")
                env[("L", i)] = l
                b, k = expm_krylov(lambda y: hop_apply(l, r, [], y.reshape(vt.shape)).ravel(),
                                   1j * dt / 2, vt.ravel(), margins=_m())
                dims.append(k)
                st.sites[i + 1] = np.tensordot(b.reshape(vt.shape), st.sites[i + 1], axes=(1, 0))
            else:
                st.sites[i] = c
            if timings is not None:
                timings.append((i, tuple(shape), _time.perf_counter() - _t0))
        st.switch_direction()
    st.krylov_margins = margins
    st.krylov_dims = dims
    if normalize_after:
        normalize(st, "mps_and_coeff" if imag_time else "mps_only")
    return st


def push_cano(state: MpsState, idx):
    """mps/mp.py:890-908 + _update_ms (245-295) for an MPS (QR, no sigma)."""
    mt = state.sites[idx]
    qnbigl, qnbigr, _ = get_big_qn(state.qn[idx], state.qn[idx + 1], [state.sigmaqn[idx]], state.to_right)
    system = "L" if state.to_right else "R"
    u, qnl, v, qnr = svd_qn(mt, qnbigl, qnbigr, state.qntot, QR=True, system=system, full_matrices=False)
    _update_ms(state, idx, u, v.T, None, qnl, qnr, None)


def _update_ms(state: MpsState, idx, u, vt, sigma, qnl, qnr, m_trunc):
    """mps/mp.py:245-295 for an MPS."""
    if m_trunc is None:
        m_trunc = u.shape[1]
    u = u[:, :m_trunc]
    vt = vt[:m_trunc, :]
    q = len(state.qntot)
    if sigma is not None:
        sigma = np.asarray(sigma)[:m_trunc]
        if state.to_right:
            vt = sigma[:, None] * vt
        else:
            u = u * sigma[None, :]
    pdim = state.sites[idx].shape[1:-1]
    if state.to_right:
        state.sites[idx + 1] = np.tensordot(vt, state.sites[idx + 1], axes=1)
        state.sites[idx] = u.reshape((-1,) + tuple(pdim) + (m_trunc,)).copy()
        state.qn[idx + 1] = np.array(qnl[:m_trunc]).reshape(m_trunc, q)
        state.qnidx = idx + 1
    else:
        state.sites[idx - 1] = np.tensordot(state.sites[idx - 1], u, axes=1)
        state.sites[idx] = vt.reshape((m_trunc,) + tuple(pdim) + (-1,)).copy()
        state.qn[idx] = np.array(qnr[:m_trunc]).reshape(m_trunc, q)
        state.qnidx = idx - 1


def canonicalise(state: MpsState):
    """mps/mp.py:910-922."""
    idx = None
    for idx in state.iter_idx_list(full=False):
        push_cano(state, idx)
    state.switch_direction()
    return state


def compress(state: MpsState, criteria="threshold", threshold=1e-3, max_dims=None, temp_m_trunc=None):
    """SVD sweep of mps/mp.py:437-511 (state must be canonicalised towards the sweep start)."""
    system = "L" if state.to_right else "R"
    s_list = []
    for idx in state.iter_idx_list(full=False):
        qnbigl, qnbigr, _ = get_big_qn(state.qn[idx], state.qn[idx + 1], [state.sigmaqn[idx]], state.to_right)
        u, s, qnl, v, _, qnr = svd_qn(state.sites[idx], qnbigl, qnbigr, state.qntot, system=system,
                                      full_matrices=False)
        s_list.append(s)
        bond = idx + 1 if state.to_right else idx
        if temp_m_trunc is None:
            md = None if max_dims is None else max_dims[bond]
            m = compute_m_trunc(s, criteria, threshold, md)
        else:
            m = temp_m_trunc[bond] if np.ndim(temp_m_trunc) else temp_m_trunc
            m = min(int(m), len(s))
        _update_ms(state, idx, u, v.T, s, qnl, qnr, m)
    state.switch_direction()
    return state, s_list


def update_mps_2site(state: MpsState, cstruct, cidx, criteria, threshold, max_dims, percent=0.0, rng=None):
    """MatrixProduct._update_mps for a two-site centre and a single state, mps/mp.py:651-888."""
    qnbigl, qnbigr, _ = get_big_qn(state.qn[cidx[0]], state.qn[cidx[1] + 1],
                                   [state.sigmaqn[cidx[0]], state.sigmaqn[cidx[1]]], state.to_right)
    system = "L" if state.to_right else "R"
    U, SU, qnl, V, SV, qnr = svd_qn(cstruct, qnbigl, qnbigr, state.qntot, system=system, rng=rng)
    q = len(state.qntot)
    if state.to_right:
        bond = cidx[0] + 1
        m = compute_m_trunc(SU, criteria, threshold, None if max_dims is None else max_dims[bond])
        ms, dim, msqn, comp = select_basis(U, SU, qnl, V, m, percent)
        state.sites[cidx[0]] = ms.reshape(qnbigl.shape[:-1] + (dim,))
        state.sites[cidx[1]] = np.moveaxis(comp.reshape(qnbigr.shape[:-1] + (dim,)), -1, 0)
        state.qnidx = cidx[1]
    else:
        bond = cidx[1]
        m = compute_m_trunc(SV, criteria, threshold, None if max_dims is None else max_dims[bond])
        ms, dim, msqn, comp = select_basis(V, SV, qnr, U, m, percent)
        state.sites[cidx[1]] = np.moveaxis(ms.reshape(qnbigr.shape[:-1] + (dim,)), -1, 0)
        state.sites[cidx[0]] = comp.reshape(qnbigl.shape[:-1] + (dim,))
        state.qnidx = cidx[0]
    state.qn[cidx[1]] = np.asarray(msqn).reshape(dim, q)


def tdvp_ps2_step(state: MpsState, mpo, dt, criteria="fixed", threshold=1e-3, max_dims=None,
                  normalize_after=True) -> MpsState:
    """One ``Mps.evolve`` with EvolveMethod.tdvp_ps2 (Krylov solver), mps/mps.py:1406-1517."""
    st = state.copy()
    st.sites = [s.astype(complex) for s in st.sites]
    n = st.nsite
    env = build_environ(st.sites, mpo, None)
    dims = []
    for _ in range(2):
        for imps in st.iter_idx_list(full=False):
            if st.to_right:
                lidx, c0, c1, ridx = imps - 1, imps, imps + 1, imps + 2
                c2, last = c1, n - 2
            else:
                lidx, c0, c1, ridx = imps - 2, imps - 1, imps, imps + 1
                c2, last = c0, 1
            l, r = env[("L", lidx)], env[("R", ridx)]
            ms2 = np.tensordot(st.sites[c0], st.sites[c1], axes=1)
            shape = ms2.shape
            c, k = expm_krylov(lambda y: hop_apply(l, r, [mpo[c0], mpo[c1]], y.reshape(shape)).ravel(),
                               -1j * dt / 2, ms2.ravel())
            dims.append(k)
            update_mps_2site(st, c.reshape(shape), [c0, c1], criteria, threshold, max_dims)
            if imps == last:
                continue
            if st.to_right:
                l = contract_one_site(l, st.sites[lidx + 1], mpo[lidx + 1], "L")
                env[("L", lidx + 1)] = l
            else:
                r = contract_one_site(r, st.sites[ridx - 1], mpo[ridx - 1], "R")
                env[("R", ridx - 1)] = r
            ms1 = st.sites[c2]
            b, k = expm_krylov(lambda y: hop_apply(l, r, [mpo[c2]], y.reshape(ms1.shape)).ravel(),
                               1j * dt / 2, ms1.ravel())
            dims.append(k)
            st.sites[c2] = b.reshape(ms1.shape)
            push_cano(st, c2)
        st.switch_direction()
    st.krylov_dims = dims
    if normalize_after:
        normalize(st, "mps_only")
    return st
