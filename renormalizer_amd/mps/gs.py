"""DMRG ground-state optimisation on the device.

Counterpart of renormalizer/mps/gs.py: ``optimize_mps`` (:54-171) macro loop over the ``procedure`` of
``OptimizeConfig``, ``single_sweep`` (:174-304) over the sites, the dense solver for small centres
``get_ham_direct`` / ``eigh_direct`` (:307-407) and the iterative set-up ``get_ham_iterative`` / ``eigh_iterative``
(:410-576).  One difference by design: the quantum-number mask is kept as a dense 0/1 weight vector on the device
(the reference compresses vectors to the allowed entries on the host, gs.py:260, 520-523 - same iterates, no host
round trip).  ``nroots > 1`` (state-averaged DMRG) uses the block Davidson of the engine and the
averaged-density-matrix update of ``Mps._update_mps``.  ``omega`` (excited states through the (H - omega)^2
functional, gs.py:106-112) stacks H - omega twice between bra and ket: two-layer environments
(``contract_one_site_multi_mpo``) and two-layer centre problems (``hop_expr(twolayer=True)``)."""
import logging

import numpy as np

from ..engine import get_engine, idx1
from ..lib.davidson import davidson, davidson_multi
from ..utils import CompressConfig, CompressCriteria
from .hop_expr import hop_expr
from .lib import Environ
from .svd_qn import get_qn_mask

logger = logging.getLogger("renormalizer_amd")


def _hdiag(eng, l, r, cmo, twolayer=False):
    """Diagonal of the effective Hamiltonian (gs.py:423-477) from strided diagonal views of L and R and the
    physical-diagonal parts of the MPO sites: one layer  ba,bcg,gf->acf / ba,bce,edg,gf->acdf ; two layers
    abca,bdef,cedg,hfgh->adh / abca,bdef,cedg,fhij,gihk,ljkl->adhl, where the pair of stacked MPO bonds is treated as
    one channel and the per-site factor sum_e W[b,d,e,f] W[c,e,d,g] (KBs) is formed on the host.  Returns a float64
    device tensor."""
    cplx = l.is_complex or r.is_complex or any(w.is_complex for w in cmo)
    dt = np.complex128 if cplx else np.float64
    Dl, Dr = l.shape[0], r.shape[0]
    wl = int(np.prod(l.shape[1:-1]))           # w or w * w
    wr = int(np.prod(r.shape[1:-1]))
    wd = []                                    # per site: (channels in, d, channels out), diagonal in the physical leg
    for w in cmo:
        wh = w.to_host()
        if twolayer:
            x = np.einsum("bdef,cedg->bcdfg", wh, wh)
            wd.append(x.reshape(wh.shape[0] ** 2, wh.shape[1], wh.shape[3] ** 2))
        else:
            wd.append(np.einsum("bddf->bdf", wh))
    wdev = [eng.asdevice(np.ascontiguousarray(x)) for x in wd]
    d0, w0r = wd[0].shape[1], wd[0].shape[2]
    # X1[a,(c,g)] = sum_b L[a,b,a] wd0[b,c,g]
    x1 = eng.empty((Dl, d0 * w0r), dt)
    eng.gemm(l, wdev[0], x1, idx1(Dl, wl * Dl + 1), idx1(wl, Dl), idx1(wl, d0 * w0r), idx1(d0 * w0r, 1),
             idx1(Dl, d0 * w0r), idx1(d0 * w0r, 1))
    if len(cmo) == 1:
        # hd[(a,c),f] = sum_g X1[(a,c),g] R[f,g,f]
        out = eng.empty((Dl, d0, Dr), dt)
        eng.gemm(x1, r, out, idx1(Dl * d0, wr), idx1(wr, 1), idx1(wr, Dr), idx1(Dr, wr * Dr + 1),
                 idx1(Dl * d0, Dr), idx1(Dr, 1))
    else:
        wm, d1 = wd[1].shape[0], wd[1].shape[1]
        # X2[(e,d),f] = sum_g wd1[e,d,g] R[f,g,f]
        x2 = eng.empty((wm * d1, Dr), dt)
        eng.gemm(wdev[1], r, x2, idx1(wm * d1, wr), idx1(wr, 1), idx1(wr, Dr), idx1(Dr, wr * Dr + 1),
                 idx1(wm * d1, Dr), idx1(Dr, 1))
        # hd[(a,c),(d,f)] = sum_e X1[(a,c),e] X2[e,(d,f)]
        out = eng.empty((Dl, d0, d1, Dr), dt)
        eng.gemm(x1, x2, out, idx1(Dl * d0, wm), idx1(wm, 1), idx1(wm, d1 * Dr), idx1(d1 * Dr, 1),
                 idx1(Dl * d0, d1 * Dr), idx1(d1 * Dr, 1))
    if cplx:
        re = eng.empty(out.shape, np.float64)
        eng._check(eng.lib.mpse_real_part(eng.ctx, re.ptr, out.ptr, out.size))
        return re
    return out


def _sign_fix(c):
    """gs.py:372-380"""
    return c / np.sign(c[np.abs(c).argmax()])


def eigh_direct(mps, qn_mask, ltensor, rtensor, cmo, twolayer=False):
    """gs.py:307-407: dense projected operator (one engine call, ``Hop.dense``), restricted to the symmetry-allowed
    entries, diagonalised on the host with LAPACK like the reference.  Returns (e, c) with c as device tensor(s) of the
    centre's shape."""
    eng = get_engine()
    cshape = qn_mask.shape
    hop = hop_expr(ltensor, rtensor, cmo, cshape, twolayer)
    ham = hop.dense()
    flat = qn_mask.ravel()
    ham = ham[flat][:, flat]
    ham = (ham + ham.conj().T) / 2
    # gs.py:397-398: the spectrum of inverse * H (inverse = -1: the HIGHEST state of H; the value returned is that of
    # the scaled operator, as in the reference)
    w, v = np.linalg.eigh(ham * float(mps.optimize_config.inverse))
    nroots = mps.optimize_config.nroots

    def expand(col):
        full = np.zeros(flat.shape, dtype=v.dtype)
        full[flat] = _sign_fix(col)
        return eng.asdevice(full.reshape(cshape))

    if nroots == 1:
        return float(w[0]), expand(v[:, 0]), 0
    k = min(nroots, v.shape[1])
    return [float(x) for x in w[:k]], [expand(v[:, i]) for i in range(k)], 0


def eigh_iterative(mps, qn_mask, ltensor, rtensor, cmo, cguess, twolayer=False):
    """gs.py:486-576: ``optimize_config.algo`` selects the iterative eigensolver of the centre problem.

    "davidson"  the reference's default (PySCF-style Davidson: tol 1e-12 on the eigenvalue, 1e-6 on the residual,
                preconditioner x / (hdiag - e + 1e-4), space 12 + 3 (nroots - 1), gs.py:533-538);
    "primme"    the reference hands the problem to the PRIMME library (gs.py:552-569: ``primme.eigsh(which="SA",
                tol=1e-6, method="PRIMME_DYNAMIC")`` with the diagonal preconditioner 1 / (hdiag + 1e-4)).  PRIMME's
                dynamic method is a generalised Davidson with thick restart; here the same engine iteration runs with
                PRIMME's settings: convergence on the residual alone (|r| < 1e-6, the eigenvalue change is not
                tested), a restart space of max(15, 2 nroots + 7) vectors and the shift-free preconditioner
                (``shift`` large against ``e`` is not available, so x / (hdiag - e + 1e-4) is kept: same fixed
                point).  Same eigenpairs to the solver tolerance; iteration counts differ from the library's."""
    eng = get_engine()
    inverse = float(mps.optimize_config.inverse)
    if inverse != 1.0:
        # gs.py:470, 521: hdiag and every H c are multiplied by ``inverse`` - the operator is linear in its first MPO
        # site, so a scaled copy of that site gives inverse * H_eff for the matvec, the diagonal and the preconditioner
        # alike.  (Two layers are (H - omega)^2 with the SAME sites in both: a scaled copy would enter squared.)
        if twolayer:
            raise NotImplementedError("optimize_config.inverse != 1 with omega (two-layer environments)")
        cmo = [cmo[0].copy().scale_(inverse)] + list(cmo[1:])
    algo = mps.optimize_config.algo
    if algo not in ("davidson", "primme"):
        raise ValueError(f"optimize_config.algo = {algo!r}: 'davidson', 'primme' (iterative) or 'direct'")
    cshape = qn_mask.shape
    hop = hop_expr(ltensor, rtensor, cmo, cshape, twolayer)
    hdiag = _hdiag(eng, hop.l, hop.r, hop.cmo, twolayer)
    mask = eng.asdevice(qn_mask.astype(np.float64))
    nroots = mps.optimize_config.nroots
    if hop.operator_is_complex:
        # complex MPO / environments with a real centre: the iteration runs in complex128 (NumPy promotes silently
        # in the reference, gs.py:520-538)
        cguess = cguess.to_complex() if nroots == 1 else [g.to_complex() for g in cguess]
    # the engine stops a root when (|de| < tol and |r| < sqrt(tol)) or |r| < 1e-14: tol = 1e-12 is the reference's
    # Davidson rule; PRIMME's tol = 1e-6 bounds the residual only, i.e. sqrt(tol) = 1e-6 with the energy test open
    if algo == "primme":
        tol, space = -1e-6, max(15, 2 * nroots + 7)      # (negative: residual-only test, include/mpsengine.h)
    else:
        tol, space = 1e-12, (12 if nroots == 1 else None)
    if nroots == 1:
        e, c, ncyc = davidson(hop, cguess.reshape(cshape), hdiag, mask=mask, tol=tol, max_cycle=100,
                              max_space=space, lindep=1e-14)
        return e, c, ncyc
    guesses = [g.reshape(cshape) for g in cguess]
    e, c, ncyc = davidson_multi(hop, guesses, hdiag, nroots, mask=mask, tol=tol, max_cycle=100, max_space=space,
                                lindep=1e-14)
    return e, c, ncyc


def single_sweep(mps, mpo, environ, percent, last_opt_e_idx, omega=None):
    """gs.py:174-304; ``omega``: the centre problems are those of (H - omega)^2 on two-layer environments."""
    eng = get_engine()
    method = mps.optimize_config.method
    nroots = mps.optimize_config.nroots
    res_mps = None
    micro = []
    hops = []
    averaged_ms = None
    rng = np.random.default_rng(mps.optimize_config.__dict__.get("guess_seed", 0))
    for imps in mps.iter_idx_list(full=True):
        if method == "2site" and ((mps.to_right and imps == mps.site_num - 1) or ((not mps.to_right) and imps == 0)):
            break
        lmethod, rmethod = ("System", "Enviro") if mps.to_right else ("Enviro", "System")
        if method == "1site":
            lidx, cidx, ridx = imps - 1, [imps], imps + 1
        elif mps.to_right:
            lidx, cidx, ridx = imps - 1, [imps, imps + 1], imps + 2
        else:
            lidx, cidx, ridx = imps - 2, [imps - 1, imps], imps + 1
        operator = mpo if omega is None else [mpo, mpo]
        ltensor = environ.GetLR("L", lidx, mps, operator, itensor=None, method=lmethod)
        rtensor = environ.GetLR("R", ridx, mps, operator, itensor=None, method=rmethod)
        qnbigl, qnbigr, qnmat = mps._get_big_qn(cidx)
        qn_mask = get_qn_mask(qnmat, mps.qntot)
        cmo = [mpo.device(i, eng) for i in cidx]
        def two_site(a, b):
            return eng.matmul(a.reshape(-1, a.shape[-1]), b.reshape(b.shape[0], -1))

        if nroots == 1:
            guess = mps[cidx[0]] if method == "1site" else two_site(mps[cidx[0]], mps[cidx[1]])
        else:
            # gs.py:262-276: the rotated roots of the previous centre, padded with random vectors
            guess = []
            for ms in (averaged_ms or [mps[cidx[0]] if method == "1site" else None]):
                if ms is None:
                    guess.append(two_site(mps[cidx[0]], mps[cidx[1]]))
                elif method == "1site":
                    guess.append(ms)
                elif mps.to_right:
                    guess.append(two_site(ms, mps[cidx[1]]))
                else:
                    guess.append(two_site(mps[cidx[0]], ms))
            while len(guess) < nroots:
                guess.append(eng.asdevice((rng.random(qn_mask.shape) - 0.5) * qn_mask))
        # gs.py:245-247: small centres are diagonalised densely
        if int(np.prod(qn_mask.shape)) < 1000 or mps.optimize_config.algo == "direct":
            e, c, ncyc = eigh_direct(mps, qn_mask, ltensor, rtensor, cmo, omega is not None)
        else:
            e, c, ncyc = eigh_iterative(mps, qn_mask, ltensor, rtensor, cmo, guess, omega is not None)
        if nroots > 1:
            # fewer allowed states than roots (dense solver), or a block Davidson that lost roots to linearly
            # dependent guesses: pad like the random guesses so that the state average always sees nroots tensors
            e, c = list(e), list(c)
            if not c:
                raise RuntimeError(f"DMRG centre {cidx}: the eigensolver returned no eigenpair")
            while len(c) < nroots:
                c.append(eng.asdevice(((rng.random(qn_mask.shape) - 0.5) * qn_mask).astype(c[0].dtype)))
                e.append(e[-1])
        hops.append(ncyc)
        micro.append((e, cidx))
        if nroots == 1:
            cstruct = c.reshape(qn_mask.shape)
            if cidx == last_opt_e_idx:
                res_mps = mps.copy()
                res_mps._update_mps(cstruct, cidx, qnbigl, qnbigr, percent)
            mps._update_mps(cstruct, cidx, qnbigl, qnbigr, percent)
            if mps.compress_config.ofs is not None:
                mpo.try_swap_site(mps.model, mps.compress_config.ofs_swap_jw)     # gs.py:300-301
        else:
            cstruct = [x.reshape(qn_mask.shape) for x in c]
            if cidx == last_opt_e_idx:
                res_mps = [mps.copy() for _ in cstruct]
                for r, cs in enumerate(cstruct):
                    res_mps[r]._update_mps(cs, cidx, qnbigl, qnbigr, percent)
            averaged_ms = mps._update_mps(cstruct, cidx, qnbigl, qnbigr, percent)
    mps._switch_direction()
    logger.debug(f"Davidson cycles per site: {hops}")
    return micro, res_mps


def optimize_mps(mps, mpo, omega: float = None):
    """DMRG ground state (gs.py:54-171).  Returns (list of the lowest energy of every macro sweep, optimised mps).
    The input mps is overwritten, as in the reference."""
    if omega is not None:
        # gs.py:106-112: H - omega, stacked twice between bra and ket (two-layer environments and centre problems)
        from .mpo import Mpo
        mpo = mpo.add(Mpo.identity(mpo.model).scale(-omega))
    nroots = mps.optimize_config.nroots
    assert mps.optimize_config.method in ["2site", "1site"]
    if mps.is_left_canonical:
        mps.ensure_right_canonical()
        env = "R"
    else:
        mps.ensure_left_canonical()
        env = "L"
    compress_config_bk = mps.compress_config
    environ = Environ(mps, mpo if omega is None else [mpo, mpo], env)
    macro = []
    opt_e_idx = None
    res_mps = None
    for isweep, (cfg, percent) in enumerate(mps.optimize_config.procedure):
        if isinstance(cfg, CompressConfig):
            mps.compress_config = cfg
        elif isinstance(cfg, (int, np.integer)):
            mps.compress_config = CompressConfig(criteria=CompressCriteria.fixed, max_bonddim=int(cfg))
        else:
            raise TypeError(cfg)
        micro, res, = single_sweep(mps, mpo, environ, percent, opt_e_idx, omega)
        if res is not None:
            res_mps = res
        # gs.py:136-160: the centre with the lowest (summed, for several roots) energy marks the optimal position
        # (energies of several roots compare like the reference's lists: lowest root first)
        opt_e = min(micro, key=lambda x: x[0] if nroots == 1 else tuple(x[0]))
        macro.append(opt_e[0])
        opt_e_idx = opt_e[1]
        logger.debug(f"{isweep + 1} sweeps are finished, lowest energy = {opt_e[0]}")
        if isweep > 0 and percent == 0:
            v1, v2 = sorted(macro, key=lambda x: x if nroots == 1 else tuple(x))[:2]
            if np.allclose(v1, v2, rtol=mps.optimize_config.e_rtol, atol=mps.optimize_config.e_atol):
                logger.info("DMRG has converged!")
                break
    else:
        logger.warning("DMRG did not converge! Please increase the procedure!")
    if res_mps is None:          # a single macro sweep: nothing was recorded at the optimal centre yet
        res_mps = mps.copy() if nroots == 1 else [mps.copy() for _ in range(nroots)]
    if nroots == 1:
        res_mps = res_mps.normalize("mps_only").ensure_left_canonical().canonicalise()
        res_mps.compress_config = compress_config_bk
    else:
        res_mps = [m.normalize("mps_only").ensure_left_canonical().canonicalise() for m in res_mps]
        for m in res_mps:
            m.compress_config = compress_config_bk
    return macro, res_mps
