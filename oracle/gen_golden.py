#!/usr/bin/env python3
"""Generate golden vectors from the real reference.  DEV CONTAINER ONLY.

TEST INFRASTRUCTURE: imports the read-only Python reference from
/root/reference (with the stand-in modules in oracle/shims for the two absent
third-party packages), runs the reference's own functions at the seams of the
hot path and stores inputs + outputs as small .npz files
under tests/golden/.  Only data is written - no reference source.

    cd /tmp && python /root/repo/oracle/gen_golden.py [seams|tdvp|mpo|ps2|adaptive|pc_adaptive|fmo|thermal|obs|pc_rk|vmf|ofs|thermofield|dmrg|all]

The GPU box never runs this (no /root/reference there); tests read the .npz.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
os.environ.setdefault("RENO_NUM_THREADS", "4")
os.environ.setdefault("RENO_LOG_LEVEL", "40")
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))

import numpy as np  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def _rand(rng, shape, cplx):
    a = rng.standard_normal(shape)
    if cplx:
        a = a + 1j * rng.standard_normal(shape)
    return a


def gen_seams():
    from renormalizer.mps.lib import contract_one_site, select_basis
    from renormalizer.mps.hop_expr import hop_expr
    from renormalizer.lib.krylov.krylov import expm_krylov
    from renormalizer.mps import svd_qn as ref_svd
    from renormalizer.utils.configs import CompressConfig, CompressCriteria

    rng = np.random.default_rng(20260928)
    out = {}

    # ---- a1: contract_one_site  (lib.py:169-250)
    cases = []
    k = 0
    for cplx in (False, True):
        for anc in (False, True):
            for dom in ("L", "R"):
                Dl, Dr, d, da, wl, wr = 5, 7, 3, 2, 4, 3
                shp = (Dl, d, da, Dr) if anc else (Dl, d, Dr)
                ms = _rand(rng, shp, cplx)
                bra = _rand(rng, shp, cplx)          # independent bra (transition amplitude use)
                mo = _rand(rng, (wl, d, d, wr), False)
                if dom == "L":
                    env = _rand(rng, (Dl, wl, Dl), cplx)
                else:
                    env = _rand(rng, (Dr, wr, Dr), cplx)
                res = contract_one_site(env, ms, mo, dom, ms_conj=bra.conj())
                res2 = contract_one_site(env, ms, mo, dom)
                for nm, v in (("env", env), ("ms", ms), ("bra", bra), ("mo", mo), ("out", res), ("out_self", res2)):
                    out[f"c1s_{k}_{nm}"] = np.asarray(v)
                out[f"c1s_{k}_dom"] = np.array(dom)
                cases.append(k)
                k += 1
    out["c1s_n"] = np.array(k)

    # ---- a4: hop_expr  (hop_expr.py:57-115)
    k = 0
    for cplx in (False, True):
        for nsite in (0, 1, 2):
            for anc in ((False,) if nsite == 0 else (False, True)):
                Dl, Dr, d1, d2, da, wl, wm, wr = 6, 5, 3, 4, 2, 3, 4, 2
                if nsite == 0:
                    l = _rand(rng, (Dl, wl, Dl), cplx)
                    r = _rand(rng, (Dr, wl, Dr), cplx)
                    cmo, cshape = [], (Dl, Dr)
                elif nsite == 1:
                    l = _rand(rng, (Dl, wl, Dl), cplx)
                    r = _rand(rng, (Dr, wr, Dr), cplx)
                    cmo = [_rand(rng, (wl, d1, d1, wr), False)]
                    cshape = (Dl, d1, da, Dr) if anc else (Dl, d1, Dr)
                else:
                    l = _rand(rng, (Dl, wl, Dl), cplx)
                    r = _rand(rng, (Dr, wr, Dr), cplx)
                    cmo = [_rand(rng, (wl, d1, d1, wm), False), _rand(rng, (wm, d2, d2, wr), False)]
                    cshape = (Dl, d1, da, d2, da, Dr) if anc else (Dl, d1, d2, Dr)
                c = _rand(rng, cshape, cplx)
                expr = hop_expr(l, r, list(cmo), list(cshape))
                hc = expr(c)
                out[f"hop_{k}_l"], out[f"hop_{k}_r"], out[f"hop_{k}_c"], out[f"hop_{k}_out"] = l, r, c, np.asarray(hc)
                out[f"hop_{k}_nsite"] = np.array(nsite)
                for j, w in enumerate(cmo):
                    out[f"hop_{k}_w{j}"] = w
                k += 1
    out["hop_n"] = np.array(k)

    # ---- a5: expm_krylov  (krylov.py:27-82)
    k = 0
    for n, dt in ((1, -0.3j), (2, 0.2j), (7, -0.5j), (60, -0.4j), (150, -0.05j), (150, -0.3)):
        a = _rand(rng, (n, n), True)
        a = (a + a.conj().T) / 2 / max(1.0, np.sqrt(n))
        v = _rand(rng, (n,), True)
        res, nv = expm_krylov(lambda x: a @ x, dt, v)
        out[f"kry_{k}_a"], out[f"kry_{k}_v"], out[f"kry_{k}_dt"] = a, v, np.array(dt)
        out[f"kry_{k}_out"], out[f"kry_{k}_nvec"] = np.asarray(res), np.array(nv)
        k += 1
    out["kry_n"] = np.array(k)
    # large structured case: A = diag(x) + u u^H  (stored as x, u only)
    n = 800
    x = rng.standard_normal(n)
    u = _rand(rng, (n,), True) / np.sqrt(n)
    v = _rand(rng, (n,), True)
    res, nv = expm_krylov(lambda y: x * y + u * np.vdot(u, y), -0.2j, v)
    out["kryL_x"], out["kryL_u"], out["kryL_v"], out["kryL_dt"] = x, u, v, np.array(-0.2j)
    out["kryL_out"], out["kryL_nvec"] = np.asarray(res), np.array(nv)

    # ---- a6: svd_qn  (svd_qn.py:99-240)
    k = 0
    for cplx in (False, True):
        for (QR, system, full) in ((True, "L", False), (True, "R", False), (False, None, False), (False, None, True)):
            for qn_size in (1, 2):
                Dl, d, Dr = 9, 3, 8
                to_right = system != "R"
                while True:
                    hi = 3 if qn_size == 1 else 2
                    qnl = rng.integers(0, hi, size=(Dl, qn_size))
                    sig = rng.integers(0, 2, size=(d, qn_size))
                    qnr = rng.integers(0, hi, size=(Dr, qn_size))
                    qntot = np.full(qn_size, 3 if qn_size == 1 else 2)
                    if to_right:
                        qnbigl = ref_svd.add_outer(qnl, sig)
                        qnbigr = qnr
                    else:
                        qnbigl = qnl
                        qnbigr = ref_svd.add_outer(sig, qnr)
                    mask = ref_svd.get_qn_mask(ref_svd.add_outer(qnbigl, qnbigr), qntot)
                    if mask.sum() >= 20:
                        break
                c = _rand(rng, (Dl, d, Dr), cplx) * mask
                np.random.seed(7)
                res = ref_svd.svd_qn(c, qnbigl, qnbigr, qntot, QR=QR, system=system, full_matrices=full)
                pre = f"svd_{k}_"
                out[pre + "c"], out[pre + "qnbigl"], out[pre + "qnbigr"], out[pre + "qntot"] = c, qnbigl, qnbigr, qntot
                out[pre + "QR"], out[pre + "system"], out[pre + "full"] = np.array(QR), np.array(str(system)), np.array(full)
                if QR:
                    u, ql, v, qr_ = res
                    out[pre + "u"], out[pre + "v"] = u, v
                else:
                    u, su, ql, v, sv, qr_ = res
                    out[pre + "u"], out[pre + "v"], out[pre + "su"], out[pre + "sv"] = u, v, su, sv
                out[pre + "qnl"] = np.array(ql).reshape(len(ql), qn_size)
                out[pre + "qnr"] = np.array(qr_).reshape(len(qr_), qn_size)
                k += 1
    out["svd_n"] = np.array(k)

    # ---- a7: select_basis + compute_m_trunc  (lib.py:253-322, configs.py:196-219)
    k = 0
    for percent in (0, 0.1, 0.4):
        for mmax in (5, 12, 40):
            nrow, ncol = 14, 20
            u = _rand(rng, (nrow, ncol), True)
            vv = _rand(rng, (11, ncol - 3), True)        # compset may have fewer columns
            s = np.sort(rng.random(ncol))[::-1].copy()
            s[5] = s[4]                                   # a tie
            qn = [list(x) for x in rng.integers(0, 3, size=(ncol, 1))]
            ms, dim, mqn, comp = select_basis(u, s, qn, vv, mmax, percent=percent)
            pre = f"sel_{k}_"
            out[pre + "u"], out[pre + "s"], out[pre + "qn"], out[pre + "v"] = u, s, np.array(qn), vv
            out[pre + "mmax"], out[pre + "percent"] = np.array(mmax), np.array(percent)
            out[pre + "ms"], out[pre + "dim"], out[pre + "mqn"], out[pre + "comp"] = np.asarray(ms), np.array(dim), np.asarray(mqn), np.asarray(comp)
            k += 1
    out["sel_n"] = np.array(k)
    k = 0
    for crit, nm in ((CompressCriteria.threshold, "threshold"), (CompressCriteria.fixed, "fixed"), (CompressCriteria.both, "both")):
        for thr in (1e-3, 0.2):
            s = np.sort(rng.random(17) ** 4)[::-1].copy()
            cfg = CompressConfig(criteria=crit, threshold=thr, max_bonddim=9)
            cfg.set_bonddim(6)
            m = cfg.compute_m_trunc(s, 2, True)
            pre = f"mtr_{k}_"
            out[pre + "s"], out[pre + "crit"], out[pre + "thr"], out[pre + "maxdim"], out[pre + "m"] = s, np.array(nm), np.array(thr), np.array(9), np.array(m)
            k += 1
    out["mtr_n"] = np.array(k)
    np.savez_compressed(os.path.join(GOLD, "seams.npz"), **out)
    print("seams.npz written:", len(out), "arrays")


def _dump_mps(out, pre, mps):
    out[pre + "nsite"] = np.array(len(mps))
    for i in range(len(mps)):
        out[pre + f"site_{i}"] = np.asarray(mps[i].array)
    for i in range(len(mps) + 1):
        out[pre + f"qn_{i}"] = np.asarray(mps.qn[i]).reshape(len(mps.qn[i]), -1).astype(np.int64)
    out[pre + "qnidx"] = np.array(mps.qnidx)
    out[pre + "qntot"] = np.asarray(mps.qntot).astype(np.int64)
    out[pre + "to_right"] = np.array(bool(mps.to_right))
    out[pre + "coeff"] = np.array(complex(mps.coeff))


def _dump_mpo(out, pre, mpo):
    out[pre + "nsite"] = np.array(len(mpo))
    for i in range(len(mpo)):
        out[pre + f"w_{i}"] = np.asarray(mpo[i].array)


def _tdvp_run(model, mpo, init, obs_mpos, nsteps, dt, fname, extra=None):
    out = dict(extra or {})
    _dump_mpo(out, "mpo_", mpo)
    for j, o in enumerate(obs_mpos):
        _dump_mpo(out, f"obs{j}_", o)
    out["nobs"] = np.array(len(obs_mpos))
    for i, b in enumerate(model.basis):
        out[f"sigmaqn_{i}"] = np.asarray(b.sigmaqn).reshape(b.nbas, -1).astype(np.int64)
    _dump_mps(out, "init_", init)
    mps = init
    vals, energies, kdims, bdims, norms = [], [], [], [], []
    vals.append([mps.expectation(o) for o in obs_mpos])
    energies.append(mps.expectation(mpo))
    for step in range(nsteps):
        mps = mps.evolve(mpo, dt)
        vals.append([mps.expectation(o) for o in obs_mpos])
        energies.append(mps.expectation(mpo))
        bdims.append(mps.bond_dims)
        norms.append(mps.mp_norm)
        st = mps.evolve_config.stat
        kdims.append([st.nobs, st.minmax[0], st.minmax[1], st.mean])
        if step == 0:
            _dump_mps(out, "step1_", mps)
    out["dt"] = np.array(dt)
    out["obs_values"] = np.array(vals, dtype=complex).real
    out["energies"] = np.array(energies, dtype=complex).real
    out["krylov_stat"] = np.array(kdims)
    out["bond_dims"] = np.array(bdims)
    out["norms"] = np.array(norms)
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    print(fname, "obs[-1] =", out["obs_values"][-1], "krylov", kdims[-1])


def gen_tdvp():
    from renormalizer.model import Phonon, Mol, HolsteinModel, SpinBosonModel, Op
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, CompressConfig, EvolveConfig, EvolveMethod, CompressCriteria

    # --- Holstein chain, reduced headline config: std.yaml parameters
    nmol, pdim, D = 4, 4, 8
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    gs = Mps.ground_state(model, max_entangled=False)
    init = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(gs)
    e0 = Quantity(init.expectation(Mpo(model)))
    mpo = Mpo(model, offset=e0)
    init.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
    init.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    np.random.seed(9012)
    init = init.expand_bond_dimension(mpo)
    init.canonicalise()
    occ = [Mpo(model, Op(r"a^\dagger a", dof)) for dof in model.e_dofs]
    _tdvp_run(model, mpo, init, occ, 4, 10.0, "tdvp_holstein_small.npz",
              {"e0": np.array(e0.as_au())})

    # --- spin-boson, reduced config 2
    nmode, pdim, D = 5, 5, 10
    omegas = np.linspace(0.5, 4.0, nmode)
    cs = 0.3 / np.sqrt(np.arange(1, nmode + 1))
    ph_list = [Phonon.simple_phonon(Quantity(w), Quantity(c / w ** 2), pdim) for w, c in zip(omegas, cs)]
    model = SpinBosonModel(Quantity(0.0), Quantity(0.8), ph_list)
    mpo = Mpo(model)
    init = Mps.ground_state(model, False)
    init.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
    init.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    np.random.seed(9012)
    init = init.expand_bond_dimension(mpo, coef=1e-16, include_ex=False)
    sz = [Mpo(model, Op("sigma_z", "spin")), Mpo(model, Op("sigma_x", "spin"))]
    _tdvp_run(model, mpo, init, sz, 5, 0.1, "tdvp_sbm_small.npz")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("seams", "all"):
        gen_seams()
    if what in ("tdvp", "all"):
        gen_tdvp()


def gen_mpo():
    """Dense matrices + bond dimensions of reference MPOs for small models (pins the build's own
    MPO construction)."""
    from renormalizer.model import Phonon, Mol, HolsteinModel, SpinBosonModel, Op, Model, BasisHalfSpin
    from renormalizer.mps import Mpo
    from renormalizer.utils import Quantity
    out = {}
    # (1) Holstein 3 molecules x 2 modes (tests/parameter.py values, reduced phonon levels), scheme 3
    omega = [Quantity(106.51, "cm^{-1}"), Quantity(1555.55, "cm^{-1}")]
    dis = [Quantity(30.1370), Quantity(8.7729)]
    ph_list = [Phonon.simple_phonon(o, d, 3) for o, d in zip(omega, dis)]
    ph_list = [Phonon.simple_phonon(o, d, n) for o, d, n in zip(omega, dis, (3, 2))]
    j = np.array([[0.0, -0.1, -0.2], [-0.1, 0.0, -0.3], [-0.2, -0.3, 0.0]]) / 27.211386245988
    model = HolsteinModel([Mol(Quantity(2.67, "eV"), ph_list)] * 3, j, 3)
    mpo = Mpo(model)
    out["hol_dense"], out["hol_bond"] = mpo.todense(), np.array(mpo.bond_dims)
    mpo = Mpo(model, offset=Quantity(0.05))
    out["hol_off_dense"] = mpo.todense()
    mpo = Mpo.onsite(model, r"a^\dagger", dof_set={1})
    out["hol_adag_dense"], out["hol_adag_bond"] = mpo.todense(), np.array(mpo.bond_dims)
    mpo = Mpo(model, Op(r"a^\dagger a", 2))
    out["hol_occ_dense"] = mpo.todense()
    # (2) chain used by the headline config, 3 molecules
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), 4)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * 3, Quantity(3.0e-2), 3)
    mpo = Mpo(model)
    out["chain_dense"], out["chain_bond"] = mpo.todense(), np.array(mpo.bond_dims)
    # (3) spin-boson, 3 modes
    ph_list = [Phonon.simple_phonon(Quantity(w), Quantity(c / w ** 2), 3) for w, c in zip((0.5, 1.5, 3.0), (0.3, 0.2, 0.1))]
    model = SpinBosonModel(Quantity(0.1), Quantity(0.8), ph_list)
    mpo = Mpo(model)
    out["sbm_dense"], out["sbm_bond"] = mpo.todense(), np.array(mpo.bond_dims)
    # (4) README quick start
    model = Model([BasisHalfSpin(0), BasisHalfSpin(1)], Op("sigma_+ sigma_-", [0, 1]) + Op("sigma_+ sigma_-", [1, 0]))
    mpo = Mpo(model)
    out["qs_dense"], out["qs_bond"] = mpo.todense(), np.array(mpo.bond_dims)
    # (5) a longer-range spin model with a complex coefficient
    basis = [BasisHalfSpin(i) for i in range(5)]
    terms = []
    for i in range(5):
        terms.append(Op("Z", i, 0.3 + 0.1 * i))
        for k in range(i + 1, 5):
            terms.append(Op("X X", [i, k], 1.0 / (k - i) ** 2))
            terms.append(Op("sigma_+ sigma_-", [i, k], 0.2j / (k - i)))
            terms.append(Op("sigma_- sigma_+", [i, k], -0.2j / (k - i)))
    terms.append(Op("Z Z Z", [0, 2, 4], 0.7))
    model = Model(basis, terms)
    mpo = Mpo(model)
    out["lr_dense"], out["lr_bond"] = mpo.todense(), np.array(mpo.bond_dims)
    np.savez_compressed(os.path.join(GOLD, "mpo_dense.npz"), **out)
    print("mpo_dense.npz written; bond dims:", {k: v.tolist() for k, v in out.items() if k.endswith("bond")})


def gen_tdvp_ps2():
    """Two-site TDVP (mps.py:1406-1517) on the reduced headline model."""
    from renormalizer.model import Phonon, Mol, HolsteinModel, Op
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, CompressConfig, EvolveConfig, EvolveMethod, CompressCriteria
    nmol, pdim = 4, 4
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    gs = Mps.ground_state(model, max_entangled=False)
    init = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(gs)
    e0 = Quantity(init.expectation(Mpo(model)))
    mpo = Mpo(model, offset=e0)
    # a product state cannot grow its bonds under 2-site TDVP here (electron sites are separated by a phonon
    # site), so start from the expanded state like the reference's transport job does
    init.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=8)
    init.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps2)
    np.random.seed(9012)
    init = init.expand_bond_dimension(mpo)
    init.canonicalise()
    occ = [Mpo(model, Op(r"a^\dagger a", dof)) for dof in model.e_dofs]
    _tdvp_run(model, mpo, init, occ, 5, 10.0, "tdvp_ps2_holstein_small.npz")


def gen_adaptive():
    """Adaptive-step TDVP-PS (mps.py:46-115) on the reduced headline model: observables and the step guesses."""
    from renormalizer.model import Phonon, Mol, HolsteinModel, Op
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, CompressConfig, EvolveConfig, EvolveMethod, CompressCriteria
    nmol, pdim = 4, 4
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    gs = Mps.ground_state(model, max_entangled=False)
    init = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(gs)
    e0 = Quantity(init.expectation(Mpo(model)))
    mpo = Mpo(model, offset=e0)
    init.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=8)
    init.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps, adaptive=True, guess_dt=15.0, adaptive_rtol=5e-4)
    np.random.seed(1357)
    init = init.expand_bond_dimension(mpo)
    init.canonicalise()
    occ = [Mpo(model, Op(r"a^\dagger a", dof)) for dof in model.e_dofs]
    out = {}
    _dump_mpo(out, "mpo_", mpo)
    for j, o in enumerate(occ):
        _dump_mpo(out, f"obs{j}_", o)
    out["nobs"] = np.array(len(occ))
    for i, b in enumerate(model.basis):
        out[f"sigmaqn_{i}"] = np.asarray(b.sigmaqn).reshape(b.nbas, -1).astype(np.int64)
    _dump_mps(out, "init_", init)
    mps = init
    vals, guess, norms = [[mps.expectation(o) for o in occ]], [], []
    for step in range(3):
        mps = mps.evolve(mpo, 40.0)
        vals.append([mps.expectation(o) for o in occ])
        guess.append(mps.evolve_config.guess_dt)
        norms.append(mps.mp_norm)
    out["dt"] = np.array(40.0)
    out["obs_values"] = np.array(vals, dtype=complex).real
    out["guess_dt"] = np.array(guess, dtype=float)
    out["norms"] = np.array(norms)
    np.savez_compressed(os.path.join(GOLD, "tdvp_adaptive_holstein_small.npz"), **out)
    print("tdvp_adaptive_holstein_small.npz", out["obs_values"][-1], out["guess_dt"])


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("adaptive",)):
    gen_adaptive()


def gen_pc_adaptive():
    """Adaptive propagate & compress (mps.py:794-885, Taylor order 5 with the last term as error estimate)."""
    from renormalizer.model import Phonon, Mol, HolsteinModel, Op
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, EvolveConfig, EvolveMethod
    nmol, pdim = 4, 4
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    gs = Mps.ground_state(model, max_entangled=False)
    init = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(gs)
    e0 = Quantity(init.expectation(Mpo(model)))
    mpo = Mpo(model, offset=e0)
    init.evolve_config = EvolveConfig(EvolveMethod.prop_and_compress, adaptive=True, guess_dt=8.0)
    occ = [Mpo(model, Op(r"a^\dagger a", dof)) for dof in model.e_dofs]
    out = {}
    _dump_mpo(out, "mpo_", mpo)
    for j, o in enumerate(occ):
        _dump_mpo(out, f"obs{j}_", o)
    out["nobs"] = np.array(len(occ))
    for i, b in enumerate(model.basis):
        out[f"sigmaqn_{i}"] = np.asarray(b.sigmaqn).reshape(b.nbas, -1).astype(np.int64)
    _dump_mps(out, "init_", init)
    mps = init
    vals, guess, norms, bdims = [[mps.expectation(o) for o in occ]], [], [], []
    for step in range(4):
        mps = mps.evolve(mpo, 20.0)
        vals.append([mps.expectation(o) for o in occ])
        guess.append(mps.evolve_config.guess_dt)
        norms.append(mps.mp_norm)
        bdims.append(list(mps.bond_dims))
    out["dt"] = np.array(20.0)
    out["obs_values"] = np.array(vals, dtype=complex).real
    out["guess_dt"] = np.array(guess, dtype=float)
    out["norms"] = np.array(norms)
    out["bond_dims"] = np.array(bdims)
    np.savez_compressed(os.path.join(GOLD, "pc_adaptive_holstein_small.npz"), **out)
    print("pc_adaptive_holstein_small.npz", out["obs_values"][-1], out["guess_dt"], bdims[-1])


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("pc_adaptive",)):
    gen_pc_adaptive()


def _fmo_model(nph):
    import json
    from renormalizer.model import Phonon, Mol, HolsteinModel
    from renormalizer.utils import Quantity
    from renormalizer.utils.constant import cm2au
    sdf = np.array(json.load(open(os.path.join(GOLD, "fmo_sdf.json"))))
    j_cm = np.array([[310, -98, 6, -6, 7, -12, -10, 38], [-98, 230, 30, 7, 2, 12, 5, 8], [6, 30, 0, -59, -2, -10, 5, 2],
                     [-6, 7, -59, 180, -65, -17, -65, -2], [7, 2, -2, -65, 405, 89, -6, 5], [-12, 11, -10, -17, 89, 320, 32, -10],
                     [-10, 5, 5, -64, -6, 32, 270, -11], [38, 8, 2, -2, 5, -10, -11, 505]])
    om_cm = np.linspace(2, 300, nph)
    om = om_cm * cm2au
    hr = np.interp(om_cm, sdf[:, 0], sdf[:, 1])
    hr *= 0.42 / hr.sum()
    phonons = [Phonon.simplest_phonon(Quantity(o), Quantity(l), lam=True) for o, l in zip(om, hr * om)]
    j = j_cm * cm2au
    mols = [Mol(Quantity(e), phonons) for e in np.diag(j)]
    arr = np.array([7, 5, 3, 1, 2, 4, 6]) - 1
    return HolsteinModel(list(np.array(mols)[arr]), j[arr][:, arr])


def gen_fmo():
    """example/fmo.py (7 sites of the FMO complex, Holstein modes from the tabulated spectral density, T = 0) with
    fewer modes per site and a smaller bond dimension: phonon dimensions picked by simplest_phonon, MPO bond
    dimensions, populations after a few TDVP-PS steps."""
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod
    nph = 6
    model = _fmo_model(nph)
    out = {"nph": np.array(nph), "pbond": np.array(model.pbond_list)}
    gs = Mps.ground_state(model, max_entangled=False)
    init = Mpo.onsite(model, r"a^\dagger", dof_set={model.mol_num // 2}).apply(gs)
    mpo = Mpo(model, offset=Quantity(init.expectation(Mpo(model))))
    out["mpo_bond_dims"] = np.array(mpo.bond_dims)
    init.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=12)
    init.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    np.random.seed(4242)
    mps = init.expand_bond_dimension(mpo)
    mps.canonicalise()
    occ = [np.asarray(mps.e_occupations)]
    for _ in range(4):
        mps = mps.evolve(mpo, 160.0)
        occ.append(np.asarray(mps.e_occupations))
    out["e_occ"] = np.array(occ)
    out["bond_dims"] = np.array(mps.bond_dims)
    np.savez_compressed(os.path.join(GOLD, "fmo_small.npz"), **out)
    print("fmo_small.npz pbond", model.pbond_list[:8], "mpo bonds", mpo.bond_dims[:10], "occ", occ[-1])


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("fmo",)):
    gen_fmo()


def gen_thermal():
    """Imaginary-time propagation of the T = infinity one-exciton density operator of the reference's test model
    (mps/tests/test_mpdm.py) to 298 K: energies and occupations after every step, for P&C and fixed-step TDVP-PS."""
    from renormalizer.mps import MpDm, ThermalProp
    from renormalizer.tests import parameter
    from renormalizer.utils import Quantity, EvolveConfig, EvolveMethod, CompressConfig, CompressCriteria
    model = parameter.holstein_model
    beta = Quantity(298, "K").to_beta()
    out = {"beta": np.array(beta), "gs_zpe": np.array(model.gs_zpe)}
    for tag, method, nsteps in (("pc", EvolveMethod.prop_and_compress, 10), ("ps", EvolveMethod.tdvp_ps, 10),
                                ("ps2", EvolveMethod.tdvp_ps2, 10)):
        init = MpDm.max_entangled_ex(model)
        if tag in ("ps", "ps2"):
            init.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=12)
        tp = ThermalProp(init, evolve_config=EvolveConfig(method, adaptive=False, guess_dt=0.1 / 1j),
                         auto_expand=(tag != "ps2"))
        if tag == "ps":
            _dump_mps(out, "ps_init_", tp.latest_mps)      # the expanded D = 12 state the sweeps start from
        tp.evolve(evolve_dt=beta / 2j / nsteps, nsteps=nsteps)
        out[f"{tag}_energies"] = np.array(tp.energies, dtype=complex).real
        out[f"{tag}_e_occ"] = np.array(tp.e_occupations_array)
        out[f"{tag}_ph_occ"] = np.array(tp.ph_occupations_array)
        out[f"{tag}_bond_dims"] = np.array(tp.latest_mps.bond_dims)
        print(tag, out[f"{tag}_e_occ"][-1], out[f"{tag}_energies"][-1], tp.latest_mps.bond_dims)
    np.savez_compressed(os.path.join(GOLD, "thermal_prop_holstein.npz"), **out)


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("thermal",)):
    gen_thermal()


def gen_obs():
    """Observables of a fixed complex MPS: occupations, one-site / electronic reduced density matrices, bond
    singular values and entropies (mps/mps.py:578-609, 1547-1598, 1657-1795)."""
    from renormalizer.model import Phonon, Mol, HolsteinModel
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, CompressConfig, EvolveConfig, EvolveMethod, CompressCriteria
    nmol, pdim = 3, 4
    ph = [Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim),
          Phonon.simple_phonon(Quantity(3.1e-3), Quantity(9.5), 3)]
    model = HolsteinModel([Mol(Quantity(0), ph)] * nmol, Quantity(3.0e-2), 3)
    np.random.seed(2468)
    mps = Mps.random(model, 1, 8)
    mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=8)
    mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    mpo = Mpo(model)
    mps = mps.to_complex().evolve(mpo, 8.0).evolve(mpo, 8.0)
    mps.canonicalise().normalize("mps_only")
    out = {}
    for i, b in enumerate(model.basis):
        out[f"sigmaqn_{i}"] = np.asarray(b.sigmaqn).reshape(b.nbas, -1).astype(np.int64)
    _dump_mps(out, "mps_", mps)
    out["e_occupations"] = np.asarray(mps.e_occupations)
    out["ph_occupations"] = np.asarray(mps.ph_occupations)
    rdm = mps.calc_1site_rdm()
    for k, v in rdm.items():
        out[f"rdm1_{k}"] = np.asarray(v)
    out["edof_rdm"] = np.asarray(mps.calc_edof_rdm())
    out["bond_entropy"] = np.asarray(mps.calc_entropy("bond"))
    s1 = mps.calc_entropy("1site")
    out["site_entropy"] = np.array([s1[k] for k in range(len(mps))])
    rdm2 = mps.calc_2site_rdm()
    for (i, j) in ((0, 1), (0, 4), (2, 3), (3, 8), (7, 8)):
        out[f"rdm2_{i}_{j}"] = np.asarray(rdm2[(i, j)])
    s2 = mps.calc_entropy("2site")
    out["pair_entropy"] = np.array([[i, j, s2[(i, j)]] for (i, j) in sorted(s2)])
    out["mutual_entropy"] = np.asarray(mps.calc_entropy("mutual"))
    sv = mps.calc_bond_singular_values()
    width = max(len(x) for x in sv)
    out["bond_sv"] = np.array([np.pad(np.asarray(x), (0, width - len(x))) for x in sv])
    np.savez_compressed(os.path.join(GOLD, "observables_holstein_small.npz"), **out)
    print("observables_holstein_small.npz", out["e_occupations"], out["bond_entropy"][:3])


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("obs",)):
    gen_obs()
if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("mpo",)):
    gen_mpo()
if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("ps2",)):
    gen_tdvp_ps2()


def gen_pc_rk():
    """Propagate & compress with Runge-Kutta propagators (mps.py:664-792): classical RK4 with a fixed step
    (`prop_and_compress_tdrk4`), the general tableau driver with fixed (Kutta_RK3) and adaptive (RKF45 with its
    embedded 4th-order error estimate) steps, and RK4 under a time-dependent Hamiltonian H(t) = (1 + 0.2 t / dt) H."""
    from renormalizer.model import Phonon, Mol, HolsteinModel, Op
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, EvolveConfig, EvolveMethod, CompressConfig, CompressCriteria
    nmol, pdim = 4, 4
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    gs = Mps.ground_state(model, max_entangled=False)
    init = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(gs)
    e0 = Quantity(init.expectation(Mpo(model)))
    mpo = Mpo(model, offset=e0)
    occ = [Mpo(model, Op(r"a^\dagger a", dof)) for dof in model.e_dofs]
    out = {}
    _dump_mpo(out, "mpo_", mpo)
    for j, o in enumerate(occ):
        _dump_mpo(out, f"obs{j}_", o)
    out["nobs"] = np.array(len(occ))
    for i, b in enumerate(model.basis):
        out[f"sigmaqn_{i}"] = np.asarray(b.sigmaqn).reshape(b.nbas, -1).astype(np.int64)
    _dump_mps(out, "init_", init)
    dt = 15.0
    out["dt"] = np.array(dt)

    def run(tag, cfg, ham, nsteps=3):
        mps = init.copy()
        mps.evolve_config = cfg
        mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=12)
        vals, guess, norms, bdims = [], [], [], []
        for _ in range(nsteps):
            mps = mps.evolve(ham, dt)
            vals.append([mps.expectation(o) for o in occ])
            guess.append(mps.evolve_config.guess_dt)
            norms.append(mps.mp_norm)
            bdims.append(list(mps.bond_dims))
        out[tag + "_obs"] = np.array(vals, dtype=complex).real
        out[tag + "_guess_dt"] = np.array(guess, dtype=float)
        out[tag + "_norms"] = np.array(norms)
        out[tag + "_bond_dims"] = np.array(bdims)
        out[tag + "_energy"] = np.array(mps.expectation(mpo))
        print(tag, out[tag + "_obs"][-1], guess, bdims[-1])

    run("rk4", EvolveConfig(EvolveMethod.prop_and_compress_tdrk4), mpo)
    run("rk3", EvolveConfig(EvolveMethod.prop_and_compress_tdrk, rk_solver="Kutta_RK3"), mpo)
    run("ck45", EvolveConfig(EvolveMethod.prop_and_compress_tdrk, rk_solver="Cash-Karp45", adaptive=True, guess_dt=6.0),
        mpo)

    def mpo_t(t, *args, **kwargs):
        return mpo.scale(1.0 + 0.2 * t / dt)

    run("rk4_td", EvolveConfig(EvolveMethod.prop_and_compress_tdrk4), mpo_t)
    np.savez_compressed(os.path.join(GOLD, "pc_rk_holstein_small.npz"), **out)


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("pc_rk",)):
    gen_pc_rk()


def gen_vmf():
    """TDVP-VMF (mps.py:887-1094) on the reduced headline model: the expanded initial state is stored, then three
    evolve calls each with the matrix-unfolding regularisation (`tdvp_mu_vmf`, auto switch off), with the
    density-matrix regularisation (`tdvp_vmf`, auto switch off), without the overlap corrections
    (`force_ovlp=False`), with the automatic switch, and in imaginary time."""
    from renormalizer.model import Phonon, Mol, HolsteinModel, Op
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, EvolveConfig, EvolveMethod, CompressConfig, CompressCriteria
    nmol, pdim = 3, 4
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim)
    model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    gs = Mps.ground_state(model, max_entangled=False)
    init = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(gs)
    mpo = Mpo(model, offset=Quantity(init.expectation(Mpo(model))))
    init.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=6)
    init.evolve_config = EvolveConfig(EvolveMethod.tdvp_mu_vmf)
    init = init.expand_bond_dimension(mpo)
    init.canonicalise()
    occ = [Mpo(model, Op(r"a^\dagger a", dof)) for dof in model.e_dofs]
    out = {}
    _dump_mps(out, "init_", init)
    dt = 20.0
    out["dt"] = np.array(dt)

    def run(tag, method, step, auto=False, force_ovlp=True):
        mps = init.copy()
        mps.evolve_config = EvolveConfig(method, force_ovlp=force_ovlp)
        mps.evolve_config.vmf_auto_switch = auto
        vals, norms, methods = [], [], []
        for _ in range(3):
            mps = mps.evolve(mpo, step)
            vals.append([mps.expectation(o) for o in occ])
            norms.append(mps.mp_norm)
            methods.append(mps.evolve_config.method.name)
        out[tag + "_obs"] = np.array(vals, dtype=complex).real
        out[tag + "_norms"] = np.array(norms)
        out[tag + "_energy"] = np.array(complex(mps.expectation(mpo)).real)
        out[tag + "_methods"] = np.array(methods)
        print(tag, out[tag + "_obs"][-1], norms[-1], methods)

    run("mu", EvolveMethod.tdvp_mu_vmf, dt)
    run("vmf", EvolveMethod.tdvp_vmf, dt)
    run("mu_noovlp", EvolveMethod.tdvp_mu_vmf, dt, force_ovlp=False)
    run("auto", EvolveMethod.tdvp_mu_vmf, dt, auto=True)
    run("imag", EvolveMethod.tdvp_mu_vmf, -20.0j)

    # constant mean field (mps.py:1096-1265) from the same state; short steps: with the padded singular values the
    # per-site equations are stiff and the explicit RK45 of the reference needs ~1e6 steps at dt = 5
    def run_cmf(tag, midpoint=True, trapz=False, solver="krylov", step=0.5):
        mps = init.copy()
        mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_mu_cmf, ivp_solver=solver)
        mps.evolve_config.tdvp_cmf_midpoint = midpoint
        mps.evolve_config.tdvp_cmf_c_trapz = trapz
        vals, norms = [], []
        for _ in range(3):
            mps = mps.evolve(mpo, step)
            vals.append([mps.expectation(o) for o in occ])
            norms.append(mps.mp_norm)
        out[tag + "_obs"] = np.array(vals, dtype=complex).real
        out[tag + "_norms"] = np.array(norms)
        out[tag + "_energy"] = np.array(complex(mps.expectation(mpo)).real)
        print(tag, out[tag + "_obs"][-1], norms[-1])

    run_cmf("cmf")
    run_cmf("cmf_trapz", trapz=True)
    run_cmf("cmf_first", midpoint=False)
    run_cmf("cmf_rk", solver="RK45")
    run_cmf("cmf_imag", step=-0.5j)
    np.savez_compressed(os.path.join(GOLD, "tdvp_vmf_holstein_small.npz"), **out)


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("vmf",)):
    gen_vmf()


def gen_ofs():
    """On-the-fly swapping (mp.py:696-757, mpo.py:427-454) with two-site TDVP on the reduced headline Hamiltonian
    written as a general `Model` (the Holstein class refuses OFS), from a stored random one-exciton state of bond
    dimension 5 - on product states the two site orders tie at zero entropy and rounding noise decides: site order
    after every step, populations, energies."""
    from renormalizer.model import Phonon, Mol, HolsteinModel, Model, Op
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, EvolveConfig, EvolveMethod, CompressConfig, CompressCriteria, OFS
    nmol, pdim = 4, 4
    ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), pdim)
    hol = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
    out = {}
    np.random.seed(7)
    model0 = Model(hol.basis, hol.ham_terms)
    init = Mps.random(model0, 1, 5, percent=1.0)
    init.canonicalise()
    init.normalize("mps_and_coeff")
    _dump_mps(out, "init_", init)
    for tag, ofs in (("s", OFS.ofs_s), ("d", OFS.ofs_d), ("ds", OFS.ofs_ds)):
        model = Model(hol.basis, hol.ham_terms)
        mpo = Mpo(model)
        mps = init.copy()
        mps.model = model
        mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps2)
        mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=5, ofs=ofs)
        orders, vals, energies = [], [], []
        for _ in range(4):
            mps = mps.evolve(mpo, 20.0)
            orders.append([str(b.dofs[0]) for b in mps.model.basis])
            vals.append([mps.expectation(Mpo(mps.model, Op(r"a^\dagger a", dof))) for dof in hol.e_dofs])
            energies.append(mps.expectation(mpo))
        out[f"tdvp_{tag}_orders"] = np.array(orders)
        out[f"tdvp_{tag}_obs"] = np.array(vals, dtype=complex).real
        out[f"tdvp_{tag}_energies"] = np.array(energies, dtype=complex).real
        out[f"tdvp_{tag}_bond_dims"] = np.array(mps.bond_dims)
        out[f"tdvp_{tag}_mpo_bond_dims"] = np.array(mpo.bond_dims)
        print(tag, orders, out[f"tdvp_{tag}_obs"][-1], mpo.bond_dims)
    np.savez_compressed(os.path.join(GOLD, "ofs_holstein_small.npz"), **out)


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("ofs",)):
    gen_ofs()


def gen_thermofield():
    """Finite temperature through thermofield-doubled baths (BASELINE config 4): the Hamiltonian is written term by
    term with the reference's Model / Op exactly like transport/tests/test_spectral_function.py:16-48 (physical mode
    omega b^+b, tilde mode -omega b~^+b~, couplings -g omega cosh(theta) / sinh(theta) a^+a (b^+ + b),
    theta = arctanh(exp(-beta omega / 2)), on-site energy eps + sum g^2 omega), then propagated with the reference's
    TDVP-PS.  Two cases: a Holstein trimer with two modes per molecule, and the FMO model of example/fmo.py with
    three modes per site at 77 K (35 sites).  Stored: parameters, dense Hamiltonian (trimer, small ladders), bond
    dimensions, populations / energy per step."""
    import json
    from renormalizer.model import Op, Model
    from renormalizer.model.basis import BasisSimpleElectron, BasisSHO
    from renormalizer.mps import Mps, Mpo
    from renormalizer.utils import Quantity, CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod
    from renormalizer.utils.constant import cm2au
    from renormalizer.model import Phonon

    out = {}

    def tf_model(eps, jmat, omegas, gs, nlevels, beta):
        """eps[i], jmat[i, j], per molecule the modes (omega_k, g_k) with nlevels[k] levels (physical and tilde)"""
        nmol = len(eps)
        basis, ham = [], []
        for i in range(nmol):
            basis.append(BasisSimpleElectron(f"e{i}"))
            for k, (w, nl) in enumerate(zip(omegas, nlevels)):
                basis.append(BasisSHO(f"v{i}_{k}", w, int(nl)))
                basis.append(BasisSHO(f"t{i}_{k}", w, int(nl)))
        for i in range(nmol):
            ham.append(Op(r"a^\dagger a", f"e{i}", eps[i] + sum(g ** 2 * w for w, g in zip(omegas, gs))))
            for j in range(nmol):
                if i != j and jmat[i, j] != 0:
                    ham.append(Op(r"a^\dagger a", [f"e{i}", f"e{j}"], jmat[i, j]))
            for k, (w, g) in enumerate(zip(omegas, gs)):
                theta = np.arctanh(np.exp(-beta * w / 2))
                ham.append(Op(r"b^\dagger b", f"v{i}_{k}", w))
                ham.append(Op(r"b^\dagger b", f"t{i}_{k}", -w))
                ham.append(-g * np.cosh(theta) * w * Op(r"a^\dagger a", f"e{i}") * Op(r"b^\dagger + b", f"v{i}_{k}"))
                ham.append(-g * np.sinh(theta) * w * Op(r"a^\dagger a", f"e{i}") * Op(r"b^\dagger + b", f"t{i}_{k}"))
        return Model(basis, ham)

    def run(model, nmol, start, D, nsteps, dt, seed, pre):
        gs = Mps.ground_state(model, max_entangled=False)
        init = Mpo(model, Op(r"a^\dagger", f"e{start}")).apply(gs)
        e0 = init.expectation(Mpo(model))
        mpo = Mpo(model, offset=Quantity(e0))
        occ_ops = [Mpo(model, Op(r"a^\dagger a", f"e{i}")) for i in range(nmol)]
        init.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
        init.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
        np.random.seed(seed)
        mps = init.expand_bond_dimension(mpo)
        mps.canonicalise()
        # fixed-bond TDVP follows the 1e-10 padding of expand_bond_dimension, which no two codes reproduce beyond
        # ~1e-6 relative: the expanded state itself is part of the fixture
        _dump_mps(out, pre + "init_", mps)
        occ = [np.array([mps.expectation(o) for o in occ_ops])]
        ener = [mps.expectation(mpo)]
        for _ in range(nsteps):
            mps = mps.evolve(mpo, dt)
            occ.append(np.array([mps.expectation(o) for o in occ_ops]))
            ener.append(mps.expectation(mpo))
        return e0, np.array(mpo.bond_dims), np.array(mps.bond_dims), np.array(occ), np.array(ener)

    # ---- Holstein trimer, two modes, k T ~ omega_0
    omegas = np.array([0.005, 0.012])
    gs_ = np.array([0.9, 0.4])
    temperature = Quantity(1500, "K")
    beta = temperature.to_beta()
    eps = np.array([0.0, 0.002, -0.001])
    jmat = np.array([[0, 0.003, 0.001], [0.003, 0, 0.002], [0.001, 0.002, 0]])
    out.update(tri_omega=omegas, tri_g=gs_, tri_beta=np.array(beta), tri_eps=eps, tri_j=jmat)
    small = tf_model(eps[:2], jmat[:2, :2], omegas[:1], gs_[:1], [3], beta)
    out["dimer_dense"] = Mpo(small).todense()
    model = tf_model(eps, jmat, omegas, gs_, [8, 5], beta)
    e0, wb, mb, occ, ener = run(model, 3, 0, 16, 6, 40.0, 77, "tri_")
    out.update(tri_levels=np.array([8, 5]), tri_e0=np.array(e0), tri_mpo_bond=wb, tri_bond=mb, tri_occ=occ,
               tri_energy=ener, tri_dt=np.array(40.0))
    print("trimer occ", occ[-1], "mpo bonds", wb)
    # ---- FMO at 77 K, three modes per site (example/fmo.py parameters)
    nph = 3
    sdf = np.array(json.load(open(os.path.join(GOLD, "fmo_sdf.json"))))
    j_cm = np.array([[310, -98, 6, -6, 7, -12, -10, 38], [-98, 230, 30, 7, 2, 12, 5, 8], [6, 30, 0, -59, -2, -10, 5, 2],
                     [-6, 7, -59, 180, -65, -17, -65, -2], [7, 2, -2, -65, 405, 89, -6, 5], [-12, 11, -10, -17, 89, 320, 32, -10],
                     [-10, 5, 5, -64, -6, 32, 270, -11], [38, 8, 2, -2, 5, -10, -11, 505]], dtype=float)
    om_cm = np.linspace(2, 300, nph)
    om = om_cm * cm2au
    hr = np.interp(om_cm, sdf[:, 0], sdf[:, 1])
    hr *= 0.42 / hr.sum()
    phonons = [Phonon.simplest_phonon(Quantity(o), Quantity(l), lam=True) for o, l in zip(om, hr * om)]
    levels = [ph.n_phys_dim for ph in phonons]
    g = np.sqrt(hr)                                       # Huang-Rhys factor S = g^2
    arr = np.array([7, 5, 3, 1, 2, 4, 6]) - 1
    j = (j_cm * cm2au)[arr][:, arr]
    beta77 = Quantity(77, "K").to_beta()
    model = tf_model(np.diag(j).copy(), j - np.diag(np.diag(j)), om, g, levels, beta77)
    e0, wb, mb, occ, ener = run(model, 7, 3, 12, 4, 160.0, 78, "fmo_")
    out.update(fmo_nph=np.array(nph), fmo_levels=np.array(levels), fmo_beta=np.array(beta77), fmo_e0=np.array(e0),
               fmo_mpo_bond=wb, fmo_bond=mb, fmo_occ=occ, fmo_energy=ener)
    print("fmo levels", levels, "occ", occ[-1], "mpo bonds", wb[:8])
    np.savez_compressed(os.path.join(GOLD, "thermofield.npz"), **out)


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("thermofield",)):
    gen_thermofield()


def gen_dmrg():
    """Seams and runs of the DMRG path:
      * contract_one_site_multi_mpo (mps/lib.py:121-166) and the two-layer hop_expr (mps/hop_expr.py:24-52) on random
        inputs;
      * eigh_iterative / eigh_direct (mps/gs.py:383-407, 486-576) captured inside real sweeps of the Holstein test
        model (one-layer and (H - omega)^2): (ltensor, rtensor, cmo, qn_mask, cguess) -> (e, c);
      * H2O / STO-3G (example/h2o_qc.py): a seeded random start at M = 50, the energy after every sweep for the
        M = 50 and the M = 512 procedure (same start; the bonds saturate below 50 - SURVEY 8d item 5), final bonds.
    -> tests/golden/dmrg_seams.npz, tests/golden/h2o_dmrg.npz"""
    from renormalizer.mps import lib as ref_lib, gs as ref_gs
    from renormalizer.mps.hop_expr import hop_expr
    from renormalizer.mps import Mps, Mpo
    from renormalizer.model import Model, Phonon, Mol, HolsteinModel, h_qc
    from renormalizer.utils import Quantity, constant
    rng = np.random.default_rng(20260929)
    out = {}
    # ---- multi-MPO environment update, two MPO layers
    k = 0
    for cplx in (False, True):
        for anc in (False, True):
            for dom in ("L", "R"):
                Dl, Dr, d, da, w1l, w1r, w2l, w2r = 4, 5, 3, 2, 3, 2, 2, 4
                shp = (Dl, d, da, Dr) if anc else (Dl, d, Dr)
                ms = _rand(rng, shp, cplx)
                mo1 = _rand(rng, (w1l, d, d, w1r), False)
                mo2 = _rand(rng, (w2l, d, d, w2r), False)
                env = _rand(rng, (Dl, w1l, w2l, Dl) if dom == "L" else (Dr, w1r, w2r, Dr), cplx)
                res = ref_lib.contract_one_site_multi_mpo(env, ms, [mo1, mo2], dom)
                out.update({f"menv{k}_env": env, f"menv{k}_ms": ms, f"menv{k}_mo1": mo1, f"menv{k}_mo2": mo2,
                            f"menv{k}_dom": np.array(dom), f"menv{k}_out": np.asarray(res)})
                k += 1
    out["menv_n"] = np.array(k)
    # ---- two-layer effective Hamiltonian
    k = 0
    for cplx in (False, True):
        for nsite in (1, 2):
            Dl, Dr, d0, d1, wl, wm, wr = 4, 5, 3, 2, 3, 2, 4
            l = _rand(rng, (Dl, wl, wl, Dl), cplx)
            if nsite == 1:
                cmo = [_rand(rng, (wl, d0, d0, wr), False)]
                cshape = (Dl, d0, Dr)
            else:
                cmo = [_rand(rng, (wl, d0, d0, wm), False), _rand(rng, (wm, d1, d1, wr), False)]
                cshape = (Dl, d0, d1, Dr)
            r = _rand(rng, (Dr, wr, wr, Dr), cplx)
            c = _rand(rng, cshape, cplx)
            hc = hop_expr(l, r, [m.copy() for m in cmo], cshape, True)(c)
            out.update({f"hop2_{k}_l": l, f"hop2_{k}_r": r, f"hop2_{k}_c": c, f"hop2_{k}_out": np.asarray(hc),
                        f"hop2_{k}_nsite": np.array(nsite)})
            for i, m in enumerate(cmo):
                out[f"hop2_{k}_w{i}"] = m
            k += 1
    out["hop2_n"] = np.array(k)

    # ---- eigensolver seams inside real sweeps (renormalizer/tests/parameter.py model)
    omega = [Quantity(106.51, "cm^{-1}"), Quantity(1555.55, "cm^{-1}")]
    dis = [Quantity(30.1370), Quantity(8.7729)]
    ph_list = [Phonon.simple_phonon(o, d, 4) for o, d in zip(omega, dis)]
    j = np.array([[0.0, -0.1, -0.2], [-0.1, 0.0, -0.3], [-0.2, -0.3, 0.0]]) / constant.au2ev
    model = HolsteinModel([Mol(Quantity(2.67, "eV"), ph_list, 15.45)] * 3, j, 3)
    mpo = Mpo(model)
    records = []
    orig_it, orig_di = ref_gs.eigh_iterative, ref_gs.eigh_direct

    def rec(kind, orig, wanted):
        def wrapped(mps, qn_mask, ltensor, rtensor, cmo, omega_, *rest):
            e, c = orig(mps, qn_mask, ltensor, rtensor, cmo, omega_, *rest)
            tag = (kind, omega_ is not None, mps.optimize_config.method)
            if wanted.get(tag, 0) > 0 and np.sum(qn_mask) > 8:
                wanted[tag] -= 1
                records.append(dict(kind=kind, twolayer=omega_ is not None, l=np.asarray(ltensor), r=np.asarray(rtensor),
                                    cmo=[np.asarray(m) for m in cmo], mask=qn_mask.copy(),
                                    guess=None if not rest else np.asarray(rest[0][0]), e=float(e), c=np.asarray(c)))
            return e, c
        return wrapped

    wanted = {("it", False, "2site"): 2, ("it", False, "1site"): 1, ("di", False, "2site"): 1, ("di", False, "1site"): 1,
              ("it", True, "2site"): 1, ("it", True, "1site"): 1, ("di", True, "1site"): 1}
    ref_gs.eigh_iterative = rec("it", orig_it, wanted)
    ref_gs.eigh_direct = rec("di", orig_di, wanted)
    try:
        for method in ("2site", "1site"):
            for om in (None, 0.09):
                np.random.seed(2019)
                mps = Mps.random(model, 1, 12, percent=1.0)
                mps.optimize_config.procedure = [[12, 0.4], [16, 0.2], [20, 0]]
                mps.optimize_config.method = method
                energies, _ = ref_gs.optimize_mps(mps.copy(), mpo, omega=om)
                out[f"holstein_{method}_{'omega' if om else 'gs'}_energies"] = np.array(energies)
                print(method, om, energies)
    finally:
        ref_gs.eigh_iterative, ref_gs.eigh_direct = orig_it, orig_di
    for k, r in enumerate(records):
        pre = f"eig{k}_"
        out.update({pre + "kind": np.array(r["kind"]), pre + "twolayer": np.array(r["twolayer"]), pre + "l": r["l"],
                    pre + "r": r["r"], pre + "mask": r["mask"], pre + "e": np.array(r["e"]), pre + "c": r["c"],
                    pre + "nsite": np.array(len(r["cmo"]))})
        for i, m in enumerate(r["cmo"]):
            out[pre + f"w{i}"] = m
        if r["guess"] is not None:
            out[pre + "guess"] = r["guess"]
    out["eig_n"] = np.array(len(records))
    out["eig_omega"] = np.array(0.09)
    print("eigensolver seams:", [(r["kind"], r["twolayer"], len(r["cmo"]), r["mask"].shape, r["e"]) for r in records])
    np.savez_compressed(os.path.join(GOLD, "dmrg_seams.npz"), **out)

    # ---- H2O
    sh, aseri, nuc = h_qc.read_fcidump(os.path.join(GOLD, "h2o_fcidump.txt"), 7)
    basis, terms = h_qc.qc_model(sh, aseri)
    model = Model(basis, terms)
    mpo = Mpo(model)
    np.random.seed(512)
    start = Mps.random(model, [5, 5], 50, percent=1.0)
    h2o = {"nuc": np.array(nuc), "mpo_bond_dims": np.array(mpo.bond_dims)}
    _dump_mps(h2o, "init_", start)
    for M in (50, 512):
        mps = start.copy()
        mps.optimize_config.procedure = [[M, 0.4], [M, 0.2], [M, 0.1], [M, 0], [M, 0], [M, 0], [M, 0]]
        mps.optimize_config.method = "2site"
        energies, gs = ref_gs.optimize_mps(mps, mpo)
        h2o[f"energies_M{M}"] = np.array(energies)
        h2o[f"bond_dims_M{M}"] = np.array(gs.bond_dims)
        h2o[f"sweep_bond_dims_M{M}"] = np.array(mps.bond_dims)
        print("H2O M", M, energies, gs.bond_dims, mps.bond_dims)
    # The sweeps with percent > 0 fill block quotas with null-space vectors that svd_qn draws from numpy's global
    # generator (mps/svd_qn.py:52-63): their energies are one realisation.  Three more, to record the spread.
    spread = []
    for seed in (1, 2, 3):
        np.random.seed(seed)
        mps = start.copy()
        mps.optimize_config.procedure = [[50, 0.4], [50, 0.2], [50, 0.1], [50, 0], [50, 0], [50, 0], [50, 0]]
        mps.optimize_config.method = "2site"
        spread.append(ref_gs.optimize_mps(mps, mpo)[0])
    h2o["energies_M50_reseeded"] = np.array(spread)
    print("reseeded", np.array(spread))
    np.savez_compressed(os.path.join(GOLD, "h2o_dmrg.npz"), **h2o)


if __name__ == "__main__" and (len(sys.argv) > 1 and sys.argv[1] in ("dmrg",)):
    gen_dmrg()
