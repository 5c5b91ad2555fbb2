"""BASELINE headline size (50 sites, dphys 2/16, Dbond 256, complex128) pinned to the oracle, where every
performance path of the engine is engaged: quantum-number centre masks (only at (256, 16, 256)), carried
environments, host-predicted unit channels, beta-source C-step, asynchronous Lanczos, the blocked QR at 4096 rows.

  * one full ``Mps.evolve`` on the device against ``oracle.tdvp_ps_step`` from the same ``to_arrays()`` state
    (reference order of operations: mps/mps.py:1267-1404);
  * A/B of every engine switch that selects an alternative implementation (separate processes: the library reads its
    switches once), bitwise where the design claims bit-identity, <= 1e-10 on observables otherwise;
  * the actual bond dimensions of the benchmark state.
pytest -m gpu; the oracle leg needs about a minute of host time."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import mps_oracle as orc

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sorted_rows(a):
    a = np.asarray(a).reshape(len(a), -1)
    return a[np.lexsort(a.T[::-1])]


@pytest.fixture(scope="module")
def headline(tmp_path_factory):
    """The benchmark's start state (device-side preparation, ~6 s), evolved once so that it is a generic complex
    state with filled bonds, written to a checkpoint the A/B processes start from."""
    import bench
    path = str(tmp_path_factory.mktemp("headline") / "state.npz")
    model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical")
    mps = mps.evolve(mpo, 10.0)
    mps.dump(path)
    return model, mpo, mps, path


def test_headline_bond_dims(headline):
    """Dbond = 256 is reached on every interior bond of the benchmark state (the edges are capped by the product of
    the physical dimensions and the 1-exciton sector)."""
    _, _, mps, _ = headline
    dims = list(mps.bond_dims)
    assert len(dims) == 51 and dims[0] == dims[-1] == 1
    assert all(d == 256 for d in dims[4:-4]), dims
    assert all(len(q) == d for q, d in zip(mps.qn, dims))


def _solve_sites(nsite, to_right):
    """(centre site, neighbour the bond factor goes to or None) of every local solve of one TDVP-PS evolve, in the order
    the driver and the oracle run them: per half sweep and site a forward solve of the site, then - unless it is the last
    site of the half sweep - a backward solve of the bond factor between the site and its neighbour."""
    out = []
    for _ in range(2):
        order = list(range(nsite)) if to_right else list(range(nsite - 1, -1, -1))
        for k, i in enumerate(order):
            out.append((i, None))
            if k != nsite - 1:
                out.append((i, i + 1 if to_right else i - 1))
        to_right = not to_right
    return out


class _MarginalSolves:
    """Spy on the oracle's local solves (orc.hop_apply / orc.expm_krylov) during one ``tdvp_ps_step``: every solve is
    listed with its Krylov dimension and the margins of its stopping tests; a solve that one of those tests decided within
    a factor 3 of its threshold keeps its INPUTS (L, W, R, start vector, dt) and its result."""
    BAND = 3.0

    def __init__(self):
        self.rec, self._last = [], {}

    def __enter__(self):
        self._hop, self._expm = orc.hop_apply, orc.expm_krylov
        spy = self

        def hop(l, r, cmo, c):
            spy._last = dict(l=l, r=r, cmo=cmo, shape=c.shape)
            return spy._hop(l, r, cmo, c)

        def expm(Afunc, dt, vstart, margins=None, **kw):
            m = [] if margins is None else margins
            n0 = len(m)
            out, k = spy._expm(Afunc, dt, vstart, margins=m, **kw)
            mine = list(m[n0:])
            e = dict(k=k, margins=mine)
            if any(1 / spy.BAND <= x <= spy.BAND for x in mine):
                L = spy._last
                e.update(l=L["l"].copy(), r=L["r"].copy(), cmo=[w.copy() for w in L["cmo"]], shape=L["shape"],
                         v=np.array(vstart).copy(), dt=dt, out=out.copy())
            spy.rec.append(e)
            return out, k

        orc.hop_apply, orc.expm_krylov = hop, expm
        return self

    def __exit__(self, *exc):
        orc.hop_apply, orc.expm_krylov = self._hop, self._expm
        return False


def _oracle_step(ost, w_host):
    from conftest import oracle_threads
    with _MarginalSolves() as spy, oracle_threads():
        nxt = orc.tdvp_ps_step(ost, w_host, 10.0)
    return nxt, spy.rec


def _oracle_local_expectations(sites, mpos):
    """<psi|O_k|psi> of the oracle's state for operators that act on ONE site each (the electronic
    occupations): one pass of norm environments from either end and a local contraction per operator - the same numbers as
    ``orc.expectation`` (checked below for one of them) in a second instead of a minute at D = 256."""
    n = len(sites)
    left = [np.ones((1, 1))]
    for a in sites[:-1]:                                   # left[i + 1][b, b'] = sum conj(A)[a, p, b] left[i][a, a'] A[a', p, b']
        t = np.tensordot(left[-1], a, axes=([1], [0]))
        left.append(np.tensordot(a.conj(), t, axes=([0, 1], [0, 1])))
    right = [np.ones((1, 1))]
    for a in reversed(sites[1:]):                          # right[b, b'] <- sum conj(A)[a, p, b] A[a', p, b'] right
        t = np.tensordot(a, right[0], axes=([2], [1]))
        right.insert(0, np.tensordot(a.conj(), t, axes=([1, 2], [1, 2])))
    out = []
    for m in mpos:
        ws = [m[i] for i in range(len(m))]
        acting = [i for i, w in enumerate(ws)
                  if not (w.shape[0] == w.shape[3] == 1 and np.array_equal(w[0, :, :, 0], np.eye(w.shape[1])))]
        if len(acting) != 1 or any(w.shape[0] != 1 or w.shape[3] != 1 for w in ws):
            out.append(orc.expectation(sites, ws).real)
            continue
        i = acting[0]
        a, o = sites[i], ws[i][0, :, :, 0]                 # o[p_up, p_down]: <bra p_up| O |ket p_down>
        t = np.tensordot(left[i], a, axes=([1], [0]))      # [a, p', b']
        t = np.tensordot(t, right[i], axes=([2], [1]))     # [a, p', b]
        t = np.tensordot(o, t, axes=([1], [1]))            # [p, a, b]
        out.append(np.tensordot(a.conj(), t, axes=([0, 1, 2], [1, 0, 2])).real)
    return np.array(out)


def _compare_evolve(model, mpo, dev, ost, solves):
    """Device evolve (result ``dev``) against the oracle's evolve of the same physical state (result ``ost``, its local
    solves ``solves`` as recorded by ``_MarginalSolves``)."""
    from test_engine_gpu import dev_expm
    from renormalizer_amd.engine import get_engine
    from conftest import oracle_threads
    w_host = [mpo[i] for i in range(len(mpo))]
    occ_dev = np.asarray(dev.e_occupations)
    occ_mpos = model.mpos["e_occupations"]
    with oracle_threads():
        occ_orc = _oracle_local_expectations(ost.sites, occ_mpos)
        k = len(occ_mpos) // 2                              # (the fast path against the oracle's own routine, one operator)
        assert abs(occ_orc[k] - orc.expectation(ost.sites, [occ_mpos[k][i] for i in range(len(occ_mpos[k]))]).real) < 1e-12
        e_orc = orc.expectation(ost.sites, w_host)
    assert np.abs(occ_dev - occ_orc).max() < 1e-8, np.abs(occ_dev - occ_orc).max()
    e_dev = dev.expectation(mpo)
    assert abs(e_dev - e_orc) < 1e-8
    assert abs(dev.mp_norm - 1.0) < 1e-12
    # integer bookkeeping: bit exact
    assert list(dev.bond_dims) == list(ost.bond_dims)
    assert dev.qnidx == ost.qnidx and dev.to_right == ost.to_right
    for a, b in zip(dev.qn, ost.qn):
        assert np.array_equal(_sorted_rows(a), _sorted_rows(b))
    st = dev.evolve_config.stat
    assert st["nobs"] == len(ost.krylov_dims) == len(solves) == 2 * (2 * len(dev) - 1)
    # Solve by solve: the same Krylov dimension, except where the oracle's own stopping test was marginal.  The test is
    # np.allclose on the ELEMENTS of the local tensor (largest |res - new_res| / (atol + rtol |new_res|) <= 1), and the
    # elements depend on the gauge of the bond bases, which the two codes do not share once their QR factorisations have
    # completed the poorly determined directions of a bond differently (LAPACK's Householder, the device's Householder
    # kernels and its Cholesky-QR: profiles/r05_qr_gauge.md).  A solve whose oracle ratio lies within a factor 3 of 1 may
    # therefore differ by one check (2 vectors); every other solve has to agree exactly.  (The factor was 2 while the
    # differing solves seen had oracle ratios 0.46 ... 2.2; the third evolve - each side continuing from its own gauge -
    # has since shown 0.38 and 0.44.  The band is an observation about gauges, not the statement that carries the weight:
    # EVERY solve inside it is repeated below on the oracle's own inputs and has to reproduce the oracle's dimension.)
    dev_dims, orc_dims = list(st["steps"]), list(ost.krylov_dims)
    assert [e["k"] for e in solves] == orc_dims
    marginal = [any(1 / 3 <= m <= 3.0 for m in e["margins"]) for e in solves]
    differ = [i for i, (a, b) in enumerate(zip(dev_dims, orc_dims)) if a != b]
    assert all(marginal[i] and abs(dev_dims[i] - orc_dims[i]) <= 2 for i in differ), \
        [(i, dev_dims[i], orc_dims[i], solves[i]["margins"]) for i in differ
         if not (marginal[i] and abs(dev_dims[i] - orc_dims[i]) <= 2)]
    # measured on consecutive evolves of this state (profiles/r06_krylov_margin_probe.md): 2, 7, 7 and 12 of 198, all of
    # them among the 30 - 35 marginal solves of the evolve; the bound is a tenth of the solves
    assert len(differ) <= 20, differ
    # Round 6 makes "it is the gauge, not the solver" a checked statement instead of an argument: EVERY marginal solve
    # of the oracle's evolve (those within a factor 3: about 45 of 198 on this state, among them 1.00 and 1.01) is solved again by
    # the device's Lanczos exponential ON THE ORACLE'S INPUTS - same gauge, same numbers - and must stop at exactly the
    # oracle's dimension with the same result.  An error in the device's residual estimate of a few per cent would move
    # one of them; a wrong dimension in the evolve above can then only come from different (equivalent) inputs.
    eng = get_engine()
    checked = 0
    for i, e in enumerate(solves):
        if "l" not in e:
            continue
        out_d, k_d = dev_expm(eng, e["l"], e["r"], e["cmo"], e["v"].reshape(e["shape"]), e["dt"])
        assert k_d == e["k"], (i, k_d, e["k"], e["margins"])
        assert np.abs(out_d.ravel() - e["out"]).max() < 1e-12, (i, np.abs(out_d.ravel() - e["out"]).max())
        checked += 1
    assert checked >= len(differ)
    # (what the two bounds above allow: 20 solves, two vectors each - the differing solves of an evolve all lean the same
    # way, the device's completion of the padded directions makes the element-wise test a little stricter: 12 of 198 in
    # the third evolve are 0.12)
    assert abs(st["mean"] - float(np.mean(orc_dims))) <= 2 * 20 / len(orc_dims) + 1e-9
    ov = orc.mps_dot([s.conj() for s in ost.sites], dev.to_arrays())
    assert abs(abs(ov) - 1.0) < 1e-9, abs(ov)
    return len(differ), checked


def _oracle_state(model, mps):
    return orc.MpsState(mps.to_arrays(), [q.copy() for q in mps.qn], mps.qnidx, mps.qntot.copy(), mps.to_right,
                        [np.array(b.sigmaqn) for b in model.basis], complex(mps.coeff))


@pytest.fixture(scope="module")
def first_evolve(headline):
    """One evolve of the benchmark state on the device and in the oracle from the same tensors (~65 s of host time)."""
    model, mpo, mps, _ = headline
    ost0 = _oracle_state(model, mps)
    dev = mps.evolve(mpo, 10.0)
    ost, solves = _oracle_step(ost0, [mpo[i] for i in range(len(mpo))])
    return dev, ost, solves


def test_headline_one_evolve_vs_oracle(headline, first_evolve):
    """One evolve at the headline size on the device and in the oracle from the same tensors: electronic
    occupations and <H> within 1e-8 (north_star: 1e-6 relative), integer bookkeeping exact, same number of Krylov
    solves, |<psi_oracle|psi_device>| = 1; every marginal solve repeated by the device on the oracle's inputs."""
    model, mpo, mps, _ = headline
    dev, ost, solves = first_evolve
    assert abs(dev.expectation(mpo) - mps.expectation(mpo)) < 1e-6
    ndiff, checked = _compare_evolve(model, mpo, dev, ost, solves)
    assert checked >= 10, checked          # (the state does have marginal solves: the check above is not vacuous)


def test_headline_three_evolves_vs_oracle(headline, first_evolve):
    """Evolves two and three against the ORACLE as well (round-5 verdict: they were pinned only against a file this
    engine wrote).  Each side continues from its own state - same physics, different gauge - so the optimistic repeat of
    the first evolve, the noted sites of the second and third and the carried environments are all on the device's path.
    Every evolve: occupations and <H> to 1e-8, bookkeeping exact, Krylov dimensions solve by solve (marginal rule, the
    marginal solves repeated on the oracle's inputs), overlap 1 to 1e-9."""
    model, mpo, mps, _ = headline
    dev, ost, _ = first_evolve
    w_host = [mpo[i] for i in range(len(mpo))]
    for _ in range(2):
        dev = dev.evolve(mpo, 10.0)
        ost, solves = _oracle_step(ost, w_host)
        _compare_evolve(model, mpo, dev, ost, solves)


def test_headline_qr_schemes_agree(headline):
    """The same evolve with the Householder and with the Cholesky-QR block QR (mpse_block_qr_scheme 0 / 1): the same
    Krylov dimension in every solve, the same state (overlap 1 to 1e-12), occupations and <H> to 1e-12 - and the
    Cholesky-QR kernels did run in the second one."""
    from renormalizer_amd.engine import get_engine
    _, mpo, mps, _ = headline
    eng = get_engine()
    out = []
    try:
        for scheme in (0, 1):
            eng.block_qr_scheme(scheme)
            s0 = eng.block_qr_stats()
            ev = mps.evolve(mpo, 10.0)
            s1 = eng.block_qr_stats()
            out.append((list(ev.evolve_config.stat["steps"]), ev.to_arrays(), np.asarray(ev.e_occupations),
                        ev.expectation(mpo), s1[1] - s0[1]))
    finally:
        eng.block_qr_scheme(-1)
    (d0, a0, o0, e0, c0), (d1, a1, o1, e1, c1) = out
    assert c0 == 0 and c1 >= 40, (c0, c1)
    assert d0 == d1, [(i, x, y) for i, (x, y) in enumerate(zip(d0, d1)) if x != y]
    assert np.abs(o0 - o1).max() < 1e-12 and abs(e0 - e1) < 1e-12
    ov = orc.mps_dot([x.conj() for x in a0], a1)
    assert abs(abs(ov) - 1.0) < 1e-12, abs(ov)


def test_headline_five_evolves_conserve(headline):
    """Five consecutive evolves at the headline size (the bench runs twenty): <H> drifts by less than 1e-6 relative to
    the band width, the electronic populations sum to 1 to 1e-9 at every step, the norm stays 1, and the bond
    dimensions stay where the fixed-bond scheme keeps them."""
    model, mpo, mps, _ = headline
    e0 = mps.expectation(mpo)
    dims0 = list(mps.bond_dims)
    # the prepared benchmark state is pinned (tests/golden/headline_state_pin.npz, written by
    # tools/make_headline_state_pin.py from this engine): a change in expand_bond_dimension / the block SVD that makes
    # the local problems easier or harder moves bench.py's figure - it has to fail here first
    pin = np.load(os.path.join(REPO, "tests", "golden", "headline_state_pin.npz"))
    assert abs(mps.evolve_config.stat["mean"] - float(pin["mean_krylov"][0])) < 0.15, mps.evolve_config.stat["mean"]
    assert np.abs(np.asarray(mps.e_occupations) - pin["occ"][0]).max() < 1e-6
    cur = mps
    from renormalizer_amd.mps import mps as _m
    redone0 = _m._OPTIMISTIC_REDONE[0]
    for step in range(5):
        cur = cur.evolve(mpo, 10.0)
        occ = np.asarray(cur.e_occupations)
        assert np.abs(occ - pin["occ"][step + 1]).max() < 1e-6, (step, np.abs(occ - pin["occ"][step + 1]).max())
        assert abs(cur.evolve_config.stat["mean"] - float(pin["mean_krylov"][step + 1])) < 0.15
        assert abs(occ.sum() - 1.0) < 1e-9, (step, occ.sum())
        assert occ.min() > -1e-12
        assert abs(cur.mp_norm - 1.0) < 1e-11, (step, cur.mp_norm)
        assert abs(cur.expectation(mpo) - e0) < 1e-6 * 0.12, (step, cur.expectation(mpo) - e0)   # 4 J = 0.12 a.u.
        assert list(cur.bond_dims) == dims0
    # optimistic block QR: the rank-deficient blocks next to the chain ends break the Cholesky-QR path in each of these
    # early steps; the first such step is repeated and notes the sites ON THE STATE (round 6: the notes travel with the
    # Mps, whatever other tests evolved before in this thread), the following ones send them to Householder (a note
    # expires after eight evolves; a site that breaks down again is noted anew with doubled patience): at most two
    # repeats in five evolves, never one per evolve
    assert _m._OPTIMISTIC_REDONE[0] - redone0 <= 2, _m._OPTIMISTIC_REDONE[0] - redone0


_VARIANT = r"""
import sys, numpy as np
sys.path.insert(0, {repo!r})
import bench
from renormalizer_amd import Mpo
model, mpo, mps = bench.build_workload(25, 16, 256, 0, "physical", state_file={state!r})
for _ in range(2):                       # the second evolve starts from carried environments
    mps = mps.evolve(mpo, 10.0)
arrs = mps.to_arrays()
np.savez({out!r}, occ=np.asarray(mps.e_occupations), energy=mps.expectation(mpo), norm=mps.mp_norm,
         bond_dims=np.array(mps.bond_dims), steps=np.array(mps.evolve_config.stat["steps"]),
         **{{f"s{{i}}": a for i, a in enumerate(arrs)}})
"""


def _run_variant(state, out, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    code = _VARIANT.format(repo=REPO, state=state, out=out)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


# switch -> True: the alternative path is designed to give the same bits; False: same state to rounding
_SWITCHES = {
    "MPSE_CENTRE_MASK=0": True,       # no quantum-number tile mask of the Krylov vectors (more tiles multiplied, all zeros)
    "MPSE_ENV_CARRY=0": True,         # rebuild the environments at every step
    "MPSE_LANCZOS_ASYNC=0": False,    # host-side eigen-decomposition of the tridiagonal matrix
    "MPSE_DEFER=0": True,             # QR / environment update / absorption issued from Python after each solve returns
    "MPSE_SPLIT2=0": False,           # products with R as one workgroup per tile (other summation order)
    "MPSE_WFOLD=0": False,            # one-site matvec as the three-step chain (L.C, MPO step, .R) instead of the folded plan
    "MPSE_SMALL=0": False,            # the small centres at the chain ends through the plans instead of the one-launch matvec
    "MPSE_HEFF0=0": False,            # bond and two-level-site matvecs through the plans instead of the fused launch
    "MPSE_CHOLQR=0": False,           # every block QR by the Householder kernels (same isometry up to column phases)
    "MPSE_QR_OPTIMISTIC=0": True,     # every Cholesky-QR verified as it happens (same decompositions, same fallbacks)
    "MPSE_VEC_MASK=0": True,          # the vector kernels of a solve read and write the structurally empty tiles too
    "MPSE_ENV_WFOLD=0": False,        # MPO step of the d = 16 environment updates as a batched product instead of the elementwise pass
    "MPSE_F0_ORDER=0": True,          # fused bond / two-level-site matvec: units launched in index order (same tiles, same slots)
    "MPSE_CHOLQR_TAU=0": False,       # Cholesky-QR always in three passes (round-6 default: pass 3 skipped per block on the device)
}


@pytest.fixture(scope="module")
def variants(headline, tmp_path_factory):
    """The default path and every switched variant, two evolves each in a process of its own (the library reads its
    switches once), three processes at a time on the one GPU (the suite's longest block otherwise: 13 x 25 s in a row)."""
    from concurrent.futures import ThreadPoolExecutor
    _, _, _, state = headline
    d = tmp_path_factory.mktemp("ab")
    jobs = {"": {}}
    for sw in _SWITCHES:
        k, v = sw.split("=")
        jobs[sw] = {k: v}

    def run(item):
        name, env = item
        try:
            return name, _run_variant(state, str(d / f"v{abs(hash(name))}.npz"), env)
        except BaseException as exc:       # noqa: BLE001 - reported by the test of that switch
            return name, exc

    with ThreadPoolExecutor(max_workers=3) as pool:
        return dict(pool.map(run, jobs.items()))


@pytest.mark.parametrize("switch", sorted(_SWITCHES))
def test_headline_switch_ab(variants, switch):
    """Two evolves at the headline size with one engine switch flipped, against the default path."""
    got, base = variants[switch], variants[""]
    for r in (got, base):
        if isinstance(r, BaseException):
            raise r
    assert np.array_equal(got["bond_dims"], base["bond_dims"])
    if _SWITCHES[switch]:
        assert np.array_equal(got["steps"], base["steps"])
        for i in range(len(base["bond_dims"]) - 1):
            assert np.array_equal(got[f"s{i}"], base[f"s{i}"]), (switch, i)
        return
    assert len(got["steps"]) == len(base["steps"])
    assert np.abs(got["occ"] - base["occ"]).max() < 1e-10
    assert abs(got["energy"] - base["energy"]) < 1e-10
    assert abs(got["norm"] - base["norm"]) < 1e-12
    ov = orc.mps_dot([got[f"s{i}"].conj() for i in range(50)], [base[f"s{i}"] for i in range(50)])
    assert abs(abs(ov) - 1.0) < 1e-10, abs(ov)
