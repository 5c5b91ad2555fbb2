"""GPU parity of the observable layer against values the real reference computed
for a fixed complex MPS (tests/golden/observables_holstein_small.npz, oracle/gen_golden.py obs)."""
import os

import numpy as np
import pytest

from renormalizer_amd import HolsteinModel, Phonon, Mol, Quantity, CompressConfig, CompressCriteria

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def state(golden_dir):
    from renormalizer_amd.mps.mps import Mps
    z = np.load(os.path.join(golden_dir, "observables_holstein_small.npz"))
    ph = [Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), 4),
          Phonon.simple_phonon(Quantity(3.1e-3), Quantity(9.5), 3)]
    model = HolsteinModel([Mol(Quantity(0), ph)] * 3, Quantity(3.0e-2), 3)
    n = int(z["mps_nsite"])
    assert n == model.nsite
    for i, b in enumerate(model.basis):
        assert np.array_equal(np.asarray(b.sigmaqn).reshape(b.nbas, -1), z[f"sigmaqn_{i}"])
    mps = Mps.from_arrays(model, [z[f"mps_site_{i}"] for i in range(n)], [z[f"mps_qn_{i}"] for i in range(n + 1)],
                          int(z["mps_qnidx"]), z["mps_qntot"], bool(z["mps_to_right"]), complex(z["mps_coeff"]))
    return z, mps


def test_occupations(state):
    z, mps = state
    assert np.abs(np.asarray(mps.e_occupations) - z["e_occupations"]).max() < 1e-12
    assert np.abs(np.asarray(mps.ph_occupations) - z["ph_occupations"]).max() < 1e-12


def test_one_site_rdm_and_entropy(state):
    z, mps = state
    rdm = mps.calc_1site_rdm()
    assert sorted(rdm) == list(range(len(mps)))
    for k, v in rdm.items():
        assert v.shape == z[f"rdm1_{k}"].shape
        assert np.abs(v - z[f"rdm1_{k}"]).max() < 1e-12
    assert set(mps.calc_1site_rdm([2, 5])) == {2, 5}
    s1 = mps.calc_entropy("1site")
    assert np.abs(np.array([s1[k] for k in range(len(mps))]) - z["site_entropy"]).max() < 1e-10


def test_edof_rdm(state):
    z, mps = state
    rho = mps.calc_edof_rdm()
    assert np.abs(rho - z["edof_rdm"]).max() < 1e-12
    assert np.allclose(np.diag(rho).real, mps.e_occupations)      # mps/tests/test_mps.py:46-53


def test_bond_entropy(state):
    z, mps = state
    sv = mps.calc_bond_singular_values()
    ref = z["bond_sv"]
    w = min(sv.shape[1], ref.shape[1])
    assert np.abs(sv[:, :w] - ref[:, :w]).max() < 1e-11
    assert np.abs(mps.calc_entropy("bond") - z["bond_entropy"]).max() < 1e-10
    # the state itself is untouched (the sweep runs on a copy)
    assert np.abs(np.asarray(mps.e_occupations) - z["e_occupations"]).max() < 1e-12
    # test_mps.py:74-88: the first / last bond entropy equals the one-site entropy of the edge sites
    s1 = mps.calc_entropy("1site")
    sb = mps.calc_entropy("bond")
    assert abs(sb[0] - s1[0]) < 1e-10 and abs(sb[-1] - s1[len(mps) - 1]) < 1e-10


def test_dense_round_trip():
    """mps/tests/test_mps.py:91-96: todense / from_dense; and <H> from the dense vector against the MPS contraction"""
    from renormalizer_amd import BasisSimpleElectron, Model, Mpo, Op
    from renormalizer_amd.mps.mps import Mps
    ham = sum((Op(r"a^\dagger a", [i, i + 1], 0.3) + Op(r"a^\dagger a", [i + 1, i], 0.3) for i in range(4)),
              Op(r"a^\dagger a", 0, 0.1))
    model = Model([BasisSimpleElectron(i) for i in range(5)], ham)
    ref = Mps.random(model, 1, 20, rng=np.random.default_rng(4))
    dense = ref.todense()
    assert dense.shape == (2,) * 5 and abs(np.linalg.norm(dense) - 1) < 1e-12
    loaded = Mps.from_dense(model, dense)
    assert np.abs(loaded.todense() - dense).max() < 1e-13
    h = Mpo(model)
    hd = h.todense() if hasattr(h, "todense") else None
    if hd is not None:
        v = dense.ravel()
        assert abs(v.conj() @ hd.reshape(32, 32) @ v - ref.expectation(h)) < 1e-12


def test_two_site_rdm_and_mutual_entropy(state):
    """mps/mps.py:1600-1655, 1734-1757 and mps/tests/test_mps.py:74-88"""
    z, mps = state
    rdm2 = mps.calc_2site_rdm()
    n = len(mps)
    assert len(rdm2) == n * (n - 1) // 2
    for (i, j) in ((0, 1), (0, 4), (2, 3), (3, 8), (7, 8)):
        ref = z[f"rdm2_{i}_{j}"]
        assert rdm2[(i, j)].shape == ref.shape
        assert np.abs(rdm2[(i, j)] - ref).max() < 1e-12
    s2 = mps.calc_entropy("2site")
    for i, j, v in z["pair_entropy"]:
        assert abs(s2[(int(i), int(j))] - v) < 1e-9
    assert np.abs(mps.calc_entropy("mutual") - z["mutual_entropy"]).max() < 1e-9
    sb = mps.calc_entropy("bond")
    assert abs(sb[1] - s2[(0, 1)]) < 1e-9 and abs(sb[-2] - s2[(n - 2, n - 1)]) < 1e-9


def test_canonical_checks_angle_and_size(state):
    """mps/tests/test_mp.py: canonical-form predicates; angle; memory accounting"""
    z, mps = state
    m = mps.copy()
    m.ensure_left_canonical()
    assert m.check_left_canonical() and not m.check_right_canonical()
    m.ensure_right_canonical()
    assert m.check_right_canonical() and not m.check_left_canonical()
    assert abs(m.angle(mps) - 1.0) < 1e-12
    assert m.total_bytes == sum(int(np.prod(t.shape)) * 16 for t in m)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("kind", ["mps", "mpdm"])
def test_variational_compress(cplx, kind):
    """mps/tests/test_mp.py::test_variational_compress: the variationally compressed H|psi> (two-site sweeps, then
    one-site sweeps from that result) is within 1e-4 of the exact product, for pure states and density operators,
    real and complex"""
    from renormalizer_amd.mps import Mps, MpDm, Mpo
    from renormalizer_amd.utils import constant
    omega = [Quantity(106.51, "cm^{-1}"), Quantity(1555.55, "cm^{-1}")]
    ph_list = [Phonon.simple_phonon(o, d, 4) for o, d in zip(omega, [Quantity(30.1370), Quantity(8.7729)])]
    j = np.array([[0.0, -0.1, -0.2], [-0.1, 0.0, -0.3], [-0.2, -0.3, 0.0]]) / constant.au2ev
    model = HolsteinModel([Mol(Quantity(2.67, "eV"), ph_list, 15.45)] * 3, j, 3)
    mps = Mps.random(model, 1, 10, rng=np.random.default_rng(3))
    if kind == "mpdm":
        mps = MpDm.from_mps(mps)
    mps.canonicalise().normalize("mps_only")
    M = 36
    mpo = Mpo(model)
    if cplx:
        mps = mps.to_complex()
        mpo = mpo.scale(-1.0j)
    std = mpo.apply(mps, canonicalise=True).canonicalise()
    mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=M, vmethod="2site",
                                         vprocedure=[[M, 1.0], [M, 0.2], [M, 0.1]] + [[M, 0]] * 10)
    var = mps.variational_compress(mpo, guess=None)
    assert max(var.bond_dims) <= M
    assert var.distance(std) / std.mp_norm < 1e-4 and abs(var.mp_norm - std.mp_norm) < 1e-4
    var.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=M, vmethod="1site", vprocedure=[[M, 0]] * 10)
    var1 = mps.variational_compress(mpo, guess=var)
    assert var1.distance(std) / std.mp_norm < 1e-4 and abs(var1.mp_norm - std.mp_norm) < 1e-4
    via_contract = mpo.contract(mps, algo="variational")
    assert via_contract.distance(std) / std.mp_norm < 1e-4


def test_evolve_exact_digest_and_threshold():
    """Mps.evolve_exact (mps/mps.py:1519-1523): the bond-dimension-1 propagator of the vibrational Hamiltonian in the
    electron-free space against the dense exponential; `digest` and the `threshold` shortcut"""
    import scipy.linalg
    from renormalizer_amd.mps import Mps, Mpo
    ph = [Phonon.simple_phonon(Quantity(0.01), Quantity(3.0), 3), Phonon.simple_phonon(Quantity(0.017), Quantity(1.0), 3)]
    model = HolsteinModel([Mol(Quantity(0.1), ph)] * 2, Quantity(0.02), 3)
    mps = Mps.random(model, 0, 4, rng=np.random.default_rng(5)).to_complex()
    mps.canonicalise().normalize("mps_and_coeff")
    h = Mpo(model)
    dt = 7.0
    out = mps.evolve_exact(h, dt, "GS")
    dense_h = h.todense()
    psi = mps.todense().ravel()
    ref = scipy.linalg.expm(-1j * dt * (dense_h - model.gs_zpe * np.eye(len(psi)))) @ psi
    assert np.abs(out.todense().ravel() - ref).max() < 1e-12
    assert list(out.bond_dims) == list(mps.bond_dims)
    d = mps.digest
    dense = psi / mps.coeff
    assert abs(d["var"] - dense.var()) < 1e-14 and abs(d["mean"] - dense.mean()) < 1e-14
    mps.threshold = 1e-5
    assert mps.compress_config.threshold == 1e-5 and mps.threshold == 1e-5
