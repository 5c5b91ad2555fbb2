"""Edge cases of the hot path on the GPU: Krylov breakdown / zero vectors / tiny spaces, invalid quantum numbers,
degenerate shapes (bond dimension 1, empty blocks, two-site chains), error reporting of the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import mps_oracle as orc
from renormalizer_amd import (BasisHalfSpin, CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, Model, Mpo, Op)
from renormalizer_amd import engine as E
from renormalizer_amd.lib.krylov import expm_krylov
from renormalizer_amd.mps import svd_qn
from renormalizer_amd.mps.hop_expr import hop_expr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return E.get_engine()


def _herm_env(rng, D, w):
    a = rng.standard_normal((D, w, D)) + 1j * rng.standard_normal((D, w, D))
    return (a + a.transpose(2, 1, 0).conj()) / 4


def test_krylov_breakdown_on_an_eigenvector(eng):
    """krylov.py:72-74: beta < 100 n eps stops the recurrence; an eigenvector start gives a 1-dimensional space"""
    rng = np.random.default_rng(1)
    D, d, w = 5, 3, 2
    l, r = _herm_env(rng, D, w), _herm_env(rng, D, w)
    wm = rng.standard_normal((w, d, d, w))
    wm = (wm + wm.transpose(0, 2, 1, 3)) / 2
    h = orc.hop_dense(l, r, [wm])
    h = (h + h.conj().T) / 2
    ev, U = np.linalg.eigh(h)
    c = U[:, 3].reshape(D, d, D)
    hop = hop_expr(eng.asdevice(l), eng.asdevice(r), [eng.asdevice(wm)], c.shape)
    out, nv = expm_krylov(hop, -0.3j, eng.asdevice(c))
    ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r, [wm], y.reshape(c.shape)).ravel(), -0.3j, c.ravel())
    assert nv == nref and nv <= 2
    assert np.abs(out.to_host().ravel() - np.exp(-0.3j * ev[3]) * c.ravel()).max() < 1e-10


def test_krylov_zero_vector_is_an_error(eng):
    rng = np.random.default_rng(2)
    l, r = _herm_env(rng, 3, 2), _herm_env(rng, 3, 2)
    wm = rng.standard_normal((2, 2, 2, 2))
    hop = hop_expr(eng.asdevice(l), eng.asdevice(r), [eng.asdevice(wm)], (3, 2, 3))
    with pytest.raises(E.EngineError):
        expm_krylov(hop, -0.1j, eng.zeros((3, 2, 3), np.complex128))


def test_krylov_space_equals_full_space(eng):
    """krylov.py:59-61: n = 2 centre, the second Lanczos vector completes the space and the result is exact"""
    one = np.ones((1, 1, 1))
    wm = np.array([[0.3, 0.7], [0.7, -0.2]]).reshape(1, 2, 2, 1)
    c = np.array([0.6, 0.8j]).reshape(1, 2, 1)
    hop = hop_expr(eng.asdevice(one), eng.asdevice(one), [eng.asdevice(wm)], c.shape)
    out, nv = expm_krylov(hop, -1.7j, eng.asdevice(c))
    ev, U = np.linalg.eigh(wm[0, :, :, 0])
    exact = U @ (np.exp(-1.7j * ev) * (U.T @ c.ravel()))
    assert nv == 2 and np.abs(out.to_host().ravel() - exact).max() < 1e-13


def test_invalid_quantum_number_raises_value_error(eng):
    """svd_qn.py:219-220: no row block has a partner column block"""
    c = eng.asdevice(np.ones((2, 2)))
    qnl = np.array([[0], [0]])
    qnr = np.array([[0], [0]])
    with pytest.raises(ValueError, match="Invalid quantum number"):
        svd_qn.svd_qn(c, qnl, qnr, np.array([1]), QR=True, system="L", full_matrices=False)
    with pytest.raises(ValueError):
        svd_qn.svd_qn(eng.asdevice(np.ones((2, 3))), qnl, qnr, np.array([0]), QR=True, system="L", full_matrices=False)


def test_block_decompositions_of_degenerate_shapes(eng):
    """1 x 1 and 1 x n blocks, a block with more columns than rows, and an all-zero matrix"""
    rng = np.random.default_rng(3)
    for (m, n) in ((1, 1), (1, 5), (5, 1), (3, 7)):
        a = rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
        qnl, qnr = np.zeros((m, 1), dtype=int), np.zeros((n, 1), dtype=int)
        for system in ("L", "R"):
            u, _, v, _ = svd_qn.svd_qn(eng.asdevice(a), qnl, qnr, np.array([0]), QR=True, system=system, full_matrices=False)
            uh, vth = u.to_host(), v.T.to_host()
            assert np.abs(uh @ vth - a).max() < 1e-12
            iso = uh if system == "L" else vth.conj().T
            assert np.abs(iso.conj().T @ iso - np.eye(iso.shape[1])).max() < 1e-12
        u, s, _, v, _, _ = svd_qn.svd_qn(eng.asdevice(a), qnl, qnr, np.array([0]), full_matrices=False)
        assert np.abs((u.to_host() * s) @ v.T.to_host() - a).max() < 1e-12
        assert np.all(np.diff(s) <= 1e-14)
    z = np.zeros((4, 3))
    u, _, v, _ = svd_qn.svd_qn(eng.asdevice(z), np.zeros((4, 1), dtype=int), np.zeros((3, 1), dtype=int), np.array([0]),
                               QR=True, system="L", full_matrices=False)
    uh = u.to_host()
    assert np.abs(uh.T @ uh - np.eye(3)).max() < 1e-13 and np.abs(v.T.to_host()).max() == 0     # Q stays an isometry


def test_two_site_chain_tdvp_ps_vs_oracle():
    """the smallest chain: both sites are edge sites, every environment is the 1 x 1 x 1 sentinel"""
    from renormalizer_amd.mps.mps import Mps
    ham = Op("sigma_+ sigma_-", [0, 1]) + Op("sigma_+ sigma_-", [1, 0]) + Op("Z", 0, 0.3)
    model = Model([BasisHalfSpin(0), BasisHalfSpin(1)], ham)
    mpo = Mpo(model)
    mps = Mps.hartree_product_state(model, {0: [0.6, 0.8], 1: [1.0, 0.0]})
    mps.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
    mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=2)
    mps = mps.expand_bond_dimension(mpo, include_ex=False)
    sig = [np.asarray(b.sigmaqn).reshape(b.nbas, -1) for b in model.basis]
    st = orc.MpsState([t.to_host() for t in mps], [np.asarray(q) for q in mps.qn], mps.qnidx, mps.qntot, mps.to_right,
                      sig, mps.coeff)
    w = [mpo[i] for i in range(2)]
    z = Mpo(model, Op("Z", 1))
    for _ in range(3):
        mps = mps.evolve(mpo, 0.4)
        st = orc.tdvp_ps_step(st, w, 0.4)
        assert abs(mps.expectation(z) - orc.expectation(st.sites, [z[i] for i in range(2)])) < 1e-10
        assert abs(mps.expectation(mpo) - orc.expectation(st.sites, w)) < 1e-10


def test_engine_reports_errors_instead_of_crashing(eng):
    a = eng.asdevice(np.ones((4, 3)))
    b = eng.asdevice(np.ones((5, 2)))
    with pytest.raises(ValueError):
        eng.matmul(a, b)
    d = E.mpse_gemm_desc()
    d.dtype_a = d.dtype_b = E.F64
    d.m_a, d.k_a, d.k_b, d.n_b = E.idx1(4, 3), E.idx1(3, 1), E.idx1(5, 2), E.idx1(2, 1)     # K extents disagree
    d.m_c, d.n_c = E.idx1(4, 2), E.idx1(2, 1)
    d.batch = 1
    out = eng.empty((4, 2), np.float64)
    st = eng.lib.mpse_gemm(eng.ctx, C.byref(d), a.ptr, b.ptr, out.ptr)
    assert st == 2                                                     # MPSE_ERR_SHAPE
    assert b"extents disagree" in eng.lib.mpse_last_error(eng.ctx)
    with pytest.raises(E.DeviceMemoryError):
        eng.empty((1 << 20, 1 << 20), np.complex128)                   # 16 TiB: MPSE_ERR_OOM, not a crash
    assert eng.lib.mpse_gemm(eng.ctx, None, a.ptr, b.ptr, out.ptr) == 5    # MPSE_ERR_ARG
    # the context is still usable afterwards
    assert np.allclose(eng.matmul(a, eng.asdevice(np.ones((3, 2)))).to_host(), 3.0)


def test_block_qr_columns_with_denormal_squared_norms(eng):
    """state preparation meets columns of norm 1e-160 and below: the squared norm is denormal or zero.  Such columns
    get H = I (their sub-diagonal part, < 1e-140 in absolute terms, is dropped); nothing may turn into NaN
    (regression: NaN in the bond expansion of the headline state)"""
    rng = np.random.default_rng(7)
    m, n = 300, 12
    a = rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))
    a = a * np.array([1.0, 1e-100, 1e-150, 1e-160, 1e-165, 1e-170, 1e-200, 0.0, 1.0, 1e-158, 1e-162, 3.0])
    qnl, qnr = np.zeros((m, 1), dtype=int), np.zeros((n, 1), dtype=int)
    u, _, v, _ = svd_qn.svd_qn(eng.asdevice(a), qnl, qnr, np.array([0]), QR=True, system="L", full_matrices=False)
    q, r = u.to_host(), v.T.to_host()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(r))
    assert np.abs(q.conj().T @ q - np.eye(n)).max() < 1e-12
    err = np.abs(q @ r - a).max(axis=0)
    scale = np.abs(a).max(axis=0)
    assert np.all(err <= 1e-12 * scale + 1e-139)


def _raw_lanczos(eng, hop, dt, c, max_dim):
    v = eng.asdevice(c)
    out = eng.empty(v.shape, v.dtype)
    nv = C.c_int()
    st = eng.lib.mpse_expm_lanczos(eng.ctx, v.code, C.byref(hop.heff), dt.real, dt.imag, v.ptr, out.ptr, 1e-5, 1e-8,
                                   max_dim, C.byref(nv))
    return st, out, nv.value


def test_krylov_dimension_limit_and_estimate_schedule(eng):
    """The first convergence estimate (j = 4) is formed at the second check (j = 6); every way the recurrence can end
    around those two checks must still behave like the reference: Krylov dimension limits that fall on / between the
    checks report MPSE_ERR_NOCONV without crashing, and for time steps that need 5 ... 15 vectors the result and the
    number of vectors are the oracle's."""
    rng = np.random.default_rng(5)
    D, d, w = 6, 4, 3
    l, r = _herm_env(rng, D, w), _herm_env(rng, D, w)
    wm = rng.standard_normal((w, d, d, w))
    wm = (wm + wm.transpose(0, 2, 1, 3)) / 2
    c = rng.standard_normal((D, d, D)) + 1j * rng.standard_normal((D, d, D))
    hop = hop_expr(eng.asdevice(l), eng.asdevice(r), [eng.asdevice(wm)], c.shape)
    scale = np.abs(np.linalg.eigvalsh((lambda h: (h + h.conj().T) / 2)(orc.hop_dense(l, r, [wm])))).max()
    seen = set()
    for x in (0.02, 0.2, 0.6, 1.5, 3.0, 6.0):
        dt = -1j * x / scale
        ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r, [wm], y.reshape(c.shape)).ravel(), dt, c.ravel())
        out, nv = expm_krylov(hop, dt, eng.asdevice(c))
        assert nv == nref, (x, nv, nref)
        assert np.abs(out.to_host().ravel() - ref).max() < 1e-10 * np.abs(ref).max()
        seen.add(nv)
    assert len(seen) >= 3 and min(seen) <= 7 and max(seen) >= 11          # several estimate schedules exercised
    dt = -1j * 6.0 / scale
    for max_dim in (3, 5, 6, 7, 8):
        st, out, nv = _raw_lanczos(eng, hop, dt, c, max_dim)
        assert st == E.MPSE_ERR_NOCONV and nv == max_dim, (max_dim, st, nv)
    st, out, nv = _raw_lanczos(eng, hop, dt, c, 64)
    assert st == 0 and nv == max(seen)


@pytest.mark.parametrize("shape", [(3, 3, 3), (4, 3, 2), (5, 2, 7), (1, 3, 1)])
def test_real_lanczos_odd_and_even_lengths(eng, shape):
    """real-dtype (imaginary-time) Lanczos: odd-length vectors take the 8-byte kernels, even-length ones the 16-byte
    kernels; both against the oracle"""
    rng = np.random.default_rng(sum(shape))
    Dl, d, Dr = shape
    w = 2
    l, r = _herm_env(rng, Dl, w).real.copy(), _herm_env(rng, Dr, w).real.copy()
    l, r = (l + l.transpose(2, 1, 0)) / 2, (r + r.transpose(2, 1, 0)) / 2
    wm = rng.standard_normal((w, d, d, w))
    wm = (wm + wm.transpose(0, 2, 1, 3)) / 2
    c = rng.standard_normal(shape)
    hop = hop_expr(eng.asdevice(l), eng.asdevice(r), [eng.asdevice(wm)], c.shape)
    out, nv = expm_krylov(hop, -0.4, eng.asdevice(c))
    assert not out.is_complex
    ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r, [wm], y.reshape(c.shape)).ravel(), -0.4, c.ravel())
    assert nv == nref
    assert np.abs(out.to_host().ravel() - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


def test_engine_used_from_a_worker_thread():
    """A second context driven entirely from another thread (a new thread's current HIP device is 0, so every entry
    point binds the context's device itself), while buffers of the main engine are released concurrently."""
    import threading
    res = {}

    def work():
        e2 = E.Engine(0)
        E.use_engine(e2)
        try:
            rng = np.random.default_rng(5)
            a, b = rng.standard_normal((70, 33)), rng.standard_normal((33, 45))
            for _ in range(20):
                out = e2.matmul(e2.asdevice(a), e2.asdevice(b)).to_host()
            res["err"] = float(np.abs(out - a @ b).max())
        finally:
            E.use_engine(None)
            e2.close()

    t = threading.Thread(target=work)
    t.start()
    main = E.get_engine()
    for _ in range(200):                      # allocator traffic on the main context while the worker runs
        main.zeros((64, 64))
    t.join()
    assert res["err"] < 1e-12


def test_mpdm_complex_promotion_after_apply():
    """A real density operator times a complex operator is complex - in its tensors AND in its dtype flag (the
    reference calls to_complex(inplace=True), mpdm.py:140-141): later sums / products must allocate complex results."""
    from renormalizer_amd import HolsteinModel, Mol, Phonon, Quantity
    from renormalizer_amd.mps import MpDm
    ph = [Phonon.simple_phonon(Quantity(0.01), Quantity(3.0), 3)]
    model = HolsteinModel([Mol(Quantity(0.1), ph)] * 2, Quantity(0.02), 3)
    h = Mpo(model)
    gs = MpDm.max_entangled_gs(model)
    assert not gs.is_complex
    ev = gs.evolve_exact(h, 5.0, "GS")
    assert ev.is_complex and all(t.is_complex for t in ev)
    twice = h.contract(ev)
    assert twice.is_complex
    dense = ev.todense()
    assert np.abs((ev + ev).todense() - 2 * dense).max() < 1e-12
    c = ev.conj()
    assert abs(c.coeff - np.conjugate(ev.coeff)) < 1e-15


def test_mps_add_with_different_coeff():
    """mps/mps.py:1802-1808: prefactors that differ are folded into the states before the direct sum"""
    from renormalizer_amd import HolsteinModel, Mol, Phonon, Quantity
    from renormalizer_amd.mps.mps import Mps
    ph = [Phonon.simple_phonon(Quantity(0.01), Quantity(3.0), 3)]
    model = HolsteinModel([Mol(Quantity(0.1), ph)] * 2, Quantity(0.02), 3)
    a = Mps.random(model, 1, 4, rng=np.random.default_rng(1))
    b = Mps.random(model, 1, 4, rng=np.random.default_rng(2))
    a.coeff = 0.5j
    b.coeff = 2.0
    assert np.abs((a + b).todense() - (a.todense() + b.todense())).max() < 1e-13


def test_rccl_collective_through_ctypes(tmp_path, monkeypatch):
    """parallel.RcclCollective (librccl.so bound with ctypes, communicator on the engine's device and stream): a
    one-rank communicator on this box's GPU - unique-id rendezvous file, all-reduce(max), all-gather, teardown.  The
    N-rank flow is the same code; its bookkeeping is covered by tests/test_sharding_gloo.py."""
    from renormalizer_amd import parallel
    monkeypatch.setenv("MPSE_RENDEZVOUS_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29655")
    coll = parallel.RcclCollective(E.get_engine(), 0, 1)
    assert coll.kind == "rccl" and coll.world == 1
    assert os.path.exists(parallel._rendezvous_path())
    assert coll.allreduce_max(2.75) == 2.75
    coll.barrier()
    rows = parallel.gather_observables(coll, np.array([[0.25, 0.5, 0.125]]), [0], 1)
    assert rows.tolist() == [[0.25, 0.5, 0.125]]
    coll.close()
    assert not os.path.exists(parallel._rendezvous_path())


def test_collective_falls_back_to_files_when_rccl_cannot_start(tmp_path):
    """Two ranks on ONE GPU: RCCL refuses (or never completes) a communicator with two ranks on the same device.  All
    ranks vote, fall back to the file collective together and still gather their rows - a job does not die at
    start-up because its communicator cannot be created (parallel.make_collective)."""
    import subprocess
    import sys
    import textwrap
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {repo!r})
        import numpy as np
        from renormalizer_amd.engine import get_engine
        from renormalizer_amd.parallel import make_collective, gather_observables, units_of_rank
        eng = get_engine()
        coll = make_collective(eng)
        rank = coll.rank
        units = units_of_rank(4, rank, coll.world)
        rows = gather_observables(coll, np.array([[float(u), 2.0 * u] for u in units]), units, 4)
        assert np.array_equal(rows[:, 0], np.arange(4.0)) and np.array_equal(rows[:, 1], 2.0 * np.arange(4.0))
        coll.barrier()
        print("KIND", coll.kind, rank)
        coll.close()
    """))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MPSE_RCCL_TIMEOUT="25",
                   MPSE_RENDEZVOUS_DIR=str(tmp_path), MPSE_RENDEZVOUS_TAG="fallback", MASTER_PORT="29671")
        env.pop("MPSE_COLLECTIVE", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    kinds = []
    for p in procs:
        out, err = p.communicate(timeout=400)
        assert p.returncode == 0, err[-2000:]
        kinds += [ln.split()[1] for ln in out.splitlines() if ln.startswith("KIND")]
    # both ranks agree; on one device that is the file collective (a box whose RCCL accepts the shared device would
    # say rccl twice - equally fine)
    assert len(kinds) == 2 and kinds[0] == kinds[1] and kinds[0] in ("file", "rccl"), kinds


def test_asynchronous_lanczos_guesses_and_fallbacks(eng):
    """The solve that runs ahead of its convergence decision (vectors longer than 256 elements): Krylov dimension and
    result equal the oracle's when the run-ahead guess (dimension of the previous solve of the same size) is too
    short, too long or right; a step whose |dt| * spectral bound exceeds the on-device exponential's range goes
    through the host's eigen-decomposition; real vectors of odd length use the unvectorised kernels."""
    rng = np.random.default_rng(21)
    D, d, w = 9, 5, 3                                     # n = 405
    l, r = _herm_env(rng, D, w), _herm_env(rng, D, w)
    wm = rng.standard_normal((w, d, d, w))
    wm = (wm + wm.transpose(0, 2, 1, 3)) / 2
    c = rng.standard_normal((D, d, D)) + 1j * rng.standard_normal((D, d, D))
    hop = hop_expr(eng.asdevice(l), eng.asdevice(r), [eng.asdevice(wm)], c.shape)
    dims = []
    for dt in (-0.02j, -0.4j, -0.02j, -0.15j, -0.15j):              # dimensions go up, down, up, stay
        ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l, r, [wm], y.reshape(c.shape)).ravel(), dt, c.ravel())
        out, nv = expm_krylov(hop, dt, eng.asdevice(c))
        assert nv == nref, (dt, nv, nref)
        assert np.abs(out.to_host().ravel() - ref).max() < 1e-10 * np.abs(ref).max(), dt
        dims.append(nv)
    assert len(set(dims)) >= 3
    # H = 900 + small: few Krylov vectors, but |dt| times the Gershgorin bound of the tridiagonal matrix is far beyond
    # the range of the on-device series -> this check is handed to the host's eigen-decomposition
    l2, r2 = 1e-3 * l, 1e-3 * r
    l2[:, 0, :] += 30 * np.eye(D)
    r2[:, 0, :] += 30 * np.eye(D)
    w2 = 1e-3 * wm
    w2[0, :, :, 0] += np.eye(d)
    hop2 = hop_expr(eng.asdevice(l2), eng.asdevice(r2), [eng.asdevice(w2)], c.shape)
    ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(l2, r2, [w2], y.reshape(c.shape)).ravel(), -0.5j, c.ravel())
    out, nv = expm_krylov(hop2, -0.5j, eng.asdevice(c))
    assert nv == nref and nv < 20
    assert np.abs(out.to_host().ravel() - ref).max() < 1e-9 * np.abs(ref).max()
    lr, rr = l.real + l.real.transpose(2, 1, 0), r.real + r.real.transpose(2, 1, 0)
    cr = rng.standard_normal((D, d, D))                               # 405 real elements: odd length
    hop_r = hop_expr(eng.asdevice(lr), eng.asdevice(rr), [eng.asdevice(wm)], cr.shape)
    for dt in (-0.1, -0.3):
        ref, nref = orc.expm_krylov(lambda y: orc.hop_apply(lr, rr, [wm], y.reshape(cr.shape)).ravel(), dt, cr.ravel())
        out, nv = expm_krylov(hop_r, dt, eng.asdevice(cr))
        assert nv == nref and np.abs(out.to_host().ravel() - ref).max() < 1e-10 * np.abs(ref).max()
    with pytest.raises(E.EngineError):
        expm_krylov(hop, -0.1j, eng.asdevice(np.zeros_like(c)))       # zero vector: reported, not a hang


def test_deferred_calls_run_after_the_armed_solve():
    """mpse_defer_*: a recorded GEMM reads the buffer the next Lanczos solve writes; blocks freed while the list waits are
    not handed out again before it has run; discard drops everything."""
    import ctypes as C
    from renormalizer_amd.engine import get_engine
    from renormalizer_amd.lib.krylov import expm_krylov
    from renormalizer_amd.mps.hop_expr import hop_expr
    eng = get_engine()
    rng = np.random.default_rng(3)
    D, w = 24, 3
    l = rng.normal(size=(D, w, D))
    l = l + l.transpose(2, 1, 0)
    r = rng.normal(size=(D, w, D))
    r = r + r.transpose(2, 1, 0)
    hop = hop_expr(eng.asdevice(l), eng.asdevice(r), [], (D, D))
    c = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    c /= np.linalg.norm(c)
    m = rng.normal(size=(D, 7)) + 0j
    cd, md = eng.asdevice(c), eng.asdevice(m)
    ref, nref = expm_krylov(hop, -0.3j, cd)
    expect = ref.to_host() @ m
    out = eng.empty((D, D), np.complex128)
    scratch = eng.asdevice(np.full((D, 7), 7.0 + 0j))
    with eng.recording(1):
        prod = eng.matmul(out, md)
        ptr = scratch.ptr
        del scratch                       # freed while the list is open: must stay out of the pool
        other = eng.empty((D, 7), np.complex128)
        assert other.ptr != ptr
    assert eng.recording_list == -1
    eng.arm(1)
    res, n = expm_krylov(hop, -0.3j, cd, out=out)
    assert n == nref and res is out
    assert np.abs(prod.to_host() - expect).max() < 1e-13
    again = eng.empty((D, 7), np.complex128)      # the held block is back in the pool now
    del again
    # an unarmed list does not run; discard empties it
    with eng.recording(0):
        prod2 = eng.matmul(out, md)
    eng._check(eng.lib.mpse_memset_zero(eng.ctx, prod2.ptr, prod2.nbytes))
    expm_krylov(hop, -0.3j, cd)
    assert np.all(prod2.to_host() == 0)
    eng._check(eng.lib.mpse_defer_run(eng.ctx, 0))
    assert np.abs(prod2.to_host() - expect).max() < 1e-13
    with eng.recording(0):
        eng.matmul(out, md)
    eng.defer_discard()
    assert eng.lib.mpse_defer_run(eng.ctx, 0) == 0
    # nesting and solving while recording are refused
    with eng.recording(0):
        assert eng.lib.mpse_defer_begin(eng.ctx, 1) != 0
        with pytest.raises(Exception):
            expm_krylov(hop, -0.3j, cd)
    eng.defer_discard()


def test_bench_two_ranks_fail_loudly_without_rccl(tmp_path):
    """The driver's N = 2 command line (torch.distributed.run, one rank per GPU) with both ranks forced onto ONE device,
    which RCCL does not accept: a scaling run whose communicator cannot be created must END - non-zero exit on every
    rank, no JSON line, an error that names the way out - instead of printing a number obtained through the file
    collective (verdict round 3, item 4).  (A box whose RCCL accepts two ranks on one device prints a line with
    "collective": "rccl" and the communicator's own rank / device report - equally fine.)"""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPSE_RCCL_TIMEOUT="25", MPSE_RENDEZVOUS_DIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MPSE_COLLECTIVE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29677", os.path.join(repo, "bench.py"), "--gpus", "2", "--share-gpu",
           "--steps", "1", "--warmup", "0", "--cpu-updates", "0", "--nmol", "3", "--pdim", "4", "--bond-dim", "16"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode == 0:
        d = json.loads(lines[-1])
        assert d["config"]["collective"] == "rccl" and d["config"]["rccl_comm_count"] == [2, 2]
        assert d["config"]["rccl_comm_user_rank"] == [0, 1]
        return
    assert not lines, "a failed scaling run must not print a result line"
    assert "CollectiveUnavailable" in r.stderr and "MPSE_COLLECTIVE=file" in r.stderr, r.stderr[-1500:]


def test_bench_gpus_2_without_a_launcher_starts_two_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher starts the two ranks itself (verdict round 4, item 3): with both forced
    onto one device RCCL refuses the communicator, the strict vote fails on BOTH ranks (two ranks were really running:
    each reports its own rank in the error), no JSON line, non-zero exit.  (A box whose RCCL accepts two ranks on one
    device prints a line that says 2 ranks - equally fine.)"""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPSE_RCCL_TIMEOUT="25", MPSE_RENDEZVOUS_DIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("MPSE_COLLECTIVE", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MPSE_RENDEZVOUS_TAG"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "0",
           "--cpu-updates", "0", "--nmol", "3", "--pdim", "4", "--bond-dim", "16"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    assert "started 2 ranks" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode == 0:
        d = json.loads(lines[-1])
        assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2
        return
    assert not lines, "a failed scaling run must not print a result line"
    assert "rank 0" in r.stderr and "rank 1" in r.stderr and "CollectiveUnavailable" in r.stderr, r.stderr[-1500:]
