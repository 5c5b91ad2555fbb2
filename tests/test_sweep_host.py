"""Host logic of the TDVP-PS driver that needs no GPU: when the environments left behind by one step may be taken
over by the next (renormalizer_amd/mps/mps.py::_carry_environ / _carried_environ)."""
import types

import renormalizer_amd.mps.mps as M


class _Env:
    def __init__(self):
        self.dropped = []

    def drop(self, domain):
        self.dropped.append(domain)


class _Mps:
    def __init__(self, sites, to_right):
        self._mp, self.to_right = list(sites), to_right

    def __len__(self):
        return len(self._mp)


def test_environments_are_taken_over_only_from_identical_objects(monkeypatch):
    monkeypatch.delenv("MPSE_ENV_CARRY", raising=False)
    sites = [object() for _ in range(5)]
    mpo = types.SimpleNamespace(_mp=[object() for _ in range(5)])
    env = _Env()
    M._carry_environ(_Mps(sites, True), mpo, env)
    assert env.dropped == ["L"]                       # only the environments ahead of the next sweep are kept
    # same objects (the centre site 0 may have been replaced by normalize): taken over, and the slot is used up
    assert M._carried_environ(_Mps([object()] + sites[1:], True), mpo, "R") is env
    assert M._carried_environ(_Mps(sites, True), mpo, "R") is None
    for bad in (
        lambda: M._carried_environ(_Mps(sites[:2] + [object()] + sites[3:], True), mpo, "R"),       # another site tensor
        lambda: M._carried_environ(_Mps(sites, True), types.SimpleNamespace(_mp=list(mpo._mp)), "R"),  # another MPO
        lambda: M._carried_environ(_Mps(sites, False), mpo, "L"),                                     # other direction
        lambda: M._carried_environ(_Mps(sites[:4], True), mpo, "R"),                                  # other length
    ):
        M._carry_environ(_Mps(sites, True), mpo, env)
        assert bad() is None
    # an MPO site replaced in place
    M._carry_environ(_Mps(sites, True), mpo, env)
    mpo._mp[3] = object()
    assert M._carried_environ(_Mps(sites, True), mpo, "R") is None
    # left-moving start: the last site is the centre
    M._carry_environ(_Mps(sites, False), mpo, env)
    assert M._carried_environ(_Mps(sites[:4] + [object()], False), mpo, "L") is env
    # switched off
    M._carry_environ(_Mps(sites, True), mpo, env)
    monkeypatch.setenv("MPSE_ENV_CARRY", "0")
    assert M._carried_environ(_Mps(sites, True), mpo, "R") is None


def test_centre_tile_mask_matches_the_allowed_pattern():
    """hop_expr.centre_tile_mask: byte [tn][kt] marks the 16 x 64 tiles of C[a, (sigma, b)] that hold an entry allowed by
    the quantum numbers (include/mpsengine.h, mpse_expm_centre_mask)."""
    import numpy as np
    import renormalizer_amd.mps.hop_expr as H

    class _Eng:
        def asdevice(self, a):
            return np.array(a)

    rng = np.random.default_rng(0)
    eng = _Eng()
    for (Dl, d, Dr, nq) in ((40, 3, 70, 1), (16, 16, 64, 2), (5, 2, 3, 1)):
        qnl, sig, qnr = (rng.integers(0, 2, size=(n, nq)) for n in (Dl, d, Dr))
        qt = np.ones(nq, dtype=int)
        ql = qnl[:, None, :] + sig[None, :, :]
        H._CMASK_CACHE.clear()
        m = H.centre_tile_mask(eng, ql, qnr, qt, (Dl, d, Dr))
        allowed = ((qnl[:, None, None, :] + sig[None, :, None, :] + qnr[None, None, :, :]) == qt).all(-1).reshape(Dl, d * Dr)
        nkt, ntn = (Dl + 15) // 16, (d * Dr + 63) // 64
        nkw = (nkt + 7) // 8
        ref = np.zeros((ntn, nkw * 8), np.uint8)
        for kt in range(nkt):
            for tn in range(ntn):
                ref[tn, kt] = allowed[kt * 16:(kt + 1) * 16, tn * 64:(tn + 1) * 64].any()
        assert np.array_equal(m.view(np.uint8).reshape(ntn, nkw * 8), ref)
        assert H.centre_tile_mask(eng, ql, qnr, qt, (Dl, d, Dr)) is m        # cached
    assert H.centre_tile_mask(eng, ql, qnr, qt, (Dl + 1, d, Dr)) is None      # quantum numbers of another shape


def test_qr_notes_travel_with_the_state():
    """The (site, direction) pairs whose block QR fell back to the Householder kernels are noted ON the state (round 6;
    rounds 4-5 kept them in a thread-local keyed by the chain's shape, so that an evolve depended on what the thread had
    evolved before): a copy starts from its parent's notes and evolves its own, a note expires after its patience, and a
    site that breaks down again is noted with doubled patience."""
    class Dummy:
        pass

    a = Dummy()
    notes = M._householder_sites(a)
    assert notes is M._householder_sites(a) and (3, True) not in notes
    notes.note((3, True))
    assert (3, True) in notes and (3, False) not in notes
    b = Dummy()
    b._qr_notes = notes.copy()                    # what Mps.metacopy does
    for _ in range(M._QrNotes.FIRST):
        M._householder_sites(b).tick()
    assert (3, True) not in M._householder_sites(b) and (3, True) in notes      # the copy's clock, not the parent's
    M._householder_sites(b).note((3, True))       # breaks down again: twice the patience
    for _ in range(M._QrNotes.FIRST):
        M._householder_sites(b).tick()
    assert (3, True) in M._householder_sites(b)
    for _ in range(M._QrNotes.FIRST):
        M._householder_sites(b).tick()
    assert (3, True) not in M._householder_sites(b)
    # the real class copies them in metacopy
    import inspect
    assert "_qr_notes" in inspect.getsource(M.Mps.metacopy)
