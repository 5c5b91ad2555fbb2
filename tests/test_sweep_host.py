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
