"""Dev-container stand-in for the `opt_einsum` package (absent from this image).

TEST INFRASTRUCTURE ONLY: lets `oracle/gen_golden.py` import the read-only
reference from /root/reference to produce golden vectors.  Never shipped to or
used on the GPU box, never imported by renormalizer_amd.

Implements just the two entry points the reference calls
(`contract`, `contract_expression(..., constants=[...])`) on top of
numpy.einsum with an unlimited-memory optimal path (np.einsum's default memory
cap silently degrades to a naive loop).
"""
import numpy as np

_PATH_CACHE = {}


def _path(subs, shapes):
    key = (subs, shapes)
    if key not in _PATH_CACHE:
        ops = [np.empty(s) for s in shapes]
        _PATH_CACHE[key] = np.einsum_path(subs, *ops, optimize=("optimal", 2 ** 62))[0]
    return _PATH_CACHE[key]


def contract(subs, *ops, **kw):
    ops = [np.asarray(o) for o in ops]
    return np.einsum(subs, *ops, optimize=_path(subs, tuple(o.shape for o in ops)))


def contract_expression(subs, *ops, constants=(), **kw):
    constants = set(constants)
    shapes = tuple(tuple(np.shape(o)) if i in constants else tuple(o) for i, o in enumerate(ops))
    path = _path(subs, shapes)
    var_pos = [i for i in range(len(ops)) if i not in constants]
    fixed = [np.asarray(o) if i in constants else None for i, o in enumerate(ops)]

    def expr(*xs, **kw2):
        full = list(fixed)
        for p, x in zip(var_pos, xs):
            full[p] = np.asarray(x)
        return np.einsum(subs, *full, optimize=path)

    return expr
