"""Symbolic operators in sum-of-products form (kept API of renormalizer/model/op.py).

An ``Op`` is a product of elementary symbols, each acting on one named degree of
freedom, times a scalar factor; ``OpSum`` is a list of them.  Only what the MPO
builder and the model classes need is implemented."""
from itertools import chain
from numbers import Number

import numpy as np

from ..utils import Quantity

_PLUS = r"b^\dagger + b"
_PLUS_JOINED = r"b^\dagger+b"


def _default_qn(symbol):
    if symbol == r"a^\dagger":
        return 1
    if symbol == "a":
        return -1
    return 0


class Op:
    def __init__(self, symbol, dof, factor=1.0, qn=None):
        if not isinstance(symbol, str):
            raise TypeError(f"symbol should be a str. Got {symbol} as {type(symbol)}")
        self.symbol = symbol
        self.split_symbol = symbol.replace(_PLUS, _PLUS_JOINED).split(" ")
        n = len(self.split_symbol)
        if isinstance(dof, list):
            if len(dof) != n:
                raise ValueError(f"The number of symbols ({n}) and DoFs ({len(dof)}) differ: {symbol} {dof}")
            self.dofs = list(dof)
        else:
            self.dofs = [dof] * n
        if isinstance(factor, Quantity):
            factor = factor.as_au()
        self.factor = factor
        if qn is None:
            qn_list = [np.array([_default_qn(s)]) for s in self.split_symbol]
        elif isinstance(qn, (int, np.integer)):
            if n != 1:
                raise ValueError("an integer qn is only valid for a single symbol")
            qn_list = [np.array([int(qn)])]
        else:
            qn = list(qn)
            if len(qn) != n:
                raise ValueError(f"The number of symbols ({n}) and quantum numbers ({len(qn)}) differ")
            qn_list = [np.atleast_1d(np.array(q, dtype=int)) for q in qn]
        self.qn_list = qn_list

    # ---- construction helpers
    @classmethod
    def product(cls, op_list):
        symbol = " ".join(o.symbol for o in op_list)
        dofs = list(chain.from_iterable(o.dofs for o in op_list))
        factor = 1.0
        for o in op_list:
            factor = factor * o.factor
        qn = list(chain.from_iterable(o.qn_list for o in op_list))
        return cls(symbol, dofs, factor, qn)

    @classmethod
    def identity(cls, dof, qn_size=1, factor=1.0):
        if isinstance(dof, list):
            return cls(" ".join(["I"] * len(dof)), dof, factor, qn=[np.zeros(qn_size, dtype=int)] * len(dof))
        return cls("I", dof, factor, qn=[np.zeros(qn_size, dtype=int)])

    # ---- properties
    @property
    def qn_size(self):
        return len(self.qn_list[0])

    @property
    def qn(self):
        return sum(self.qn_list)

    @property
    def is_identity(self):
        return all(s == "I" for s in self.split_symbol)

    def to_tuple(self):
        """(symbol, dofs, factor, quantum numbers) as nested tuples: hashable, used by result dumps (op.py:321-332)"""
        return self.symbol, tuple(self.dofs), self.factor, tuple(tuple(np.atleast_1d(q).tolist()) for q in self.qn_list)

    def split_by_dof(self):
        """[(dof, Op restricted to that dof)] in order of first appearance; factor on the first."""
        groups = {}
        for s, d, q in zip(self.split_symbol, self.dofs, self.qn_list):
            groups.setdefault(d, ([], []))
            groups[d][0].append(s)
            groups[d][1].append(q)
        out = []
        for i, (d, (syms, qns)) in enumerate(groups.items()):
            out.append((d, Op(" ".join(syms), d, self.factor if i == 0 else 1.0, qns)))
        return out

    # ---- algebra
    def __neg__(self):
        return Op(self.symbol, self.dofs, -self.factor, self.qn_list)

    def __mul__(self, other):
        if isinstance(other, Quantity):
            other = other.as_au()
        if isinstance(other, Number):
            return Op(self.symbol, self.dofs, self.factor * other, self.qn_list)
        if isinstance(other, Op):
            return Op.product([self, other])
        if isinstance(other, OpSum):
            return OpSum([self * o for o in other])
        return NotImplemented

    def __rmul__(self, other):
        if isinstance(other, Quantity):
            other = other.as_au()
        if isinstance(other, Number):
            return Op(self.symbol, self.dofs, self.factor * other, self.qn_list)
        return NotImplemented

    def __truediv__(self, other):
        return self * (1.0 / other)

    def __add__(self, other):
        if isinstance(other, Op):
            return OpSum([self, other])
        if isinstance(other, list):
            return OpSum([self] + list(other))
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, Op):
            return OpSum([self, -other])
        if isinstance(other, list):
            return OpSum([self] + [-o for o in other])
        return NotImplemented

    def __repr__(self):
        return f"Op({self.symbol!r}, {self.dofs!r}, {self.factor!r})"


class OpSum(list):
    def __add__(self, other):
        if isinstance(other, Op):
            return OpSum(list(self) + [other])
        return OpSum(list(self) + list(other))

    def __iadd__(self, other):
        if isinstance(other, Op):
            self.append(other)
        else:
            self.extend(other)
        return self

    def __neg__(self):
        return OpSum([-o for o in self])

    def __sub__(self, other):
        if isinstance(other, Op):
            return OpSum(list(self) + [-other])
        return OpSum(list(self) + [-o for o in other])

    def __mul__(self, other):
        if isinstance(other, (Number, Quantity, Op)):
            return OpSum([o * other for o in self])
        if isinstance(other, list):
            return OpSum([a * b for a in self for b in other])
        return NotImplemented

    def __rmul__(self, other):
        if isinstance(other, (Number, Quantity)):
            return OpSum([other * o for o in self])
        if isinstance(other, Op):
            return OpSum([other * o for o in self])
        return NotImplemented
