"""Local basis sets: dimension, quantum numbers and operator matrices of one MPS site.

Subset of renormalizer/model/basis.py used by the supported models: BasisSHO (:110-339),
BasisSimpleElectron (:882-929), BasisHalfSpin (:932-996), BasisMultiElectron(Vac) (:755-879).
Symbol tables are evaluated lazily from second-quantised building blocks."""
import numpy as np

from .op import Op


class BasisSet:
    is_electron = False
    is_phonon = False
    is_spin = False
    multi_dof = False

    def __init__(self, dof, nbas, sigmaqn):
        self.dof = dof
        self.nbas = int(nbas)
        self.sigmaqn = np.array([np.atleast_1d(q) for q in sigmaqn], dtype=int).reshape(self.nbas, -1)

    @property
    def dofs(self):
        return tuple(self.dof) if self.multi_dof else (self.dof,)

    def op_mat(self, op):
        raise NotImplementedError

    def _as_op(self, op):
        return op if isinstance(op, Op) else Op(op, None)

    def _product(self, op, single):
        """matrix of a (possibly multi-symbol) operator from a single-symbol table"""
        key = op.symbol.replace(r"b^\dagger + b", r"b^\dagger+b")
        try:
            return single(key) * op.factor
        except KeyError:
            pass
        mat = np.eye(self.nbas)
        for s in op.split_symbol:
            try:
                mat = mat @ single(s)
            except KeyError:
                raise ValueError(f"op_symbol:{s} is not supported by {type(self).__name__}") from None
        return mat * op.factor

    def copy(self, new_dof):
        """the same local basis attached to another degree of freedom (basis.py:93-108)"""
        import copy as _copy
        new = _copy.copy(self)
        new.dof = list(new_dof) if self.multi_dof else new_dof
        if self.multi_dof:
            shift = 1 if type(self).__name__ == "BasisMultiElectronVac" else 0
            new.dof_name_map = {name: i + shift for i, name in enumerate(new.dof)}
        return new

    def __repr__(self):
        return f"{type(self).__name__}(dof: {self.dof}, nbas: {self.nbas})"


class BasisSHO(BasisSet):
    """Harmonic-oscillator number basis (truncated).  ``x^2``/``p^2`` use the exact
    second-quantised forms, not squares of truncated matrices (basis.py:227-250, 298-306)."""
    is_phonon = True

    def __init__(self, dof, omega, nbas, x0=0.0):
        self.omega = float(omega)
        self.x0 = float(x0)
        super().__init__(dof, nbas, [0] * int(nbas))

    def _single(self, sym):
        n, w, x0 = self.nbas, self.omega, self.x0
        num = np.arange(n)
        if sym == "I":
            return np.eye(n)
        if sym == "b":
            return np.diag(np.sqrt(num[1:]), k=1)
        if sym == r"b^\dagger":
            return np.diag(np.sqrt(num[1:]), k=-1)
        if sym == "b b":
            return np.diag(np.sqrt(num[1:-1] * num[2:]), k=2) if n > 1 else np.zeros((1, 1))
        if sym == r"b^\dagger b^\dagger":
            return np.diag(np.sqrt(num[1:-1] * num[2:]), k=-2) if n > 1 else np.zeros((1, 1))
        if sym in (r"b^\dagger b", "n"):
            return np.diag(num.astype(float))
        if sym == r"b b^\dagger":
            return np.diag(num + 1.0)
        if sym == r"b^\dagger+b":
            return self._single(r"b^\dagger") + self._single("b")
        if sym == r"b^\dagger-b":
            return self._single(r"b^\dagger") - self._single("b")
        if sym == "x":
            return np.sqrt(0.5 / w) * self._single(r"b^\dagger+b") + np.eye(n) * x0
        if sym in ("x^2", "x x"):
            y2 = 0.5 / w * (self._single(r"b^\dagger b^\dagger") + self._single(r"b^\dagger b")
                            + self._single(r"b b^\dagger") + self._single("b b"))
            return np.eye(n) * x0 ** 2 + 2 * x0 * np.sqrt(0.5 / w) * self._single(r"b^\dagger+b") + y2
        if sym == "p":
            return 1j * np.sqrt(w / 2) * self._single(r"b^\dagger-b")
        if sym in ("p^2", "p p"):
            return -w / 2 * (self._single(r"b^\dagger b^\dagger") - self._single(r"b^\dagger b")
                             - self._single(r"b b^\dagger") + self._single("b b"))
        raise KeyError(sym)

    def op_mat(self, op):
        return self._product(self._as_op(op), self._single)


class BasisSimpleElectron(BasisSet):
    """Two states: 0 unoccupied, 1 occupied."""
    is_electron = True

    def __init__(self, dof, sigmaqn=None):
        super().__init__(dof, 2, [0, 1] if sigmaqn is None else sigmaqn)

    @staticmethod
    def _single(sym):
        m = np.zeros((2, 2))
        if sym == r"a^\dagger":
            m[1, 0] = 1.0
        elif sym == "a":
            m[0, 1] = 1.0
        elif sym == r"a^\dagger a":
            m[1, 1] = 1.0
        elif sym == "I":
            m = np.eye(2)
        else:
            raise KeyError(sym)
        return m

    def op_mat(self, op):
        return self._product(self._as_op(op), self._single)


class BasisHalfSpin(BasisSet):
    """Spin-1/2, state 0 = +z."""
    is_spin = True

    def __init__(self, dof, sigmaqn=None):
        super().__init__(dof, 2, [0, 0] if sigmaqn is None else sigmaqn)

    @staticmethod
    def _single(sym):
        if sym == "I":
            return np.eye(2)
        if sym in ("sigma_x", "X", "x"):
            return np.array([[0.0, 1.0], [1.0, 0.0]])
        if sym in ("sigma_y", "Y", "y"):
            return np.array([[0.0, -1.0j], [1.0j, 0.0]])
        if sym in ("isigma_y", "iY", "iy"):
            return np.array([[0.0, 1.0], [-1.0, 0.0]])
        if sym in ("sigma_z", "Z", "z"):
            return np.diag([1.0, -1.0])
        if sym in ("sigma_-", "-"):
            return np.array([[0.0, 0.0], [1.0, 0.0]])
        if sym in ("sigma_+", "+"):
            return np.array([[0.0, 1.0], [0.0, 0.0]])
        raise KeyError(sym)

    def op_mat(self, op):
        return self._product(self._as_op(op), self._single)


class BasisMultiElectron(BasisSet):
    """Several electronic states on one site; a^dagger_i a_j connects state j to state i."""
    is_electron = True
    multi_dof = True

    def __init__(self, dof, sigmaqn):
        assert len(dof) == len(sigmaqn)
        self.dof_name_map = {name: i for i, name in enumerate(dof)}
        super().__init__(list(dof), len(dof), sigmaqn)

    def op_mat(self, op):
        op = self._as_op(op)
        syms, dofs = op.split_symbol, op.dofs
        mat = np.zeros((self.nbas, self.nbas))
        if syms == ["I"]:
            mat = np.eye(self.nbas)
        elif syms == [r"a^\dagger", "a"]:
            mat[self.dof_name_map[dofs[0]], self.dof_name_map[dofs[1]]] = 1.0
        elif syms == [r"a^\dagger a"] or syms == [r"a^\dagger", "a"]:
            i = self.dof_name_map[dofs[0]]
            mat[i, i] = 1.0
        else:
            raise ValueError(f"op_symbol:{op.symbol} is not supported by BasisMultiElectron")
        return mat * op.factor


class BasisMultiElectronVac(BasisSet):
    """As BasisMultiElectron with an extra vacuum state at index 0."""
    is_electron = True
    multi_dof = True

    def __init__(self, dof):
        self.dof_name_map = {name: i + 1 for i, name in enumerate(dof)}
        super().__init__(list(dof), len(dof) + 1, [0] + [1] * len(dof))

    def op_mat(self, op):
        op = self._as_op(op)
        syms, dofs = op.split_symbol, op.dofs
        mat = np.zeros((self.nbas, self.nbas))
        if syms == ["I"]:
            mat = np.eye(self.nbas)
        elif syms == [r"a^\dagger", "a"]:
            mat[self.dof_name_map[dofs[0]], self.dof_name_map[dofs[1]]] = 1.0
        elif syms == [r"a^\dagger a"]:
            i = self.dof_name_map[dofs[0]]
            mat[i, i] = 1.0
        elif syms == [r"a^\dagger"]:
            mat[self.dof_name_map[dofs[0]], 0] = 1.0
        elif syms == ["a"]:
            mat[0, self.dof_name_map[dofs[0]]] = 1.0
        else:
            raise ValueError(f"op_symbol:{op.symbol} is not supported by BasisMultiElectronVac")
        return mat * op.factor
