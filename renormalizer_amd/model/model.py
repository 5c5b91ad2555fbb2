"""Model = ordered local bases + Hamiltonian terms (kept API of renormalizer/model/model.py:
Model :18-228, HolsteinModel :231-407, SpinBosonModel :410-439)."""
from collections import Counter

import numpy as np

from ..utils import Quantity
from .basis import BasisSet, BasisSHO, BasisSimpleElectron, BasisHalfSpin, BasisMultiElectronVac
from .phonon import Phonon, Mol
from .op import Op, OpSum


class Model:
    def __init__(self, basis, ham_terms, dipole=None, output_ordering=None):
        if not isinstance(basis, list) or len(basis) == 0:
            raise TypeError("Basis should be a non-empty list")
        if not isinstance(basis[0], BasisSet):
            raise TypeError("Elements of the basis list should be of type BasisSet")
        names = [d for b in basis for d in b.dofs]
        dup = [k for k, v in Counter(names).items() if v > 1]
        if dup:
            raise ValueError(f"Duplicate DoF definition found in the basis list: {dup}")
        self.basis = basis
        sizes = {b.sigmaqn.shape[1] for b in basis}
        if len(sizes) != 1:
            raise ValueError(f"Inconsistent quantum number size: {sizes}")
        self.qn_size = sizes.pop()
        self.output_ordering = basis if output_ordering is None else output_ordering
        self.dof_to_siteidx = self.order = {}
        self.dof_to_basis = {}
        for i, b in enumerate(basis):
            for d in b.dofs:
                self.dof_to_siteidx[d] = i
                self.dof_to_basis[d] = b
        self.ham_terms = self.check_operator_terms(ham_terms)
        self.dipole = dipole
        self.mpos = dict()
        self.pbond_list = [b.nbas for b in basis]

    def check_operator_terms(self, terms):
        if isinstance(terms, Op):
            terms = [terms]
        flat = []
        for t in terms:
            if isinstance(t, Op):
                flat.append(t)
            elif isinstance(t, (OpSum, list)):
                flat.extend(t)
            else:
                raise ValueError(f"Expected Op in terms. Got {type(t)}. Str representation: {t}")
        out = []
        for t in flat:
            for name in t.dofs:
                if name not in self.dof_to_siteidx:
                    raise ValueError(f"{t} contains DoF not in the basis.")
            if t.factor == 0:
                continue
            out.append(t)
        return out

    @property
    def nsite(self):
        return len(self.basis)

    @property
    def dofs(self):
        return list(self.dof_to_siteidx.keys())

    def _dofs(self, pred):
        return [d for b in self.basis if pred(b) for d in b.dofs]

    @property
    def e_dofs(self):
        return self._dofs(lambda b: b.is_electron)

    @property
    def v_dofs(self):
        return self._dofs(lambda b: b.is_phonon)

    @property
    def n_edofs(self):
        return len(self.e_dofs)

    def copy(self):
        return Model(list(self.basis), list(self.ham_terms), self.dipole, self.output_ordering)

    def to_dict(self):
        """Plain-type description for result dumps (model/model.py:213-228)"""
        return {"Hamiltonian": [op.to_tuple() for op in self.ham_terms], "dipole": self.dipole}


def construct_j_matrix(mol_num, j_constant, periodic=False):
    j = j_constant.as_au() if isinstance(j_constant, Quantity) else float(j_constant)
    m = np.zeros((mol_num, mol_num))
    for i in range(mol_num - 1):
        m[i, i + 1] = m[i + 1, i] = j
    if periodic and mol_num > 2:
        m[0, -1] = m[-1, 0] = j
    return m


class HolsteinModel(Model):
    """H = sum_ij J_ij a+_i a_j + sum (p^2/2 + w^2 x^2/2) - sum w^2 d a+_i a_i x   (model.py:231-345)"""

    def __init__(self, mol_list, j_matrix, scheme: int = 2, periodic: bool = False):
        n = len(mol_list)
        self.mol_list = mol_list
        if isinstance(j_matrix, Quantity):
            j_matrix = construct_j_matrix(n, j_matrix, periodic)
        j_matrix = np.asarray(j_matrix)
        assert j_matrix.shape[0] == n
        self.j_matrix = j_matrix
        self.scheme = scheme
        basis = []
        if scheme < 4:
            for i, mol in enumerate(mol_list):
                basis.append(BasisSimpleElectron(i))
                for k, ph in enumerate(mol.ph_list):
                    basis.append(BasisSHO((i, k), ph.omega[0], ph.n_phys_dim))
        elif scheme == 4:
            n_left = n // 2
            n_left_ph = 0
            for i, mol in enumerate(mol_list):
                for k, ph in enumerate(mol.ph_list):
                    if i < n_left:
                        n_left_ph += 1
                    basis.append(BasisSHO((i, k), ph.omega[0], ph.n_phys_dim))
            basis.insert(n_left_ph, BasisMultiElectronVac(list(range(n))))
        else:
            raise ValueError(f"invalid model.scheme: {scheme}")
        ham = []
        for i in range(n):
            for j in range(n):
                f = mol_list[i].elocalex + mol_list[i].e0 if i == j else j_matrix[i, j]
                ham.append(Op(r"a^\dagger a", [i, j], f))
        for i, mol in enumerate(mol_list):
            for k, ph in enumerate(mol.ph_list):
                ham += [Op("p^2", (i, k), 0.5), Op("x^2", (i, k), 0.5 * ph.omega[0] ** 2)]
        for i, mol in enumerate(mol_list):
            for k, ph in enumerate(mol.ph_list):
                if not np.allclose(ph.omega[0], ph.omega[1]):
                    ham.append(Op(r"a^\dagger a", i) * Op("x^2", (i, k)) * (0.5 * (ph.omega[1] ** 2 - ph.omega[0] ** 2)))
                ham.append(Op(r"a^\dagger a", i) * Op("x", (i, k)) * (-ph.omega[1] ** 2 * ph.dis[1]))
        dipole = {i: mol.dipole for i, mol in enumerate(mol_list)}
        super().__init__(basis, ham, dipole=dipole)
        self.mol_num = n

    @property
    def gs_zpe(self):
        return sum(m.gs_zpe for m in self.mol_list)

    def switch_scheme(self, scheme: int) -> "HolsteinModel":
        """the same molecules and couplings laid out in another site ordering (model/model.py:349-363)"""
        return HolsteinModel(self.mol_list, self.j_matrix, scheme)

    def copy(self):
        return HolsteinModel(self.mol_list, self.j_matrix, self.scheme)

    @property
    def j_constant(self):
        """the single non-zero electronic coupling; ValueError when the couplings differ (model/model.py:372-391)"""
        vals = set(np.asarray(self.j_matrix).ravel().tolist()) - {0.0}
        if len(vals) != 1:
            raise ValueError("J is not constant")
        return vals.pop()

    def __iter__(self):
        return iter(self.mol_list)

    def __getitem__(self, i):
        return self.mol_list[i]

    def __len__(self):
        return len(self.mol_list)


class SpinBosonModel(Model):
    """H = eps sz + Delta sx + sum (p^2 + w^2 q^2)/2 + sz sum c q   (model.py:410-439)"""

    def __init__(self, epsilon, delta, ph_list, dipole=None):
        self.epsilon = epsilon.as_au()
        self.delta = delta.as_au()
        self.ph_list = ph_list
        basis = [BasisHalfSpin("spin")]
        for k, ph in enumerate(ph_list):
            basis.append(BasisSHO(k, ph.omega[0], ph.n_phys_dim))
        ham = [Op("sigma_z", "spin", self.epsilon), Op("sigma_x", "spin", self.delta)]
        for k, ph in enumerate(ph_list):
            assert ph.is_simple
            ham += [Op("p^2", k, 0.5), Op("x^2", k, 0.5 * ph.omega[0] ** 2)]
            ham.append(Op("sigma_z", "spin") * Op("x", k) * (-ph.omega[1] ** 2 * ph.dis[1]))
        super().__init__(basis, ham, dipole={"spin": dipole or 0})


class TI1DModel(Model):
    """Translationally invariant chain with periodic boundary: ``basis`` and ``local_ham_terms`` describe one unit
    cell (degrees of freedom renamed to ("cell<i>", dof) in the full model), ``nonlocal_ham_terms`` name their
    degrees of freedom as (cell offset, dof) relative to the cell they are attached to (model/model.py:442-510)."""

    def __init__(self, basis, local_ham_terms, nonlocal_ham_terms, ncell: int):
        full_basis = []
        for i in range(ncell):
            for b in basis:
                new_dofs = [(f"cell{i}", dof) for dof in b.dofs]
                full_basis.append(b.copy(new_dofs if b.multi_dof else new_dofs[0]))
        terms = []
        for i in range(ncell):
            for op in local_ham_terms:
                terms.append(Op(op.symbol, [(f"cell{i}", dof) for dof in op.dofs], op.factor, op.qn_list))
            for op in nonlocal_ham_terms:
                dofs = []
                for off, dof in op.dofs:
                    assert isinstance(off, int)
                    dofs.append((f"cell{(i + off) % ncell}", dof))
                terms.append(Op(op.symbol, dofs, op.factor, op.qn_list))
        super().__init__(full_basis, terms)
        self.ncell = ncell


def load_from_dict(param, scheme, lam: bool):
    """(HolsteinModel, temperature) from the parameter dictionary of the transport examples (keys "mol num",
    "j constant", "ph modes", "temperature", each quantity as [value, unit]); the phonon level counts are chosen by
    ``Phonon.simplest_phonon`` at that temperature (model/model.py:523-533)."""
    temperature = Quantity(*param["temperature"])
    ph_list = [Phonon.simplest_phonon(Quantity(*omega), Quantity(*displacement), temperature=temperature, lam=lam)
               for omega, displacement in param["ph modes"]]
    model = HolsteinModel([Mol(Quantity(0), ph_list)] * param["mol num"], Quantity(*param["j constant"]), scheme)
    return model, temperature


def heisenberg_ops(nspin):
    """terms of the spin-1/2 Heisenberg chain sum_i S_i . S_{i+1} in Pauli / ladder operators (model/model.py:536-543)"""
    terms = []
    for i in range(nspin - 1):
        terms += [Op("sigma_z sigma_z", [i, i + 1], 1.0 / 4), Op("sigma_+ sigma_-", [i, i + 1], 1.0 / 2),
                  Op("sigma_- sigma_+", [i, i + 1], 1.0 / 2)]
    return terms
