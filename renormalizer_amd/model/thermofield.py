"""Thermofield (thermo-Bogoliubov) form of a Holstein model: finite temperature with pure states.

Every vibrational mode (omega, coupling g) is replaced by a physical mode and a fictitious "tilde" mode of frequency
-omega; after the Bogoliubov rotation with theta = arctanh(exp(-beta omega / 2)) the thermal state of the bath is the
vacuum of both, and the electron couples to the physical mode with g cosh(theta) and to the tilde mode with
g sinh(theta).  This is route (ii) of SURVEY.md section 8d item 4 (the reference builds the same Hamiltonian by hand
in transport/tests/test_spectral_function.py:16-48 and example/ttns/sbm_ft.py:69-75); it needs only ordinary
three-leg sites, at the price of twice the number of bath sites."""
import numpy as np

from ..utils import Quantity
from .basis import BasisSHO, BasisSimpleElectron
from .model import Model
from .op import Op


def thermofield_holstein(mol_list, j_matrix, temperature: Quantity, tilde_dims=None) -> Model:
    """Model with sites [e_0, (v_00, v~_00), (v_01, v~_01), ..., e_1, ...]; degrees of freedom: electron i, physical
    mode (i, k), tilde mode (i, k, "t").  ``tilde_dims``: levels kept for the tilde modes (default: same as the
    physical mode)."""
    beta = temperature.to_beta()
    n = len(mol_list)
    j_matrix = np.asarray(j_matrix)
    basis, ham = [], []
    for i, mol in enumerate(mol_list):
        basis.append(BasisSimpleElectron(i))
        for k, ph in enumerate(mol.ph_list):
            if not ph.is_simple:
                raise NotImplementedError("thermofield form of modes with a frequency change")
            nt = ph.n_phys_dim if tilde_dims is None else int(tilde_dims)
            basis.append(BasisSHO((i, k), ph.omega[0], ph.n_phys_dim))
            basis.append(BasisSHO((i, k, "t"), ph.omega[0], nt))
    for i in range(n):
        for j in range(n):
            f = mol_list[i].elocalex + mol_list[i].e0 if i == j else j_matrix[i, j]
            if f != 0:
                ham.append(Op(r"a^\dagger a", [i, j], f))
    for i, mol in enumerate(mol_list):
        for k, ph in enumerate(mol.ph_list):
            w = ph.omega[0]
            theta = np.arctanh(np.exp(-beta * w / 2.0)) if np.isfinite(beta) else 0.0
            c = -w ** 2 * ph.dis[1]                       # coupling of a^+ a x in the Holstein Hamiltonian
            ham += [Op("p^2", (i, k), 0.5), Op("x^2", (i, k), 0.5 * w ** 2)]
            ham += [Op("p^2", (i, k, "t"), -0.5), Op("x^2", (i, k, "t"), -0.5 * w ** 2)]
            ham.append(Op(r"a^\dagger a", i) * Op("x", (i, k)) * (c * np.cosh(theta)))
            if theta != 0.0:
                ham.append(Op(r"a^\dagger a", i) * Op("x", (i, k, "t")) * (c * np.sinh(theta)))
    model = Model(basis, ham)
    model.mol_num = n
    return model
