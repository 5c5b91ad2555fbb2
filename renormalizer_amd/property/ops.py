"""Ready-made operator sets for ``Property`` (renormalizer/property/ops.py)."""
import numpy as np

from ..model import Op
from ..mps.mpo import Mpo
from ..utils import Quantity


def e_ph_static_correlation(model, imol: int = 0, jph: int = 0, periodic: bool = False, name: str = "S"):
    r"""Electron-phonon static correlation of a Holstein polaron (J. Chem. Phys. 142, 174103; ops.py:8-69):
    S_(n,m,j) = <x_{m,j} a+_n a_n> / D_{m,j} for a fixed electron site n = ``imol`` (keys "S_n_m_j"), or, for a
    homogeneous periodic chain, summed over n at fixed distance (keys "S_distance_j")."""
    if model.scheme == 4:
        raise NotImplementedError
    nmols = model.mol_num

    def one(n, m):
        ph = model[m].ph_list[jph]
        return Mpo.intersite(model, {n: r"a^\dagger a"}, {(m, jph): r"b^\dagger+b"},
                             scale=Quantity(np.sqrt(1.0 / 2.0 / ph.omega[0]) / ph.dis[1]))

    out = {}
    if not periodic:
        for m in range(nmols):
            out["_".join([name, str(imol), str(m), str(jph)])] = one(imol, m)
        return out
    for dis in range(nmols):
        total = None
        for n in range(nmols):
            term = one(n, (n + dis) % nmols)
            total = term if total is None else total.add(term)
        out["_".join([name, str(dis), str(jph)])] = total
    return out


def x_average(model):
    """<x> of every vibrational degree of freedom"""
    return {"x": [Mpo(model, Op("x", dof)) for dof in model.v_dofs]}


def x_square_average(model):
    """<x^2> of every vibrational degree of freedom (the reference nests the list under a second key "x";
    here the list is the value, which is what ``Property`` consumes)"""
    return {"x^2": [Mpo(model, Op("x^2", dof)) for dof in model.v_dofs]}
