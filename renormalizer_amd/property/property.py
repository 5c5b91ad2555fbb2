"""``Property``: named operators whose expectation values a TdMpsJob records after every step
(renormalizer/property/property.py).  ``prop_mpos`` maps a name to an ``Mpo`` or to a list of ``Mpo`` (evaluated
with shared environments); the name "e_rdm" needs no operator and stores the electronic reduced density matrix."""
from typing import Dict, List

from ..mps.mpo import Mpo


class Property:
    # names whose operators are evaluated in the bra and in the ket separately when a bra-ket pair is given
    DIAGONAL_IN_PAIR = ("x", "x^2", "n")

    def __init__(self, prop_strs: List[str], prop_mpos: Dict[str, Mpo]):
        self.prop_strs = prop_strs
        self.prop_mpos = prop_mpos
        self.prop_res = {name: [] for name in prop_strs}

    def calc_properties(self, mps, mps_conj=None):
        """one value (or array) per name appended to ``prop_res`` (property.py:48-80)"""
        for name in self.prop_strs:
            if name == "e_rdm":
                self.prop_res[name].append(mps.calc_edof_rdm())
            elif name in self.prop_mpos:
                mpo = self.prop_mpos[name]
                if isinstance(mpo, Mpo):
                    self.prop_res[name].append(mps.expectation(mpo, mps_conj))
                elif isinstance(mpo, list):
                    assert mps_conj is None
                    self.prop_res[name].append(mps.expectations(mpo))
                else:
                    raise TypeError(f"property {name}: expected an Mpo or a list of Mpo, got {type(mpo)}")
            else:
                raise NotImplementedError(f"no operator registered for the property {name}")

    def calc_properties_braketpair(self, pair):
        """for a BraKetPair: <bra|O|ket> in general; for the coordinate / number operators ("x", "x^2", "n") the two
        diagonal values [<bra|O|bra>, <ket|O|ket>] (property.py:27-45)"""
        bra, ket = pair.bra_mps, pair.ket_mps
        for name in self.prop_strs:
            mpo = self.prop_mpos[name]
            if name in self.DIAGONAL_IN_PAIR:
                if isinstance(mpo, list):
                    self.prop_res[name].append([bra.expectations(mpo), ket.expectations(mpo)])
                else:
                    self.prop_res[name].append([bra.expectation(mpo, None), ket.expectation(mpo, None)])
            else:
                # the reference hands the bra over as is (not conjugated), property.py:44-45
                self.prop_res[name].append(ket.expectation(mpo, bra))
