from .op import Op, OpSum
from .basis import (BasisSet, BasisSHO, BasisSimpleElectron, BasisHalfSpin, BasisMultiElectron,
                    BasisMultiElectronVac)
from .phonon import Phonon, Mol
from .model import (Model, HolsteinModel, SpinBosonModel, TI1DModel, construct_j_matrix, load_from_dict,
                    heisenberg_ops)
from .thermofield import thermofield_holstein
from . import h_qc
