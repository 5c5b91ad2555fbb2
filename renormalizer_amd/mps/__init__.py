from .backend import backend
from .mpo import Mpo
from .mps import Mps
from .mpdm import MpDm
from .thermalprop import thermal_state
