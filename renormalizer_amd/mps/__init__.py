from .backend import backend
from .mpo import Mpo
from .mps import Mps
