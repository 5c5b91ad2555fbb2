from .mpo import Mpo
