from .quantity import Quantity
from . import constant
from .configs import CompressConfig, CompressCriteria, OptimizeConfig, EvolveConfig, EvolveMethod, OFS
