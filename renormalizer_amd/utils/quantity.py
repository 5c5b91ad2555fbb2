"""A number with a unit, convertible to atomic units (API of renormalizer/utils/quantity.py)."""
import math

from . import constant

_PER_AU = {"meV": constant.au2ev * 1e3, "eV": constant.au2ev, "cm^{-1}": constant.au2cm, "cm-1": constant.au2cm,
           "K": constant.au2K, "a.u.": 1.0, "au": 1.0, "fs": constant.au2fs}
_PER_AU.update({k.lower(): v for k, v in list(_PER_AU.items())})


class Quantity:
    def __init__(self, value, unit="a.u."):
        if unit not in _PER_AU:
            raise ValueError(f"Unit not in {set(_PER_AU)}, got {unit}.")
        self.value = float(value)
        self.unit = unit

    def as_au(self):
        return self.value / _PER_AU[self.unit]

    def as_unit(self, unit):
        return Quantity(self.as_au() * _PER_AU[unit], unit)

    def to_beta(self):
        return math.inf if self.value == 0 else 1.0 / self.as_au()

    def __neg__(self):
        return Quantity(-self.value, self.unit)

    def __add__(self, other):
        return Quantity(self.as_au() + other.as_au())

    def __sub__(self, other):
        return Quantity(self.as_au() - other.as_au())

    def __mul__(self, other):
        if isinstance(other, Quantity):
            raise TypeError("Quantity * Quantity is not defined")
        return Quantity(self.as_au() * other)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return Quantity(self.as_au() / other)

    def __eq__(self, other):
        if hasattr(other, "as_au"):
            return self.as_au() == other.as_au()
        if other == 0:
            return self.value == 0
        raise TypeError(f"Quantity can only compare with Quantity or 0, not {type(other)}")

    def __ne__(self, other):
        return not self == other

    def __float__(self):
        return self.as_au()

    def __repr__(self):
        return f"Quantity({self.value}, {self.unit!r})"
