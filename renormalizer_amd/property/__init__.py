"""User-defined observables recorded by the time-stepping jobs (counterpart of renormalizer/property)."""
from .property import Property
from . import ops
