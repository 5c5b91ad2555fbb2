"""Iterative solvers on device-resident vectors: Lanczos exponential (krylov.py), Davidson (davidson.py)."""
