"""Butcher tableaux of the explicit Runge-Kutta propagators selectable through ``EvolveConfig(rk_solver=...)``
(the solver names and the (tableau, stage, order) attributes of renormalizer/utils/rk.py:14-202), and the Taylor
coefficients of the formal propagator.  The tableaux are the published ones (Hairer, Norsett, Wanner I, II.1-II.5;
Fehlberg 1969; Cash & Karp 1990), written as exact fractions."""
from fractions import Fraction as F
import math

import numpy as np


def _rows(*rows):
    n = len(rows)
    a = np.zeros((n, n))
    for i, r in enumerate(rows):
        for j, v in enumerate(r):
            a[i, j] = float(v)
    return a


def _two_stage(alpha):
    # the one-parameter family of second-order methods: alpha = 1 midpoint, 1/2 Heun, 2/3 Ralston
    return _rows([], [alpha]), [[1 - F(1, 2) / alpha, F(1, 2) / alpha]], [0, alpha], (2,)


_FEHLBERG_A = _rows([], [F(1, 4)], [F(3, 32), F(9, 32)], [F(1932, 2197), F(-7200, 2197), F(7296, 2197)],
                    [F(439, 216), -8, F(3680, 513), F(-845, 4104)],
                    [F(-8, 27), 2, F(-3544, 2565), F(1859, 4104), F(-11, 40)])
_FEHLBERG_C = [0, F(1, 4), F(3, 8), F(12, 13), 1, F(1, 2)]
_FEHLBERG_B5 = [F(16, 135), 0, F(6656, 12825), F(28561, 56430), F(-9, 50), F(2, 55)]
_FEHLBERG_B4 = [F(25, 216), 0, F(1408, 2565), F(2197, 4104), F(-1, 5), 0]

_TABLEAUX = {
    "Forward_Euler": lambda: (_rows([]), [[1]], [0], (1,)),
    "midpoint_RK2": lambda: _two_stage(F(1)),
    "Heun_RK2": lambda: _two_stage(F(1, 2)),
    "Ralston_RK2": lambda: _two_stage(F(2, 3)),
    "Kutta_RK3": lambda: (_rows([], [F(1, 2)], [-1, 2]), [[F(1, 6), F(2, 3), F(1, 6)]], [0, F(1, 2), 1], (3,)),
    "C_RK4": lambda: (_rows([], [F(1, 2)], [0, F(1, 2)], [0, 0, 1]), [[F(1, 6), F(1, 3), F(1, 3), F(1, 6)]],
                      [0, F(1, 2), F(1, 2), 1], (4,)),
    "38rule_RK4": lambda: (_rows([], [F(1, 3)], [F(-1, 3), 1], [1, -1, 1]), [[F(1, 8), F(3, 8), F(3, 8), F(1, 8)]],
                           [0, F(1, 3), F(2, 3), 1], (4,)),
    "Fehlberg5": lambda: (_FEHLBERG_A, [_FEHLBERG_B5], _FEHLBERG_C, (5,)),
    "RKF45": lambda: (_FEHLBERG_A, [_FEHLBERG_B5, _FEHLBERG_B4], _FEHLBERG_C, (5, 4)),
    "Cash-Karp45": lambda: (
        _rows([], [F(1, 5)], [F(3, 40), F(9, 40)], [F(3, 10), F(-9, 10), F(6, 5)],
              [F(-11, 54), F(5, 2), F(-70, 27), F(35, 27)],
              [F(1631, 55296), F(175, 512), F(575, 13824), F(44275, 110592), F(253, 4096)]),
        [[F(37, 378), 0, F(250, 621), F(125, 594), 0, F(512, 1771)],
         [F(2825, 27648), 0, F(18575, 48384), F(13525, 55296), F(277, 14336), F(1, 4)]],
        [0, F(1, 5), F(3, 10), F(3, 5), 1, F(7, 8)], (5, 4)),
}

method_list = list(_TABLEAUX)


class RungeKutta:
    """``tableau = [a (stage, stage), b (n_orders, stage), c (stage,)]``, ``stage``, ``order`` (one entry per row of b;
    embedded pairs carry the higher order first)."""

    def __init__(self, method="C_RK4"):
        if method not in _TABLEAUX:
            raise ValueError(f"unknown Runge-Kutta method {method}; known: {method_list}")
        self.method = method
        a, b, c, order = _TABLEAUX[method]()
        b = np.array([[float(v) for v in row] for row in b])
        c = np.array([float(v) for v in c])
        self.tableau = [a, b, c]
        self.stage = len(c)
        self.order = order

    def runge_kutta_ti_coefficient(self):
        """Coefficients d_k of y(t + dt) = sum_k d_k (f dt)^k y(t) when f is a constant linear map
        (utils/rk.py:204-243): the elementary weights b . A^(k-1) . 1."""
        a, b, _ = self.tableau
        coeff = np.zeros((b.shape[0], self.stage + 1))
        coeff[:, 0] = 1.0
        power = np.ones(self.stage)
        for k in range(1, self.stage + 1):
            coeff[:, k] = b @ power
            power = a @ power
        return coeff if b.shape[0] > 1 else coeff[0]


class TaylorExpansion:
    """1/k! for k <= order: the Taylor propagator of a time-independent Hamiltonian (utils/rk.py:27-34)"""

    def __init__(self, order):
        self.order = order
        self.coeff = np.array([1.0 / math.factorial(k) for k in range(order + 1)])
