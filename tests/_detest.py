"""The non-stiff DETEST problem set (Hull, Enright, Fellen & Sedgwick 1972) — classes A (single equations), B (small
systems), C (moderate linear systems), D (two-body orbits, eccentricity 0.1 .. 0.9), E (second-order equations as
first-order systems) — as used by the reference's integration benchmark tests/DETEST (24 of its 25 problems; its
five-body problem C5 carries a mistyped initial value and is left out).  Each entry: name -> (field, y0), t in [0, 20],
fp64.  make_golden.gen_detest checks these restatements against the reference's own definitions before using them."""
import math

import torch


def _banded(n, diag, upper, lower):
    a = torch.zeros(n, n, dtype=torch.float64)
    idx = torch.arange(n)
    a[idx, idx] = torch.as_tensor(diag, dtype=torch.float64)
    if upper is not None:
        a[idx[:-1], idx[:-1] + 1] = torch.as_tensor(upper, dtype=torch.float64)
    if lower is not None:
        a[idx[1:], idx[1:] - 1] = torch.as_tensor(lower, dtype=torch.float64)
    return a


def _linear(a):
    return lambda t, y: torch.mv(a.to(y.device), y)


def _unit(n):
    y0 = torch.zeros(n, dtype=torch.float64)
    y0[0] = 1.0
    return y0


def _orbit(ecc):
    def field(t, y):
        r3 = (y[0] ** 2 + y[1] ** 2) ** 1.5
        return torch.stack([y[2], y[3], -y[0] / r3, -y[1] / r3])
    return field, torch.tensor([1 - ecc, 0.0, 0.0, math.sqrt((1 + ecc) / (1 - ecc))], dtype=torch.float64)


def _v(*xs):
    return torch.tensor(xs, dtype=torch.float64)


def problems():
    p = {}
    p["A1"] = (lambda t, y: -y, torch.tensor(1.0, dtype=torch.float64))
    p["A2"] = (lambda t, y: -y ** 3 / 2, torch.tensor(1.0, dtype=torch.float64))
    p["A3"] = (lambda t, y: y * torch.cos(t), torch.tensor(1.0, dtype=torch.float64))
    p["A4"] = (lambda t, y: y / 4 * (1 - y / 20), torch.tensor(1.0, dtype=torch.float64))
    p["A5"] = (lambda t, y: (y - t) / (y + t), torch.tensor(4.0, dtype=torch.float64))
    p["B1"] = (lambda t, y: torch.stack([2 * (y[0] - y[0] * y[1]), -(y[1] - y[0] * y[1])]), _v(1.0, 3.0))
    p["B2"] = (_linear(_banded(3, [-1.0, -2.0, -1.0], [1.0, 1.0], [1.0, 1.0])), _v(2.0, 0.0, 1.0))
    p["B3"] = (lambda t, y: torch.stack([-y[0], y[0] - y[1] * y[1], y[1] * y[1]]), _v(1.0, 0.0, 0.0))

    def b4(t, y):
        a = torch.sqrt(y[0] * y[0] + y[1] * y[1])
        return torch.stack([-y[1] - y[0] * y[2] / a, y[0] - y[1] * y[2] / a, y[0] / a])
    p["B4"] = (b4, _v(3.0, 0.0, 0.0))
    p["B5"] = (lambda t, y: torch.stack([y[1] * y[2], -y[0] * y[2], -0.51 * y[0] * y[1]]), _v(0.0, 1.0, 1.0))
    # radioactive decay chains: y_1' = -k_1 y_1, y_i' = k_{i-1} y_{i-1} - k_i y_i, the last species only accumulates
    p["C1"] = (_linear(_banded(10, [-1.0] * 9 + [0.0], None, [1.0] * 9)), _unit(10))
    k = [float(i) for i in range(1, 10)]
    p["C2"] = (_linear(_banded(10, [-v for v in k] + [0.0], None, k)), _unit(10))
    # discretised heat equation: tridiagonal (1, -2, 1)
    p["C3"] = (_linear(_banded(10, [-2.0] * 10, [1.0] * 9, [1.0] * 9)), _unit(10))
    p["C4"] = (_linear(_banded(51, [-2.0] * 51, [1.0] * 50, [1.0] * 50)), _unit(51))
    for i, ecc in enumerate([0.1, 0.3, 0.5, 0.7, 0.9], start=1):
        p[f"D{i}"] = _orbit(ecc)
    p["E1"] = (lambda t, y: torch.stack([y[1], -(y[1] / (t + 1) + (1 - 0.25 / (t + 1) ** 2) * y[0])]),
               _v(0.671396707141803, 0.0954005144474744))
    p["E2"] = (lambda t, y: torch.stack([y[1], (1 - y[0] ** 2) * y[1] - y[0]]), _v(2.0, 0.0))
    p["E3"] = (lambda t, y: torch.stack([y[1], y[0] ** 3 / 6 - y[0] + 2 * torch.sin(2.78535 * t)]), _v(0.0, 0.0))
    p["E4"] = (lambda t, y: torch.stack([y[1], 0.32 - 0.4 * y[1] ** 2]), _v(30.0, 0.0))
    p["E5"] = (lambda t, y: torch.stack([y[1], torch.sqrt(1 + y[1] ** 2) / (25 - t)]), _v(0.0, 0.0))
    return p
