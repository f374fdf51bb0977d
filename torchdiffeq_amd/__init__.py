"""torchdiffeq_amd — MI355X-native explicit Runge–Kutta ODE solvers behind the torchdiffeq API.

Drop-in surface for the reference's hot path (torchdiffeq/__init__.py:1-5, scope: SURVEY.md §8):
`odeint`, `odeint_adjoint` and the `SOLVERS` plugin table for dopri5 / dopri8 / rk4.  The state-sized
arithmetic runs in hand-written gfx950 HIP kernels (libtdeq_hip.so, C-ABI in include/tdeq_hip.h).
"""
from .odeint import SOLVERS, odeint
from .adjoint import odeint_adjoint

__version__ = "0.1.0"
__all__ = ["odeint", "odeint_adjoint", "SOLVERS"]
