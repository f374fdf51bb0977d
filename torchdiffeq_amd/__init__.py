"""torchdiffeq_amd — MI355X-native explicit Runge–Kutta ODE solvers behind the torchdiffeq API.

Drop-in surface for the reference's hot path (torchdiffeq/__init__.py:1-5, scope: SURVEY.md §8):
`odeint`, `odeint_adjoint`, `odeint_event`, `odeint_dense` and the `SOLVERS` plugin table with every method name
of the reference (explicit Runge–Kutta = the hot path; Adams multistep and implicit RK on the same kernels; the
SciPy bridge).  The state-sized arithmetic runs in hand-written gfx950 HIP kernels (libtdeq_hip.so, C-ABI in
include/tdeq_hip.h).
"""
from .odeint import SOLVERS, odeint, odeint_dense, odeint_event
from .adjoint import odeint_adjoint
from .solvers import clear_graph_cache
from ._fallback import HostPathWarning

__version__ = "0.1.0"
__all__ = ["odeint", "odeint_adjoint", "odeint_event", "odeint_dense", "SOLVERS", "clear_graph_cache",
           "HostPathWarning"]
