"""Host drivers of the explicit Runge–Kutta solvers (dopri5, dopri8, rk4) over the HIP kernels.

The accept/reject loop stays on the host (one per process / shard): per trial step it issues S
`stage_combine` launches interleaved with the user's `func`, one fused `error_norm` launch, reads back
n_seg doubles, and runs the step controller in Python doubles — instead of the reference's ≈220 eager
ops and ≈19 device->host syncs per trial step.  Control flow and numerics follow

  RKAdaptiveStepsizeODESolver   torchdiffeq/_impl/rk_common.py:161-369
  _runge_kutta_step             rk_common.py:43-90
  _select_initial_step / _compute_error_ratio / _optimal_step_size   misc.py:36-95
  _interp_fit / _interp_evaluate  interp.py:1-48 (fused, evaluated lazily: only for requested outputs)
  FixedGridODESolver / RK4      solvers.py:52-181, fixed_grid.py:24-29, rk_common.py:110-118
  AdamsBashforth(Moulton)       fixed_adams.py:164-228

with time-like scalars (t0, t1, dt, rtol, ...) as host doubles instead of 0-dim device tensors.
"""
# r05: one module per solver family —
#   adaptive.py   RKAdaptiveStepsizeODESolver + dopri5 / dopri8 / tsit5 / bosh3 / fehlberg2 / adaptive_heun
#   fixed.py      FixedGridODESolver + euler / midpoint / heun2 / heun3 / rk4
#   multistep.py  Adams–Bashforth(–Moulton)   (out of the hot path's scope, frozen)
#   events.py     event mode of both families (mixins)
#   _common.py    controller scalars, step shadows
# Everything is re-exported here: `from torchdiffeq_amd.solvers import X` keeps working for every name it ever had.
from .._graph import (_AUTO_CAPTURE_AFTER_STEPS, _AUTO_MIN_GRID_STEPS, _GRAPH_AUTO_MAX_ELEMENTS,  # noqa: F401
                      _GRAPH_MODE_MAX_ELEMENTS, _CaptureFailed, _DtCell, _GraphStep, _capture, _graph_request,
                      _held_tensor_ptrs, _reusable_across_solves, _scalar_state, _side_effect_fingerprint, _side_stream,
                      _request_is_explicit, _stream_is_capturing, clear_graph_cache)
from ._common import (_nan_max, _nan_min, _clamp, _norm_value, _as_float, optimal_step_size, optimal_step_size_in, _StepShadow, _NoShadow, _NO_SHADOW)  # noqa: F401
from .adaptive import (AdaptiveHeunSolver, Bosh3Solver, Dopri5Solver, Dopri8Solver, Fehlberg2,  # noqa: F401
                       RKAdaptiveStepsizeODESolver, Tsit5Solver, _DenseRecord, _InitialStepShadow, _LockStep)
from .events import AdaptiveEvents, FixedGridEvents  # noqa: F401
from .fixed import (RK4, Euler, FixedGridODESolver, Heun2, Heun3, Midpoint, _host_times, _rk4_38_step,  # noqa: F401
                    _uniform_grid)
from .multistep import AdamsBashforth, AdamsBashforthMoulton  # noqa: F401

SOLVER_CLASSES = {"dopri8": Dopri8Solver, "dopri5": Dopri5Solver, "tsit5": Tsit5Solver, "bosh3": Bosh3Solver,
                  "fehlberg2": Fehlberg2, "adaptive_heun": AdaptiveHeunSolver, "euler": Euler,
                  "midpoint": Midpoint, "heun2": Heun2, "heun3": Heun3, "rk4": RK4,
                  "explicit_adams": AdamsBashforth, "implicit_adams": AdamsBashforthMoulton,
                  "fixed_adams": AdamsBashforthMoulton}
