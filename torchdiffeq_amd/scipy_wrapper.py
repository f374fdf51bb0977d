"""`scipy_solver` — the reference's SciPy bridge (torchdiffeq/_impl/scipy_wrapper.py:8-60) behind the same
plugin protocol.

Not part of the MI355X hot path and deliberately outside the C-ABI: SciPy's `solve_ivp` integrates on the
host in numpy; only `func` runs on the state's device.  It exists so that the `SOLVERS` table is a drop-in —
the same names, options (`solver`, `min_step`, `max_step`), no-gradient behaviour and return layout as the
reference.  Every solver call costs a device -> host -> device round trip of the state per evaluation; use the
HIP-backed methods for anything large.
"""
from __future__ import annotations

import numpy as np
import torch

from .misc import OdeFunc, handle_unused_kwargs


class ScipyWrapperODESolver:
    flat_state_native = True

    def __init__(self, func: OdeFunc, y0: torch.Tensor, rtol, atol, min_step=0, max_step=float("inf"),
                 solver="LSODA", **unused_kwargs):
        unused_kwargs.pop("norm", None)
        unused_kwargs.pop("grid_points", None)
        unused_kwargs.pop("eps", None)
        unused_kwargs.pop("dist_sync", None)
        unused_kwargs.pop("dist_replicated", None)
        unused_kwargs.pop("hip_graph", None)          # host-side solver: nothing to capture
        handle_unused_kwargs(self, unused_kwargs)
        del unused_kwargs
        if not isinstance(func, OdeFunc):
            raise TypeError("solver classes of torchdiffeq_amd take the wrapped func built by check_inputs")
        self.func = func
        self.layout = func.layout
        self.dtype = y0.dtype
        self.device = y0.device
        self.y0 = y0
        self.rtol, self.atol = rtol, atol
        self.min_step, self.max_step = min_step, max_step
        self.solver = solver

    @classmethod
    def valid_callbacks(cls):
        return set()

    # the flat state of this package is chunk-padded for tuple states; SciPy gets the components back to back
    def _compact(self, flat: torch.Tensor) -> np.ndarray:
        lay = self.layout
        if lay.n_seg == 1:
            return flat.detach().cpu().numpy().reshape(-1)
        return torch.cat([flat[off:off + n] for off, n in zip(lay.offsets, lay.numels)]).detach().cpu().numpy()

    def _expand(self, vec) -> torch.Tensor:
        """numpy [..., sum(numels)] -> device tensor [..., layout.total]."""
        lay = self.layout
        v = torch.as_tensor(np.ascontiguousarray(vec)).to(self.device, self.dtype)
        if lay.n_seg == 1:
            return v
        out = torch.zeros(*v.shape[:-1], lay.total, dtype=self.dtype, device=self.device)
        pos = 0
        for off, n in zip(lay.offsets, lay.numels):
            out[..., off:off + n] = v[..., pos:pos + n]
            pos += n
        return out

    def _tolerance(self, tol):
        """Scalar, or per-component values spread over the compact vector (misc.py:115-123)."""
        if isinstance(tol, torch.Tensor):
            tol = tol.tolist() if tol.dim() > 0 else float(tol)
        if isinstance(tol, (tuple, list)):
            return np.concatenate([np.full(n, float(v)) for v, n in zip(tol, self.layout.numels)])
        return tol

    def integrate(self, t: torch.Tensor) -> torch.Tensor:
        from scipy.integrate import solve_ivp
        y0 = self._compact(self.y0)
        if t.numel() == 1:
            return self._expand(y0)[None]
        t_np = t.detach().cpu().numpy()
        func = self.func

        def np_func(tv, yv):
            with torch.no_grad():
                f = func(torch.tensor(tv).to(self.device, self.dtype), self._expand(yv))
            return self._compact(f)

        sol = solve_ivp(np_func, t_span=[t_np.min(), t_np.max()], y0=y0, t_eval=t_np, method=self.solver,
                        rtol=self._tolerance(self.rtol), atol=self._tolerance(self.atol), min_step=self.min_step,
                        max_step=self.max_step)
        return self._expand(sol.y.T)
