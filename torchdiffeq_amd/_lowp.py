"""bfloat16 / float16 states on a ROCm device: the host-driven step on the HIP kernels (csrc/tdeq_kernels_lp.hpp).

The reference integrates a reduced-precision state in its own precision (torchdiffeq/_impl/misc.py:185-187,
rk_common.py:61-65: every time-like scalar is cast to `y0.abs().dtype`).  r04 served such states with the torch-op host
path on whatever device they lived (`_fallback.LowPrecisionHostKernels`: ~220 ATen launches and dense [N, row] product
tensors per trial step).  `LowPrecisionHipKernels` keeps that class's interface and host scalars (`_scalars.py`) and
replaces every state-sized operation of the explicit Runge–Kutta path — stage combines, error norm, initial-step norms,
dense output, the fixed-grid stages — with ONE fused HIP launch whose per-element arithmetic is the torch-op sequence
with the same roundings (each ATen op = float32 operation + one rounding to the storage type; a tableau row's
`torch.sum` = rounded products accumulated in float32, rounded once).  What the kernels do not cover (the Adams
methods' predictor / corrector, segment packing, the backward-pass helpers `scale_many` / `multi_dot`) is inherited
from the torch-op class — same numbers, more launches.

Selection: `_native.get_kernels` for a bf16 / fp16 state on a `cuda` device; a missing libtdeq_hip.so raises there.
The loop is the fp32 one: the norm launch's finalize step also runs the step controller on the device in the state's type
and the next trial step's first stage + func evaluation are enqueued before the decision is read back (look-ahead);
adaptive trial steps can be captured and replayed as hipGraphs like fp32 ones (`hip_graph`).  What reduced precision does
not get: the fused error split and carried partial sums (a row would be rounded twice) and captured fixed grids.
"""
from __future__ import annotations

import ctypes
import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _fallback
from ._fallback import HostPlan, LowPrecisionHostKernels
from ._native import _check, dtype_code


class LowPlan(HostPlan):
    """`HostPlan` (what the inherited torch-op methods read) + the native plan the norm kernels write into."""

    def __init__(self, segments, total, chunk, native):
        super().__init__(segments, total, chunk)
        self.hip = native
        self.pending = None        # (kind, dtype) of the norm launch whose words have not been read yet

    @property
    def ctrl_dev(self):            # the device-resident controller words {accept, sign * T(dt'), t0', dt'} (captured steps)
        return self.hip.ctrl_dev

    @property
    def native_segs(self):         # the ctypes segment table (chunk_start, numel, rtol, atol) the kernels were given
        return self.hip.segs


# (no `_no_grad_methods` here: the methods below only launch kernels through ctypes — entering a no_grad context per call
#  costs ~2.5 us of host time, 20 us per trial step of a loop that is host-bound at 16-bit sizes; the inherited torch-op
#  methods keep their wrappers)
class LowPrecisionHipKernels(LowPrecisionHostKernels):
    name = "hip-low"
    device_controller = True     # tdeq_error_norm_partial_ctrl on the WHOLE error row + tdeq_stage_combine_sel (look-ahead)
    whole_row_controller = True  # ... i.e. without the fused error split the fp32 / fp64 controller launch continues
    literal_row_sums = False     # the kernels skip structural zeros of a row, like the fp32 / fp64 ones (docs/LAB_NOTEBOOK.md §8)
    split_row_sums = False       # ... but a row is never split over two launches (it is rounded ONCE): no fused error split
    literal_norms = True         # plan.rms0 / rms1 / abs0 hold the norm values in the state's type (read_norms)

    def __init__(self, hip):
        self._hip = hip          # _native.HipKernels: ctypes wrappers + stream / read-back machinery

    def make_plan(self, segments, total, chunk, device) -> LowPlan:
        return LowPlan(segments, total, chunk, self._hip.make_plan(segments, total, chunk, device))

    # -- stage combines (tdeq_stage_combine*, dtype TDEQ_BF16 / TDEQ_F16) -------------------------------------------
    def stage_combine(self, out, y0, ks, coefs, dt: float) -> None:
        self._hip.stage_combine(out, y0, ks, coefs, dt)

    def stage_combine_fill(self, out, y0, ks, coefs, dt: float, fill_dst, fill_vals) -> None:
        if fill_dst.dtype == y0.dtype and len(ks) <= 2:
            self._hip.stage_combine_fill(out, y0, ks, coefs, dt, fill_dst, fill_vals)
        else:       # (stage times kept in another type: the torch-op fill)
            self._hip.stage_combine(out, y0, ks, coefs, dt)
            self.fill_scalars(fill_dst, fill_vals)

    def stage_combine_err(self, out, err_out, y0, ks, coefs, err_coefs, dt: float) -> None:
        self._hip.stage_combine_err(out, err_out, y0, ks, coefs, err_coefs, dt)

    def stage_combine_multi(self, outs, rows, y0, acc_in, ks, dt: float, events=None) -> None:
        raise NotImplementedError("carried partial sums re-associate a row: not for reduced-precision states")

    # -- norms ----------------------------------------------------------------------------------------------------
    def error_norm(self, plan: LowPlan, y0, y1, ks, coefs, dt: float, scaled_out=None) -> None:
        hip, p = self._hip, plan.hip
        ptrs, cf, n = hip._terms(ks, coefs)
        dev = p.segs_dev.data_ptr() if p.segs_dev is not None else None
        hip._arm(p, 1)
        _check(hip.lib.tdeq_error_norm(None if scaled_out is None else scaled_out.data_ptr(), y0.data_ptr(),
                                       y1.data_ptr(), ptrs, cf, n, dt, p.segs, dev, p.n_seg, p.chunk, p.n_chunks,
                                       p.out_ptr, p.bad_ptr, p.workspace.data_ptr(), p.workspace_bytes,
                                       dtype_code(y0.dtype), hip._stream()), "tdeq_error_norm")
        plan.pending = ("err", y0.dtype)

    def stage_combine_dev(self, out, err_out, y0, ks, coefs, err_coefs, plan: LowPlan) -> None:
        """`stage_combine` with the step size read on the device from the controller words (captured steps)."""
        assert err_out is None, "a 16-bit row is rounded once: no partial error sum"
        self._hip.stage_combine_dev(out, None, y0, ks, coefs, None, plan.hip)

    def arm_readback(self, plan: LowPlan) -> None:
        self._hip.arm_readback(plan.hip)

    def error_norm_ctrl(self, plan: LowPlan, y0, y1, ks, coefs, dt: float, ctrl, next_times, state_in_dev: bool = False) -> None:
        """`error_norm` whose finalize step also runs the step controller on the device (tdeq_error_norm_partial_ctrl
        with err_partial = NULL: the whole error row in one launch): accept flag, next step size and the next trial
        step's stage times, all in the state's type — read with `read_ctrl`."""
        hip, p = self._hip, plan.hip
        ptrs, cf, n = hip._terms(ks, coefs)
        hip._arm(p, 1, ctrl=True)
        _check(hip.lib.tdeq_error_norm_partial_ctrl(
            None, y0.data_ptr(), y1.data_ptr(), ptrs, cf, n, dt, p.segs,
            p.segs_dev.data_ptr() if p.segs_dev is not None else None, p.n_seg, p.chunk, p.n_chunks, p.out_ptr, p.bad_ptr,
            ctypes.byref(ctrl), p.ctrl_ptr, p.ctrl_dev.data_ptr(), next_times.data_ptr(), 1 if state_in_dev else 0,
            None,       # copy_last_k (ABI 21): fp32 / fp64 states only
            p.workspace.data_ptr(), p.workspace_bytes, dtype_code(y0.dtype), hip._stream()), "tdeq_error_norm_partial_ctrl")
        plan.pending = None

    def read_ctrl(self, plan: LowPlan):
        return self._hip.read_ctrl(plan.hip)

    def stage_combine_sel(self, out, y_acc, f_acc, y_rej, f_rej, coef: float, plan: LowPlan) -> None:
        self._hip.stage_combine_sel(out, y_acc, f_acc, y_rej, f_rej, coef, plan.hip)

    def error_norm_partial(self, plan, err_partial, y0, y1, ks, coefs, dt: float) -> None:
        raise NotImplementedError("a continued error sum is rounded twice: not for reduced-precision states")

    def error_scaled(self, plan, out, y0, y1, ks, coefs, dt: float) -> None:
        self.error_norm(plan, y0, y1, ks, coefs, dt, scaled_out=out)

    def init_norms(self, plan: LowPlan, mode: int, a, b, yscale) -> None:
        self._hip.init_norms(plan.hip, mode, a, b, yscale)
        plan.pending = ("init0" if mode == 0 else "init1", yscale.dtype)

    def init_scaled(self, plan: LowPlan, mode: int, a, b, yscale, out0, out1=None) -> None:
        self._hip.init_scaled(plan.hip, mode, a, b, yscale, out0, out1)

    def read_norms(self, plan: LowPlan):
        """The last norm launch's words -> the norm values in the state's type: sqrt(mean(|x|^2)) as ATen evaluates it
        for a reduced-precision tensor (misc.py:22-23 — squares rounded by the kernel, float32 sum / n rounded once, the
        `sqrt` rounded; `_scalars.BFloat16Scalar` / `Float16Scalar` arithmetic).  A segment of ONE element arrives as
        |x| itself (`abs0` / `abs1`: adjoint.py:250 takes `t.abs()`); its square is formed here with the same rounding."""
        if plan.pending is not None:
            kind, dtype = plan.pending
            plan.pending = None
            s0, s1, bad = self._hip.read_norms(plan.hip)
            T = _fallback.real_np_dtype(dtype)
            plan.bad = list(bad)

            def values(total: float, n: int):
                """(fp64 sum of the rounded squares, rms in the state's type, |x| of a one-element segment)"""
                if n == 0:
                    return 0.0, math.nan, math.nan
                with np.errstate(all="ignore"):
                    one = math.nan
                    if n == 1:
                        a = T(np.float32(total))
                        one, total = float(a), float(a * a)
                    # ATen's mean of a reduced-precision tensor: float32 sum / n, rounded ONCE (the sum itself never is:
                    # a float16 sum of squares would overflow the type), then sqrt in the type
                    return total, float(T(np.float32(total) / np.float32(n)) ** 0.5), one
            for s, n in enumerate(plan.numels):
                plan.sums0[s], plan.rms0[s], plan.abs0[s] = values(s0[s], n)
                if kind == "init0":
                    plan.sums1[s], plan.rms1[s], plan.abs1[s] = values(s1[s], n)
        return list(plan.sums0), list(plan.sums1), list(plan.bad)

    # -- dense output ------------------------------------------------------------------------------------------------
    def dense_eval(self, out, y0, y1, f0, f1, ks, coefs, dt: float, x: float) -> None:
        self._hip.dense_eval(out, y0, y1, f0, f1, ks, coefs, dt, x)

    def dense_eval_multi(self, out_rows, y0, y1, f0, f1, ks, coefs, dt: float, xs: Sequence[float]) -> None:
        self._hip.dense_eval_multi(out_rows, y0, y1, f0, f1, ks, coefs, dt, xs)

    def interp_fit(self, coeffs, y0, y1, f0, f1, ks, coefs, dt: float) -> None:
        self._hip.interp_fit(coeffs, y0, y1, f0, f1, ks, coefs, dt)

    # -- fixed-grid steps --------------------------------------------------------------------------------------------
    def rk4_stage(self, stage: int, out, y0, k1, k2, k3, k4, dt: float) -> None:
        self._hip.rk4_stage(stage, out, y0, k1, k2, k3, k4, dt)

    def lerp(self, out, y0, y1, slope: float) -> None:
        self._hip.lerp(out, y0, y1, slope)

    def fixed_stage(self, mode: int, out, y0, ks, ws, dt: float) -> None:
        if len(ks) <= 4:
            self._hip.fixed_stage(mode, out, y0, ks, ws, dt)
        else:
            super().fixed_stage(mode, out, y0, ks, ws, dt)

    def scaled_add(self, out, y0, k, scalar: float) -> None:
        # y0 + k * s with s taken at float32 (second operand) = the generic fixed stage with dt = 1 and one weight:
        # y0 + fl(fl(k * s) * 1) — the multiplication by the exactly representable 1 rounds nothing
        self._hip.fixed_stage(0, out, y0, [k], (scalar,), 1.0)

    def weighted_sum(self, out, xs, ws) -> None:
        if len(xs) <= 8:
            self._hip.weighted_sum(out, xs, ws)
        else:
            super().weighted_sum(out, xs, ws)
