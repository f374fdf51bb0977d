"""ctypes binding of oracle/_build/librk_oracle.so with the same tensor-level interface as
torchdiffeq_amd._native.HipKernels, operating on CPU torch tensors.

TEST INFRASTRUCTURE ONLY (see rk_oracle.c).  Two uses:
  * kernel parity: tests call the same method on `HipKernels` (GPU) and `OracleKernels` (CPU) with the
    same seeded inputs and compare;
  * host-logic tests without a GPU: tests monkeypatch `torchdiffeq_amd._native.get_kernels` to return an
    `OracleKernels`, which lets the product's solver / adjoint / sharding control flow run on CPU
    tensors.  The product itself never imports this module.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import List, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librk_oracle.so")

_c_void_pp = ctypes.POINTER(ctypes.c_void_p)
_c_double_p = ctypes.POINTER(ctypes.c_double)


class MultiOut(ctypes.Structure):
    _fields_ = [("out", ctypes.c_void_p), ("coef", ctypes.c_double * 14), ("mask", ctypes.c_uint32),
                ("add_y0", ctypes.c_int32)]


class Segment(ctypes.Structure):
    _fields_ = [("chunk_start", ctypes.c_int64), ("numel", ctypes.c_int64),
                ("rtol", ctypes.c_double), ("atol", ctypes.c_double)]


def build(force: bool = False) -> str:
    """Compile rk_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "rk_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/librk_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def load() -> ctypes.CDLL:
    lib = ctypes.CDLL(build())
    V, I, I64, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
    sigs = {
        "oracle_abi_version": [],
        "oracle_stage_combine": [V, V, _c_void_pp, _c_double_p, I, D, I64, I],
        "oracle_error_norm": [V, V, V, _c_void_pp, _c_double_p, I, D, ctypes.POINTER(Segment), I, I64, I64,
                              V, V, I],
        "oracle_init_norms": [I, V, V, V, ctypes.POINTER(Segment), I, I64, I64, V, V, I],
        "oracle_init_scaled": [I, V, V, V, ctypes.POINTER(Segment), I, I64, I64, V, V, I],
        "oracle_dense_eval": [V, V, V, V, V, _c_void_pp, _c_double_p, I, D, D, I64, I],
        "oracle_interp_fit": [V, V, V, V, V, _c_void_pp, _c_double_p, I, D, I64, I],
        "oracle_rk4_38_stage": [I, V, V, V, V, V, V, D, I64, I],
        "oracle_lerp": [V, V, V, D, I64, I],
        "oracle_fixed_stage": [I, V, V, _c_void_pp, _c_double_p, I, D, I64, I],
        "oracle_weighted_sum": [V, _c_void_pp, _c_double_p, I, I64, I],
        "oracle_stage_combine_err": [V, V, V, _c_void_pp, _c_double_p, _c_double_p, I, D, I64, I],
        "oracle_stage_combine_multi": [ctypes.POINTER(MultiOut), I, V, V, _c_void_pp, I, D, I64, I],
        "oracle_error_norm_partial": [V, V, V, _c_void_pp, _c_double_p, I, D, ctypes.POINTER(Segment), I, I64, I64,
                                      V, V, I],
        "oracle_error_norm_partial_ctrl": [V, V, V, _c_void_pp, _c_double_p, I, D, ctypes.POINTER(Segment), I, I64, I64,
                                           V, V, V, V, V, V, I],
        "oracle_step_controller": [ctypes.POINTER(Segment), I, V, V, V, V, V, I],
        "oracle_stage_combine_sel": [V, V, V, V, V, D, V, I64, I],
        "oracle_pack_segments": [V, _c_void_pp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                 _c_double_p, I, I64, I64, I],
        "oracle_scale_many": [_c_void_pp, V, _c_double_p, I, I64, I],
        "oracle_multi_dot": [V, _c_void_pp, I, I64, V, I],
        "oracle_adams_predict": [V, V, V, V, _c_void_pp, _c_double_p, _c_double_p, I, D, I64, I],
        "oracle_adams_correct": [V, V, V, V, V, V, D, I, ctypes.POINTER(Segment), I, I64, I64, I64, V, V, I],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = argtypes
    return lib


def _code(dtype: torch.dtype) -> int:
    return {torch.float32: 0, torch.float64: 1}[dtype]


def _ok(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}")


class OraclePlan:
    def __init__(self, segments: Sequence[Tuple[int, int, float, float]], total: int, chunk: int):
        self.chunk = chunk
        self.n_seg = len(segments)
        self.numels = [int(s[1]) for s in segments]
        self.n_chunks = max(1, math.ceil(total / chunk))
        arr = (Segment * self.n_seg)()
        for i, (off, numel, rtol, atol) in enumerate(segments):
            assert off % chunk == 0
            arr[i] = Segment(off // chunk, numel, float(rtol), float(atol))
        self.segs = arr
        self.out = torch.zeros(3 * self.n_seg + 4, dtype=torch.float64)
        self.out_ptr = self.out.data_ptr()
        self.bad_ptr = self.out_ptr + 16 * self.n_seg
        self.ctrl_ptr = self.out_ptr + 24 * self.n_seg
        self.ctrl_dev = torch.zeros(2, dtype=torch.float64)


class OracleKernels:
    """CPU twin of HipKernels (same method names and argument meaning)."""
    name = "oracle"

    def __init__(self):
        self.lib = load()

    @staticmethod
    def _terms(ks, coefs):
        n = len(ks)
        for k in ks:
            assert k.device.type == "cpu" and k.is_contiguous()
        return (ctypes.c_void_p * n)(*[k.data_ptr() for k in ks]), (ctypes.c_double * n)(*coefs), n

    def make_plan(self, segments, total, chunk, device) -> OraclePlan:
        return OraclePlan(segments, total, chunk)

    def stage_combine(self, out, y0, ks, coefs, dt):
        ptrs, cf, n = self._terms(ks, coefs)
        _ok(self.lib.oracle_stage_combine(out.data_ptr(), y0.data_ptr(), ptrs, cf, n, dt, y0.numel(),
                                          _code(y0.dtype)), "oracle_stage_combine")

    def stage_combine_fill(self, out, y0, ks, coefs, dt, fill_dst, fill_vals):
        """Host twin of tdeq_stage_combine_fill = stage_combine + fill_scalars."""
        self.stage_combine(out, y0, ks, coefs, dt)
        self.fill_scalars(fill_dst, fill_vals)

    def error_norm(self, plan, y0, y1, ks, coefs, dt, scaled_out=None):
        ptrs, cf, n = self._terms(ks, coefs)
        so = None if scaled_out is None else scaled_out.data_ptr()
        _ok(self.lib.oracle_error_norm(so, y0.data_ptr(), y1.data_ptr(), ptrs, cf, n, dt, plan.segs, plan.n_seg,
                                       plan.chunk, plan.n_chunks, plan.out_ptr, plan.bad_ptr, _code(y0.dtype)),
            "oracle_error_norm")

    def stage_combine_err(self, out, err_out, y0, ks, coefs, err_coefs, dt):
        ptrs, cf, n = self._terms(ks, coefs)
        ef = (ctypes.c_double * n)(*err_coefs)
        _ok(self.lib.oracle_stage_combine_err(out.data_ptr(), err_out.data_ptr(), y0.data_ptr(), ptrs, cf, ef, n, dt,
                                              y0.numel(), _code(y0.dtype)), "oracle_stage_combine_err")

    def stage_combine_multi(self, outs, rows, y0, acc_in, ks, dt):
        n = len(ks)
        for k in ks:
            assert k.device.type == "cpu" and k.is_contiguous()
        ptrs = (ctypes.c_void_p * n)(*[k.data_ptr() for k in ks])
        spec = (MultiOut * len(outs))()
        for o, (t, (coefs, mask, add_y0)) in enumerate(zip(outs, rows)):
            spec[o].out = t.data_ptr()
            for j, c in enumerate(coefs):
                spec[o].coef[j] = c
            spec[o].mask, spec[o].add_y0 = mask, 1 if add_y0 else 0
        _ok(self.lib.oracle_stage_combine_multi(spec, len(outs), y0.data_ptr(),
                                                None if acc_in is None else acc_in.data_ptr(), ptrs, n, dt, y0.numel(),
                                                _code(y0.dtype)), "oracle_stage_combine_multi")

    def error_norm_partial(self, plan, err_partial, y0, y1, ks, coefs, dt):
        n = len(ks)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[k.data_ptr() for k in ks])
        cf = (ctypes.c_double * max(n, 1))(*coefs)
        _ok(self.lib.oracle_error_norm_partial(err_partial.data_ptr(), y0.data_ptr(), y1.data_ptr(), ptrs, cf, n, dt,
                                               plan.segs, plan.n_seg, plan.chunk, plan.n_chunks, plan.out_ptr,
                                               plan.bad_ptr, _code(y0.dtype)), "oracle_error_norm_partial")

    def error_scaled(self, plan, out, y0, y1, ks, coefs, dt):
        self.error_norm(plan, y0, y1, ks, coefs, dt, scaled_out=out)

    def init_norms(self, plan, mode, a, b, yscale):
        _ok(self.lib.oracle_init_norms(mode, a.data_ptr(), b.data_ptr(), yscale.data_ptr(), plan.segs, plan.n_seg,
                                       plan.chunk, plan.n_chunks, plan.out_ptr, plan.bad_ptr, _code(yscale.dtype)),
            "oracle_init_norms")

    def init_scaled(self, plan, mode, a, b, yscale, out0, out1=None):
        _ok(self.lib.oracle_init_scaled(mode, a.data_ptr(), b.data_ptr(), yscale.data_ptr(), plan.segs, plan.n_seg,
                                        plan.chunk, plan.n_chunks, out0.data_ptr(),
                                        None if out1 is None else out1.data_ptr(), _code(yscale.dtype)),
            "oracle_init_scaled")

    def read_norms(self, plan) -> Tuple[List[float], List[float], List[float]]:
        v = plan.out.tolist()
        n = plan.n_seg
        return v[:n], v[n:2 * n], v[2 * n:3 * n]

    def error_norm_partial_ctrl(self, plan, err_partial, y0, y1, ks, coefs, dt, ctrl, next_times):
        """Host twin of tdeq_error_norm_partial_ctrl; `ctrl` is a ctypes struct laid out as oracle_step_ctrl."""
        n = len(ks)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[k.data_ptr() for k in ks])
        cf = (ctypes.c_double * max(n, 1))(*coefs)
        _ok(self.lib.oracle_error_norm_partial_ctrl(err_partial.data_ptr(), y0.data_ptr(), y1.data_ptr(), ptrs, cf, n,
                                                    dt, plan.segs, plan.n_seg, plan.chunk, plan.n_chunks, plan.out_ptr,
                                                    plan.bad_ptr, ctypes.addressof(ctrl), plan.ctrl_ptr,
                                                    plan.ctrl_dev.data_ptr(), next_times.data_ptr(), _code(y0.dtype)),
            "oracle_error_norm_partial_ctrl")

    def step_controller(self, plan, sumsq, ctrl, next_times, dtype):
        """The controller alone on given per-segment sums -> (out_ctrl[4], ctrl_dev[2]) as lists."""
        ss = (ctypes.c_double * plan.n_seg)(*sumsq)
        _ok(self.lib.oracle_step_controller(plan.segs, plan.n_seg, ctypes.addressof(ss), ctypes.addressof(ctrl),
                                            plan.ctrl_ptr, plan.ctrl_dev.data_ptr(), next_times.data_ptr(),
                                            _code(dtype)), "oracle_step_controller")
        return plan.out.tolist()[3 * plan.n_seg:], plan.ctrl_dev.tolist()

    def read_ctrl(self, plan):
        v = plan.out.tolist()
        n = plan.n_seg
        return v[3 * n] != 0.0, v[3 * n + 1], v[3 * n + 2], v[2 * n:3 * n]

    def stage_combine_sel(self, out, y_acc, f_acc, y_rej, f_rej, coef, plan):
        _ok(self.lib.oracle_stage_combine_sel(out.data_ptr(), y_acc.data_ptr(), f_acc.data_ptr(), y_rej.data_ptr(),
                                              f_rej.data_ptr(), coef, plan.ctrl_dev.data_ptr(), out.numel(),
                                              _code(out.dtype)), "oracle_stage_combine_sel")

    def dense_eval(self, out, y0, y1, f0, f1, ks, coefs, dt, x):
        ptrs, cf, n = self._terms(ks, coefs)
        _ok(self.lib.oracle_dense_eval(out.data_ptr(), y0.data_ptr(), y1.data_ptr(), f0.data_ptr(), f1.data_ptr(),
                                       ptrs, cf, n, dt, x, y0.numel(), _code(y0.dtype)), "oracle_dense_eval")

    def dense_eval_multi(self, out_rows, y0, y1, f0, f1, ks, coefs, dt, xs):
        """Host twin of tdeq_dense_eval_multi: the reference evaluates one output time at a time."""
        for q, x in enumerate(xs):
            self.dense_eval(out_rows[q], y0, y1, f0, f1, ks, coefs, dt, x)

    def interp_fit(self, coeffs, y0, y1, f0, f1, ks, coefs, dt):
        ptrs, cf, n = self._terms(ks, coefs)
        _ok(self.lib.oracle_interp_fit(coeffs.data_ptr(), y0.data_ptr(), y1.data_ptr(), f0.data_ptr(),
                                       f1.data_ptr(), ptrs, cf, n, dt, y0.numel(), _code(y0.dtype)),
            "oracle_interp_fit")

    def rk4_stage(self, stage, out, y0, k1, k2, k3, k4, dt):
        p = lambda t: None if t is None else t.data_ptr()
        _ok(self.lib.oracle_rk4_38_stage(stage, out.data_ptr(), y0.data_ptr(), p(k1), p(k2), p(k3), p(k4), dt,
                                         y0.numel(), _code(y0.dtype)), "oracle_rk4_38_stage")

    def lerp(self, out, y0, y1, slope):
        _ok(self.lib.oracle_lerp(out.data_ptr(), y0.data_ptr(), y1.data_ptr(), slope, y0.numel(),
                                 _code(y0.dtype)), "oracle_lerp")

    def fixed_stage(self, mode, out, y0, ks, ws, dt):
        ptrs, cf, n = self._terms(ks, ws)
        _ok(self.lib.oracle_fixed_stage(mode, out.data_ptr(), y0.data_ptr(), ptrs, cf, n, dt, y0.numel(),
                                        _code(y0.dtype)), "oracle_fixed_stage")

    def weighted_sum(self, out, xs, ws):
        ptrs, cf, n = self._terms(xs, ws)
        _ok(self.lib.oracle_weighted_sum(out.data_ptr(), ptrs, cf, n, out.numel(), _code(out.dtype)),
            "oracle_weighted_sum")

    def scale_many(self, outs, g, ws):
        ptrs, cf, n = self._terms(outs, ws)
        _ok(self.lib.oracle_scale_many(ptrs, g.data_ptr(), cf, n, g.numel(), _code(g.dtype)), "oracle_scale_many")

    def multi_dot(self, g, xs):
        """fp64 tensor [len(xs)] of <g, x_m>."""
        n = len(xs)
        ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
        out = torch.empty(n, dtype=torch.float64)
        _ok(self.lib.oracle_multi_dot(g.data_ptr(), ptrs, n, g.numel(), out.data_ptr(), _code(g.dtype)),
            "oracle_multi_dot")
        return out

    def adams_predict(self, y_out, y0, hist, cb, cm=None, dt=0.0, dy_out=None, delta_out=None):
        ptrs, cbf, n = self._terms(hist, cb)
        cmf = None if cm is None else (ctypes.c_double * n)(*cm)
        p = lambda t: None if t is None else t.data_ptr()
        _ok(self.lib.oracle_adams_predict(y_out.data_ptr(), p(dy_out), p(delta_out), y0.data_ptr(), ptrs, cbf, cmf, n,
                                          dt, y0.numel(), _code(y0.dtype)), "oracle_adams_predict")

    def adams_correct(self, plan, dy_out, dy_old, y_out=None, f=None, delta=None, y0=None, c=0.0, compute=True):
        p = lambda t: None if t is None else t.data_ptr()
        _ok(self.lib.oracle_adams_correct(p(y_out), dy_out.data_ptr(), p(f), p(delta), dy_old.data_ptr(), p(y0), c,
                                          1 if compute else 0, plan.segs, plan.n_seg, plan.chunk, plan.n_chunks,
                                          dy_out.numel(), plan.out_ptr, plan.bad_ptr, _code(dy_out.dtype)),
            "oracle_adams_correct")

    def pack_segments(self, out, srcs, chunk_starts, numels, scales, chunk):
        n = len(srcs)
        ptrs = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in srcs])
        cs = (ctypes.c_int64 * n)(*chunk_starts)
        nm = (ctypes.c_int64 * n)(*numels)
        sc = (ctypes.c_double * n)(*scales)
        _ok(self.lib.oracle_pack_segments(out.data_ptr(), ptrs, cs, nm, sc, n, chunk, out.numel() // chunk,
                                          _code(out.dtype)), "oracle_pack_segments")

    def fill_scalars(self, dst, vals):
        """Host twin of tdeq_fill_scalars: vals converted to dst's dtype."""
        import torch as _torch
        dst.copy_(_torch.tensor(list(vals), dtype=_torch.float64).to(dst.dtype))
