"""ctypes binding of libtdeq_hip.so (include/tdeq_hip.h) and the tensor-level kernel interface.

`HipKernels` is the compute backend of the package: every state-sized arithmetic operation of the solvers on a
real fp32 / fp64 state on a ROCm device goes through it — `get_kernels()` raises if the library is missing, there is
no substitute on the GPU.  States the kernels do not take (not on a ROCm device, or complex) are served by the
torch-op `_fallback.HostKernels` behind the same interface.
"""
from __future__ import annotations

import contextlib
import ctypes
import functools
import math
import os
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

TDEQ_ABI_VERSION = 21
TDEQ_F32, TDEQ_F64 = 0, 1
TDEQ_C64, TDEQ_C128 = 2, 3        # interleaved complex: the norm entry points only (include/tdeq_hip.h)
TDEQ_BF16, TDEQ_F16 = 4, 5        # reduced-precision states: the entry points of the host-driven step (LowPrecisionHipKernels)
TDEQ_MAX_TERMS = 14
TDEQ_INLINE_SEGMENTS = 16
TDEQ_CHUNK_QUANTUM = 1024
TDEQ_MAX_STAGE_TIMES = 16
TDEQ_MAX_DENSE_OUTPUTS = 16
TDEQ_MAX_MULTI_OUT = 4

_LIB_NAME = "libtdeq_hip.so"
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)

_c_void_pp = ctypes.POINTER(ctypes.c_void_p)
_c_double_p = ctypes.POINTER(ctypes.c_double)


class Segment(ctypes.Structure):
    """`tdeq_segment` of include/tdeq_hip.h."""
    _fields_ = [("chunk_start", ctypes.c_int64), ("numel", ctypes.c_int64),
                ("rtol", ctypes.c_double), ("atol", ctypes.c_double)]


class StepCtrl(ctypes.Structure):
    """`tdeq_step_ctrl` of include/tdeq_hip.h: the scalar inputs of the device-resident step controller."""
    _fields_ = [("t0", ctypes.c_double), ("dt", ctypes.c_double), ("safety", ctypes.c_double),
                ("ifactor", ctypes.c_double), ("dfactor", ctypes.c_double), ("exponent", ctypes.c_double),
                ("min_step", ctypes.c_double), ("max_step", ctypes.c_double), ("time_sign", ctypes.c_double),
                ("alpha", ctypes.c_double * TDEQ_MAX_STAGE_TIMES), ("alpha_is_one", ctypes.c_uint32),
                ("n_times", ctypes.c_int32), ("n_norm_seg", ctypes.c_int32), ("leading_abs", ctypes.c_int32)]


class MultiOut(ctypes.Structure):
    """`tdeq_multi_out` of include/tdeq_hip.h: one output of tdeq_stage_combine_multi."""
    _fields_ = [("out", ctypes.c_void_p), ("coef", ctypes.c_double * TDEQ_MAX_TERMS), ("mask", ctypes.c_uint32),
                ("add_y0", ctypes.c_int32)]


# name -> (restype, argtypes); the authoritative list of exported symbols (checked by the tests).
ABI_SIGNATURES = {
    "tdeq_abi_version": (ctypes.c_int, []),
    "tdeq_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64]),
    "tdeq_stage_combine": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_void_pp, _c_double_p,
                                          ctypes.c_int, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_void_p]),
    "tdeq_stage_combine_timed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_void_pp, _c_double_p,
                                                ctypes.c_int, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tdeq_stage_combine_fill": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_void_pp, _c_double_p,
                                               ctypes.c_int, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
                                               ctypes.c_void_p, _c_double_p, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_error_norm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _c_void_pp, _c_double_p,
                                       ctypes.c_int, ctypes.c_double, ctypes.POINTER(Segment),
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_error_norm_vec": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _c_void_pp, _c_double_p, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p,
                                           ctypes.c_double, ctypes.POINTER(Segment), ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_error_norm_vec_ctrl": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _c_void_pp, _c_double_p, ctypes.c_int,
                                                ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p,
                                                ctypes.c_double, ctypes.POINTER(Segment), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.POINTER(StepCtrl), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_init_norms_vec": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(Segment),
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_stage_combine_err": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _c_void_pp,
                                              _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_double,
                                              ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_stage_combine_multi": (ctypes.c_int, [ctypes.POINTER(MultiOut), ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_void_p, _c_void_pp, ctypes.c_int, ctypes.c_double,
                                                ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_stage_combine_multi_dev": (ctypes.c_int, [ctypes.POINTER(MultiOut), ctypes.c_int, ctypes.c_void_p,
                                                    ctypes.c_void_p, _c_void_pp, ctypes.c_int, ctypes.c_void_p,
                                                    ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_stage_combine_multi_timed": (ctypes.c_int, [ctypes.POINTER(MultiOut), ctypes.c_int, ctypes.c_void_p,
                                                      ctypes.c_void_p, _c_void_pp, ctypes.c_int, ctypes.c_double,
                                                      ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p]),
    "tdeq_error_norm_partial": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _c_void_pp,
                                               _c_double_p, ctypes.c_int, ctypes.c_double, ctypes.POINTER(Segment),
                                               ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_error_norm_partial_ctrl": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _c_void_pp,
                                                    _c_double_p, ctypes.c_int, ctypes.c_double,
                                                    ctypes.POINTER(Segment), ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                    ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.POINTER(StepCtrl), ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                    ctypes.c_int, ctypes.c_void_p]),
    "tdeq_step_controller": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Segment), ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.POINTER(StepCtrl), ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_stage_combine_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _c_void_pp,
                                              _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                              ctypes.c_int, ctypes.c_void_p]),
    "tdeq_step_commit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_int, ctypes.c_void_p]),
    "tdeq_stage_combine_sel": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_int64,
                                              ctypes.c_int, ctypes.c_void_p]),
    "tdeq_init_norms": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.POINTER(Segment), ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_init_scaled": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.POINTER(Segment), ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_void_p]),
    "tdeq_dense_eval": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, _c_void_pp, _c_double_p, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_void_p]),
    "tdeq_dense_eval_multi": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, _c_void_pp, _c_double_p, ctypes.c_int,
                                             ctypes.c_double, _c_double_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_void_p]),
    "tdeq_interp_fit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, _c_void_pp, _c_double_p, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_rk4_38_stage": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_double, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_rk4_38_stage_dev": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_grid_advance": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_void_p]),
    "tdeq_grid_commit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_grid_advance_stages": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                                ctypes.c_int, ctypes.c_double, _c_double_p, ctypes.POINTER(ctypes.c_int),
                                                ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_void_p]),
    "tdeq_fixed_stage_dev": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _c_void_pp, _c_double_p,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_lerp": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double,
                                 ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_fixed_stage": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _c_void_pp, _c_double_p,
                                        ctypes.c_int, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_void_p]),
    "tdeq_weighted_sum": (ctypes.c_int, [ctypes.c_void_p, _c_void_pp, _c_double_p, ctypes.c_int, ctypes.c_int64,
                                         ctypes.c_int, ctypes.c_void_p]),
    "tdeq_scale_many": (ctypes.c_int, [_c_void_pp, ctypes.c_void_p, _c_double_p, ctypes.c_int, ctypes.c_int64,
                                       ctypes.c_int, ctypes.c_void_p]),
    "tdeq_dots_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int]),
    "tdeq_multi_dot": (ctypes.c_int, [ctypes.c_void_p, _c_void_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_pack_segments": (ctypes.c_int, [ctypes.c_void_p, _c_void_pp, ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(ctypes.c_int64), _c_double_p, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_fill_scalars": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p]),
    "tdeq_adams_predict": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          _c_void_pp, _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "tdeq_adams_correct": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int,
                                          ctypes.POINTER(Segment), ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
}


class NativeLibraryError(RuntimeError):
    pass


def library_path() -> str:
    return _LIB_PATH


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen the C-ABI library and bind every declared symbol.  Needs no GPU."""
    path = path or _LIB_PATH
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} not found: build it with `python -m torchdiffeq_amd.build` "
            "(hipcc --offload-arch=gfx950).  torchdiffeq_amd has no CPU/eager fallback.")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in ABI_SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    version = lib.tdeq_abi_version()
    if version != TDEQ_ABI_VERSION:
        raise NativeLibraryError(f"{path}: ABI version {version}, expected {TDEQ_ABI_VERSION}")
    return lib


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return TDEQ_F32
    if dtype == torch.float64:
        return TDEQ_F64
    if dtype == torch.complex64:
        return TDEQ_C64
    if dtype == torch.complex128:
        return TDEQ_C128
    if dtype == torch.bfloat16:
        return TDEQ_BF16
    if dtype == torch.float16:
        return TDEQ_F16
    raise TypeError(f"torchdiffeq_amd supports float32 / float64 (and complex64 / complex128) states, got {dtype}")


def _check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code} "
                           f"({'hipError_t' if code > 0 else 'argument error'})")


def pick_chunk(numel: int) -> int:
    """Elements per reduction chunk (one workgroup, one fp64 partial per chunk)."""
    env = os.environ.get("TDEQ_CHUNK")
    if env:
        return int(env)
    return 2048 if numel >= (1 << 20) else 1024


class NormPlan:
    """Segment table + workspace + read-back buffers of the norm kernels for one state layout.

    `segments` = [(element offset, numel, rtol, atol)]; offsets must be multiples of `chunk`.
    Results of the last norm launch are read with `HipKernels.read_norms(plan)`.
    """

    def __init__(self, segments: Sequence[Tuple[int, int, float, float]], total: int, chunk: int,
                 device: torch.device, pinned: bool):
        assert chunk % TDEQ_CHUNK_QUANTUM == 0
        self.chunk = chunk
        self.n_seg = len(segments)
        self.numels = [int(s[1]) for s in segments]
        self.n_chunks = max(1, math.ceil(total / chunk))
        arr = (Segment * self.n_seg)()
        for i, (off, numel, rtol, atol) in enumerate(segments):
            assert off % chunk == 0, "segment offsets must be chunk aligned"
            arr[i] = Segment(off // chunk, numel, float(rtol), float(atol))
        self.segs = arr
        self.segs_dev = None
        if self.n_seg > TDEQ_INLINE_SEGMENTS:
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            self.segs_dev = torch.from_numpy(raw).to(device)
        self.workspace = torch.empty(3 * self.n_chunks, dtype=torch.float64, device=device)
        self.workspace_bytes = self.workspace.numel() * 8
        # [2*n_seg sums | n_seg non-finite counters | 4 controller words]; pinned host memory is written by
        # the finalize kernel directly (zero-copy), so a read-back is one stream sync and no memcpy.
        self.pinned = pinned
        self.expect = ()         # index ranges the pending norm launch will write (poll mode)
        if pinned:
            self.out = torch.zeros(3 * self.n_seg + 4, dtype=torch.float64, device="cpu", pin_memory=True)
            self.out_np = self.out.numpy()
            self.out_bits = self.out_np.view(np.uint64)
        else:
            self.out = torch.zeros(3 * self.n_seg + 4, dtype=torch.float64, device=device)
            self.out_np = None
        self.out_ptr = self.out.data_ptr()
        self.bad_ptr = self.out_ptr + 16 * self.n_seg
        self.ctrl_ptr = self.out_ptr + 24 * self.n_seg
        # {accept, sign * T(dt'), t0', dt'} of the device-resident controller (tdeq_stage_combine_sel, hipGraph mode)
        self.ctrl_dev = torch.zeros(4, dtype=torch.float64, device=device)


class HipKernels:
    """Tensor-level wrappers of the C-ABI.  All launches go to the current torch HIP stream."""

    name = "hip"

    # Read-back modes (TDEQ_READBACK): "poll" (default) = the finalize kernel stores into pinned host
    # memory and the host spins on those words (each entry is one aligned 8-byte store, pre-set to a
    # sentinel NaN payload) — no driver-level stream sync on the critical path of the accept/reject loop;
    # "pinned" = same zero-copy store, waited for with a stream synchronize; "copy" = device buffer + memcpy.
    _SENTINEL = np.uint64(0x7FF8DEADBEEF0001)
    _POLL_TIMEOUT_S = 5.0

    def __init__(self, lib: ctypes.CDLL):
        self.lib = lib
        mode = os.environ.get("TDEQ_READBACK", "poll")
        if mode not in ("poll", "pinned", "copy"):
            raise ValueError(f"TDEQ_READBACK={mode!r}: expected poll | pinned | copy")
        self._mode = mode
        self._pinned = mode in ("poll", "pinned")

    def _arm(self, plan: "NormPlan", n_sum: int, ctrl: bool = False) -> None:
        """Before a norm launch in poll mode: mark the entries the launch will overwrite."""
        if self._mode == "poll" and plan.pinned:
            n = plan.n_seg
            plan.out_bits[:n_sum * n] = self._SENTINEL
            plan.out_bits[2 * n:3 * n + (4 if ctrl else 0)] = self._SENTINEL
            plan.expect = ((0, n_sum * n), (2 * n, 3 * n + (4 if ctrl else 0)))

    # -- helpers ---------------------------------------------------------------------------------
    # The raw hipStream_t of the CURRENT device's current torch stream.  `torch.cuda.current_stream().cuda_stream`
    # gives the same value but costs 4–8 us of host time per call (device-index resolution, an `is_available()` probe
    # that reads os.environ, a Stream object) — seven calls per trial step, 12–17 % of the host time of a launch-bound
    # step (tools/host_profile2.py, r02); the C accessors below cost ~0.2 us.
    _raw_stream = staticmethod(getattr(torch._C, "_cuda_getCurrentRawStream", None))
    _cur_device = staticmethod(getattr(torch._C, "_cuda_getDevice", None))

    @classmethod
    def _stream(cls) -> int:
        if cls._raw_stream is not None and cls._cur_device is not None:
            return cls._raw_stream(cls._cur_device())
        return torch.cuda.current_stream().cuda_stream

    # Host-side argument marshalling is on the critical path of small / medium states.  The entry points copy the
    # host arrays into the kernel-argument block before they return, so one pointer array per term count is reused
    # (overwritten per call) and the coefficient arrays of the (immutable, few) tableau rows are built once.
    # (ctypes releases the GIL during a call: the reused pointer arrays are per thread.)
    _TLS = threading.local()
    _COEF_ARRAYS: dict = {}
    _PACK_TABLES: dict = {}

    @classmethod
    def _terms(cls, ks: Sequence[torch.Tensor], coefs: Sequence[float]):
        n = len(ks)
        if n > TDEQ_MAX_TERMS:
            ptrs = (ctypes.c_void_p * n)(*[k.data_ptr() for k in ks])
        else:
            arrays = getattr(cls._TLS, "ptr_arrays", None)
            if arrays is None:
                arrays = cls._TLS.ptr_arrays = [(ctypes.c_void_p * m)() for m in range(TDEQ_MAX_TERMS + 1)]
            ptrs = arrays[n]
            for j in range(n):
                ptrs[j] = ks[j].data_ptr()
        if isinstance(coefs, tuple):        # (a tableau row's RowCoefs included)
            cf = cls._COEF_ARRAYS.get(coefs)
            if cf is None:
                if len(cls._COEF_ARRAYS) > 4096:      # not a tableau row cache any more: start over
                    cls._COEF_ARRAYS.clear()
                cf = cls._COEF_ARRAYS[coefs] = (ctypes.c_double * n)(*coefs)
        else:
            cf = (ctypes.c_double * n)(*coefs)
        return ptrs, cf, n

    def make_plan(self, segments, total, chunk, device) -> NormPlan:
        return NormPlan(segments, total, chunk, device, self._pinned)

    # -- kernels ---------------------------------------------------------------------------------
    def stage_combine(self, out, y0, ks, coefs, dt: float) -> None:
        ptrs, cf, n = self._terms(ks, coefs)
        _check(self.lib.tdeq_stage_combine(out.data_ptr(), y0.data_ptr(), ptrs, cf, n, dt, y0.numel(),
                                           dtype_code(y0.dtype), self._stream()), "tdeq_stage_combine")

    def stage_combine_timed(self, out, y0, ks, coefs, dt: float, start_event, stop_event) -> None:
        """`stage_combine` whose dispatch stamps two torch.cuda.Event(enable_timing=True) objects with its own begin /
        end times (measurement hook, see include/tdeq_hip.h).  The events must have been recorded once before (torch
        creates the underlying hipEvent_t lazily)."""
        ptrs, cf, n = self._terms(ks, coefs)
        _check(self.lib.tdeq_stage_combine_timed(out.data_ptr(), y0.data_ptr(), ptrs, cf, n, dt, y0.numel(),
                                                 dtype_code(y0.dtype), self._stream(), start_event.cuda_event,
                                                 stop_event.cuda_event), "tdeq_stage_combine_timed")

    def stage_combine_fill(self, out, y0, ks, coefs, dt: float, fill_dst, fill_vals) -> None:
        """stage_combine of a step's first stage that also writes `fill_vals` (stage times) into `fill_dst`."""
        ptrs, cf, n = self._terms(ks, coefs)
        m = len(fill_vals)
        fv = (ctypes.c_double * m)(*fill_vals)
        _check(self.lib.tdeq_stage_combine_fill(out.data_ptr(), y0.data_ptr(), ptrs, cf, n, dt, y0.numel(),
                                                dtype_code(y0.dtype), fill_dst.data_ptr(), fv, m, self._stream()),
               "tdeq_stage_combine_fill")

    def error_norm(self, plan: NormPlan, y0, y1, ks, coefs, dt: float, scaled_out=None) -> None:
        ptrs, cf, n = self._terms(ks, coefs)
        dev = plan.segs_dev.data_ptr() if plan.segs_dev is not None else None
        so = None if scaled_out is None else scaled_out.data_ptr()
        self._arm(plan, 1)
        _check(self.lib.tdeq_error_norm(so, y0.data_ptr(), y1.data_ptr(), ptrs, cf, n, dt, plan.segs, dev,
                                        plan.n_seg, plan.chunk, plan.n_chunks, plan.out_ptr, plan.bad_ptr,
                                        plan.workspace.data_ptr(), plan.workspace_bytes,
                                        dtype_code(y0.dtype), self._stream()), "tdeq_error_norm")

    vec_partial = True      # error_norm_vec[_ctrl] continue a partial error row (`partial=`) like error_norm_partial

    vec_ctrl_in_graph = True     # tdeq_error_norm_vec_ctrl takes `state_in_dev` (ABI 21): captured steps with per-element tolerances

    def error_norm_vec(self, plan: NormPlan, y0, y1, ks, coefs, dt: float, rtol, atol, partial=None) -> None:
        """`error_norm` with per-element tolerances (tdeq_error_norm_vec): `rtol` / `atol` = an fp64 device vector over the
        flat padded state, or a host float for a 0-dim tolerance (at least one vector).  `partial`: the error row's leading
        run from `stage_combine_err`; `ks` / `coefs` are the remaining stages then."""
        ptrs, cf, n = self._terms(ks, coefs)
        dev = plan.segs_dev.data_ptr() if plan.segs_dev is not None else None
        rv, rs = (rtol.data_ptr(), 0.0) if isinstance(rtol, torch.Tensor) else (None, float(rtol))
        av, as_ = (atol.data_ptr(), 0.0) if isinstance(atol, torch.Tensor) else (None, float(atol))
        self._arm(plan, 1)
        pp = None if partial is None else partial.data_ptr()
        _check(self.lib.tdeq_error_norm_vec(pp, y0.data_ptr(), y1.data_ptr(), ptrs, cf, n, dt, rv, rs, av, as_, plan.segs, dev,
                                            plan.n_seg, plan.chunk, plan.n_chunks, plan.out_ptr, plan.bad_ptr,
                                            plan.workspace.data_ptr(), plan.workspace_bytes, dtype_code(y0.dtype),
                                            self._stream()), "tdeq_error_norm_vec")

    def error_norm_vec_ctrl(self, plan: NormPlan, y0, y1, ks, coefs, dt: float, rtol, atol, ctrl: StepCtrl, next_times,
                            partial=None, state_in_dev: bool = False) -> None:
        """`error_norm_vec` whose finalize step also runs the step controller on the device (tdeq_error_norm_vec_ctrl) —
        read with `read_ctrl`.  `state_in_dev` (captured steps): the step size comes from `plan.ctrl_dev`, `dt` is ignored."""
        ptrs, cf, n = self._terms(ks, coefs)
        dev = plan.segs_dev.data_ptr() if plan.segs_dev is not None else None
        rv, rs = (rtol.data_ptr(), 0.0) if isinstance(rtol, torch.Tensor) else (None, float(rtol))
        av, as_ = (atol.data_ptr(), 0.0) if isinstance(atol, torch.Tensor) else (None, float(atol))
        self._arm(plan, 1, ctrl=True)
        pp = None if partial is None else partial.data_ptr()
        _check(self.lib.tdeq_error_norm_vec_ctrl(pp, y0.data_ptr(), y1.data_ptr(), ptrs, cf, n, dt, rv, rs, av, as_, plan.segs,
                                                 dev, plan.n_seg, plan.chunk, plan.n_chunks, plan.out_ptr, plan.bad_ptr,
                                                 ctypes.byref(ctrl), plan.ctrl_ptr, plan.ctrl_dev.data_ptr(),
                                                 next_times.data_ptr(), 1 if state_in_dev else 0, plan.workspace.data_ptr(),
                                                 plan.workspace_bytes, dtype_code(y0.dtype), self._stream()),
               "tdeq_error_norm_vec_ctrl")

    def init_norms_vec(self, plan: NormPlan, mode: int, a, b, yscale, rtol, atol) -> None:
        """`init_norms` with per-element tolerances (tdeq_init_norms_vec): `rtol` / `atol` = an fp64 device vector over the
        flat padded state, or a host float for a 0-dim tolerance (at least one vector); sums in fp64, the promoted type."""
        dev = plan.segs_dev.data_ptr() if plan.segs_dev is not None else None
        rv, rs = (rtol.data_ptr(), 0.0) if isinstance(rtol, torch.Tensor) else (None, float(rtol))
        av, as_ = (atol.data_ptr(), 0.0) if isinstance(atol, torch.Tensor) else (None, float(atol))
        self._arm(plan, 2 if mode == 0 else 1)
        _check(self.lib.tdeq_init_norms_vec(mode, a.data_ptr(), b.data_ptr(), yscale.data_ptr(), rv, rs, av, as_, plan.segs, dev,
                                            plan.n_seg, plan.chunk, plan.n_chunks, plan.out_ptr, plan.bad_ptr,
                                            plan.workspace.data_ptr(), plan.workspace_bytes, dtype_code(yscale.dtype),
                                            self._stream()), "tdeq_init_norms_vec")

    def stage_combine_err(self, out, err_out, y0, ks, coefs, err_coefs, dt: float) -> None:
        """Last combine of a step + partial embedded error over the same stages (tdeq_stage_combine_err)."""
        ptrs, cf, n = self._terms(ks, coefs)
        ef = (ctypes.c_double * n)(*err_coefs)
        _check(self.lib.tdeq_stage_combine_err(out.data_ptr(), err_out.data_ptr(), y0.data_ptr(), ptrs, cf, ef, n, dt,
                                               y0.numel(), dtype_code(y0.dtype), self._stream()),
               "tdeq_stage_combine_err")

    _MULTI_SPECS: dict = {}

    @classmethod
    def multi_spec(cls, rows):
        """The constant part of a tdeq_stage_combine_multi call, built once per tableau row group: `rows` =
        ((coefs over the call's stage streams, mask, add_y0), ...), one entry per output.  Returns a `tdeq_multi_out`
        array whose `out` pointers are filled in per call (per thread: ctypes releases the GIL during a call)."""
        per_thread = getattr(cls._TLS, "multi_specs", None)
        if per_thread is None:
            per_thread = cls._TLS.multi_specs = {}
        arr = per_thread.get(rows)
        if arr is None:
            if len(per_thread) > 1024:
                per_thread.clear()
            arr = (MultiOut * len(rows))()
            for o, (coefs, mask, add_y0) in enumerate(rows):
                assert len(coefs) <= TDEQ_MAX_TERMS and mask
                for j, c in enumerate(coefs):
                    arr[o].coef[j] = c
                arr[o].mask = mask
                arr[o].add_y0 = 1 if add_y0 else 0
            per_thread[rows] = arr
        return arr

    def stage_combine_multi_dev(self, outs, rows, y0, acc_in, ks, plan: "NormPlan") -> None:
        """`stage_combine_multi` with the step size read on the device from the plan's controller words (hipGraph mode)."""
        n = len(ks)
        ptrs = (ctypes.c_void_p * n)(*[k.data_ptr() for k in ks])
        spec = self.multi_spec(rows)
        for o, t in enumerate(outs):
            spec[o].out = t.data_ptr()
        _check(self.lib.tdeq_stage_combine_multi_dev(spec, len(outs), y0.data_ptr(),
                                                     None if acc_in is None else acc_in.data_ptr(), ptrs, n,
                                                     plan.ctrl_dev.data_ptr(), y0.numel(), dtype_code(y0.dtype),
                                                     self._stream()), "tdeq_stage_combine_multi_dev")

    def stage_combine_multi(self, outs, rows, y0, acc_in, ks, dt: float, events=None) -> None:
        """One pass over the stages `ks` producing len(outs) tensors (tdeq_stage_combine_multi): output o =
        [y0 +] [acc_in +] sum over the set bits j of rows[o].mask of (rows[o].coefs[j] * dt) * ks[j] — a row's own
        stage input plus carried left-to-right partial sums of later rows.  `rows` as for `multi_spec` (a tuple)."""
        n = len(ks)
        arrays = getattr(self._TLS, "ptr_arrays", None)
        if arrays is None:
            arrays = self._TLS.ptr_arrays = [(ctypes.c_void_p * m)() for m in range(TDEQ_MAX_TERMS + 1)]
        ptrs = arrays[n]
        for j in range(n):
            ptrs[j] = ks[j].data_ptr()
        spec = self.multi_spec(rows)
        for o, t in enumerate(outs):
            spec[o].out = t.data_ptr()
        if events is not None:      # measurement hook: (start, stop) torch events stamped by the dispatch itself
            _check(self.lib.tdeq_stage_combine_multi_timed(
                spec, len(outs), y0.data_ptr(), None if acc_in is None else acc_in.data_ptr(), ptrs, n, dt, y0.numel(),
                dtype_code(y0.dtype), self._stream(), events[0].cuda_event, events[1].cuda_event),
                "tdeq_stage_combine_multi_timed")
            return
        _check(self.lib.tdeq_stage_combine_multi(spec, len(outs), y0.data_ptr(),
                                                 None if acc_in is None else acc_in.data_ptr(), ptrs, n, dt,
                                                 y0.numel(), dtype_code(y0.dtype), self._stream()),
               "tdeq_stage_combine_multi")

    def error_norm_partial(self, plan: NormPlan, err_partial, y0, y1, ks, coefs, dt: float) -> None:
        """Error norm continuing `err_partial` with the remaining stages `ks` (0..2 of them)."""
        n = len(ks)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[k.data_ptr() for k in ks])
        cf = (ctypes.c_double * max(n, 1))(*coefs)
        dev = plan.segs_dev.data_ptr() if plan.segs_dev is not None else None
        self._arm(plan, 1)
        _check(self.lib.tdeq_error_norm_partial(err_partial.data_ptr(), y0.data_ptr(), y1.data_ptr(), ptrs, cf, n, dt,
                                                plan.segs, dev, plan.n_seg, plan.chunk, plan.n_chunks, plan.out_ptr,
                                                plan.bad_ptr, plan.workspace.data_ptr(), plan.workspace_bytes,
                                                dtype_code(y0.dtype), self._stream()), "tdeq_error_norm_partial")

    def error_scaled(self, plan: NormPlan, out, y0, y1, ks, coefs, dt: float) -> None:
        """err/tol materialised into `out` (for user norm callables); norms are produced as well."""
        self.error_norm(plan, y0, y1, ks, coefs, dt, scaled_out=out)

    def init_norms(self, plan: NormPlan, mode: int, a, b, yscale) -> None:
        dev = plan.segs_dev.data_ptr() if plan.segs_dev is not None else None
        self._arm(plan, 2 if mode == 0 else 1)
        _check(self.lib.tdeq_init_norms(mode, a.data_ptr(), b.data_ptr(), yscale.data_ptr(), plan.segs, dev,
                                        plan.n_seg, plan.chunk, plan.n_chunks, plan.out_ptr, plan.bad_ptr,
                                        plan.workspace.data_ptr(), plan.workspace_bytes,
                                        dtype_code(yscale.dtype), self._stream()), "tdeq_init_norms")

    def init_scaled(self, plan: NormPlan, mode: int, a, b, yscale, out0, out1=None) -> None:
        """a/scale, b/scale (mode 0) or (a-b)/scale (mode 1) materialised for a user norm (tdeq_init_scaled)."""
        dev = plan.segs_dev.data_ptr() if plan.segs_dev is not None else None
        _check(self.lib.tdeq_init_scaled(mode, a.data_ptr(), b.data_ptr(), yscale.data_ptr(), plan.segs, dev,
                                         plan.n_seg, plan.chunk, plan.n_chunks, out0.data_ptr(),
                                         None if out1 is None else out1.data_ptr(), dtype_code(yscale.dtype),
                                         self._stream()), "tdeq_init_scaled")

    def read_norms(self, plan: NormPlan) -> Tuple[List[float], List[float], List[float]]:
        """(sumsq[0:n_seg], sumsq[n_seg:2n_seg], nonfinite[0:n_seg]) of the last norm launch."""
        n = plan.n_seg
        v = self._read_out(plan)
        return v[:n], v[n:2 * n], v[2 * n:3 * n]

    def _read_out(self, plan: NormPlan) -> List[float]:
        """All words of the plan's result buffer once the pending norm launch has written its share."""
        if plan.pinned and self._mode == "poll" and plan.expect:
            bits, sent = plan.out_bits, self._SENTINEL
            (a0, a1), (b0, b1) = plan.expect
            spins, t_start = 0, None
            while (bits[a0:a1] == sent).any() or (bits[b0:b1] == sent).any():
                spins += 1
                if spins & 0x3FFF == 0:       # every 16k spins: bounded wait, then fall back to a real sync
                    import time
                    t_start = t_start or time.monotonic()
                    if time.monotonic() - t_start > self._POLL_TIMEOUT_S:
                        torch.cuda.current_stream().synchronize()
                        if (bits[a0:a1] == sent).any() or (bits[b0:b1] == sent).any():
                            raise RuntimeError("norm kernel did not write its results (poll timeout)")
                        break
            plan.expect = ()
            return plan.out_np.tolist()
        if plan.pinned:
            torch.cuda.current_stream().synchronize()
            return plan.out_np.tolist()
        return plan.out.tolist()

    norm_copies_last_stage = True       # tdeq_error_norm_partial_ctrl(copy_last_k=...), ABI 21

    def error_norm_partial_ctrl(self, plan: NormPlan, err_partial, y0, y1, ks, coefs, dt: float, ctrl: StepCtrl,
                                next_times, state_in_dev: bool = False, copy_last_to=None) -> None:
        """`error_norm_partial` whose finalize step also runs the step controller on the device: accept flag,
        next step size and the next trial step's stage times (`next_times`, T[n_times]) — read with `read_ctrl`.
        `copy_last_to` (captured steps): the launch also writes `ks[-1]` into this buffer."""
        n = len(ks)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[k.data_ptr() for k in ks])
        cf = (ctypes.c_double * max(n, 1))(*coefs)
        self._arm(plan, 1, ctrl=True)
        _check(self.lib.tdeq_error_norm_partial_ctrl(
            err_partial.data_ptr(), y0.data_ptr(), y1.data_ptr(), ptrs, cf, n, dt, plan.segs,
            plan.segs_dev.data_ptr() if plan.segs_dev is not None else None, plan.n_seg, plan.chunk,
            plan.n_chunks, plan.out_ptr, plan.bad_ptr, ctypes.byref(ctrl), plan.ctrl_ptr, plan.ctrl_dev.data_ptr(),
            next_times.data_ptr(), 1 if state_in_dev else 0, None if copy_last_to is None else copy_last_to.data_ptr(),
            plan.workspace.data_ptr(), plan.workspace_bytes, dtype_code(y0.dtype), self._stream()),
            "tdeq_error_norm_partial_ctrl")

    def step_controller(self, plan: NormPlan, sums_plan: NormPlan, numel_plan: NormPlan, ctrl: StepCtrl, next_times,
                        dtype: torch.dtype, state_in_dev: bool = False) -> None:
        """The controller alone (tdeq_step_controller) on the per-segment sums in `sums_plan.out` (DEVICE memory: a
        plan made with pinned=False, filled by a norm launch and all-reduced over the ranks of a lock-step sharded
        solve), with the element counts of `numel_plan`'s segment table (the GLOBAL counts); results land in `plan`
        like those of `error_norm_partial_ctrl` — read with `read_ctrl(plan)`."""
        assert not sums_plan.pinned and sums_plan.n_seg == plan.n_seg == numel_plan.n_seg
        self._arm(plan, 1, ctrl=True)
        _check(self.lib.tdeq_step_controller(
            sums_plan.out_ptr, sums_plan.bad_ptr, numel_plan.segs,
            numel_plan.segs_dev.data_ptr() if numel_plan.segs_dev is not None else None, plan.n_seg, plan.out_ptr,
            plan.bad_ptr, ctypes.byref(ctrl), plan.ctrl_ptr, plan.ctrl_dev.data_ptr(), next_times.data_ptr(),
            1 if state_in_dev else 0, dtype_code(dtype), self._stream()), "tdeq_step_controller")

    def read_ctrl(self, plan: NormPlan) -> Tuple[bool, float, float, List[float]]:
        """(accept, dt_next, error_ratio, nonfinite[0:n_seg]) of the last `error_norm_partial_ctrl` launch."""
        n = plan.n_seg
        v = self._read_out(plan)
        return v[3 * n] != 0.0, v[3 * n + 1], v[3 * n + 2], v[2 * n:3 * n]

    def stage_combine_dev(self, out, err_out, y0, ks, coefs, err_coefs, plan: NormPlan) -> None:
        """stage_combine / stage_combine_err (err_out given) with the step size read on the device from the plan's
        controller words (hipGraph mode)."""
        ptrs, cf, n = self._terms(ks, coefs)
        ef = None if err_coefs is None else (ctypes.c_double * n)(*err_coefs)
        _check(self.lib.tdeq_stage_combine_dev(out.data_ptr(), None if err_out is None else err_out.data_ptr(),
                                               y0.data_ptr(), ptrs, cf, ef, n, plan.ctrl_dev.data_ptr(), y0.numel(),
                                               dtype_code(y0.dtype), self._stream()), "tdeq_stage_combine_dev")

    def step_commit(self, y_prev, f_prev, y_cur, f_cur, y1, f1, plan: NormPlan) -> None:
        _check(self.lib.tdeq_step_commit(y_prev.data_ptr(), f_prev.data_ptr(), y_cur.data_ptr(), f_cur.data_ptr(),
                                         y1.data_ptr(), f1.data_ptr(), plan.ctrl_dev.data_ptr(), y_cur.numel(),
                                         dtype_code(y_cur.dtype), self._stream()), "tdeq_step_commit")

    def arm_readback(self, plan: NormPlan, n_sum: int = 1, ctrl: bool = True) -> None:
        """Mark the words a norm launch inside a replayed hipGraph is about to write (poll mode)."""
        self._arm(plan, n_sum, ctrl=ctrl)

    def stage_combine_sel(self, out, y_acc, f_acc, y_rej, f_rej, coef: float, plan: NormPlan) -> None:
        """First stage of the next trial step on the pair the device controller selected (tdeq_stage_combine_sel)."""
        _check(self.lib.tdeq_stage_combine_sel(out.data_ptr(), y_acc.data_ptr(), f_acc.data_ptr(), y_rej.data_ptr(),
                                               f_rej.data_ptr(), coef, plan.ctrl_dev.data_ptr(), out.numel(),
                                               dtype_code(out.dtype), self._stream()), "tdeq_stage_combine_sel")

    def dense_eval(self, out, y0, y1, f0, f1, ks, coefs, dt: float, x: float) -> None:
        ptrs, cf, n = self._terms(ks, coefs)
        _check(self.lib.tdeq_dense_eval(out.data_ptr(), y0.data_ptr(), y1.data_ptr(), f0.data_ptr(),
                                        f1.data_ptr(), ptrs, cf, n, dt, x, y0.numel(),
                                        dtype_code(y0.dtype), self._stream()), "tdeq_dense_eval")

    def dense_eval_multi(self, out_rows, y0, y1, f0, f1, ks, coefs, dt: float, xs: Sequence[float]) -> None:
        """Dense output at several points of ONE step: out_rows[q] = y(x_q) for the rows of a contiguous
        [len(xs), n] tensor (a slice of the solution)."""
        ptrs, cf, n = self._terms(ks, coefs)
        m = len(xs)
        xa = (ctypes.c_double * m)(*xs)
        _check(self.lib.tdeq_dense_eval_multi(out_rows.data_ptr(), out_rows.stride(0) if m > 1 else y0.numel(),
                                              y0.data_ptr(), y1.data_ptr(), f0.data_ptr(), f1.data_ptr(), ptrs, cf, n,
                                              dt, xa, m, y0.numel(), dtype_code(y0.dtype), self._stream()),
               "tdeq_dense_eval_multi")

    def interp_fit(self, coeffs, y0, y1, f0, f1, ks, coefs, dt: float) -> None:
        ptrs, cf, n = self._terms(ks, coefs)
        _check(self.lib.tdeq_interp_fit(coeffs.data_ptr(), y0.data_ptr(), y1.data_ptr(), f0.data_ptr(),
                                        f1.data_ptr(), ptrs, cf, n, dt, y0.numel(),
                                        dtype_code(y0.dtype), self._stream()), "tdeq_interp_fit")

    def rk4_stage(self, stage: int, out, y0, k1, k2, k3, k4, dt: float) -> None:
        p = lambda t: None if t is None else t.data_ptr()
        _check(self.lib.tdeq_rk4_38_stage(stage, out.data_ptr(), y0.data_ptr(), p(k1), p(k2), p(k3), p(k4),
                                          dt, y0.numel(), dtype_code(y0.dtype), self._stream()),
               "tdeq_rk4_38_stage")

    # -- hipGraph mode of the fixed-grid rk4 solver (device-resident step state) -----------------------
    def rk4_stage_dev(self, stage: int, out, y0, k1, k2, k3, k4, dt_dev) -> None:
        p = lambda t: None if t is None else t.data_ptr()
        _check(self.lib.tdeq_rk4_38_stage_dev(stage, out.data_ptr(), y0.data_ptr(), p(k1), p(k2), p(k3), p(k4),
                                              dt_dev.data_ptr(), y0.numel(), dtype_code(y0.dtype), self._stream()),
               "tdeq_rk4_38_stage_dev")

    def grid_advance(self, grid, counter, perturb: bool, sign: float, times_out, dt_out) -> None:
        _check(self.lib.tdeq_grid_advance(grid.data_ptr(), dtype_code(grid.dtype), grid.numel(), counter.data_ptr(),
                                          1 if perturb else 0, sign, times_out.data_ptr(), dt_out.data_ptr(),
                                          dtype_code(times_out.dtype), self._stream()), "tdeq_grid_advance")

    def grid_advance_stages(self, grid, counter, perturb: bool, sign: float, fracs, modes, times_out, dt_out) -> None:
        """tdeq_grid_advance_stages: the next step's dt and stage times t0 + dt*frac_i / t1 (mode bits: 1 = t1,
        2 = Perturb.NEXT, 4 = Perturb.PREV), formed on the device."""
        n = len(fracs)
        fr = (ctypes.c_double * n)(*fracs)
        md = (ctypes.c_int * n)(*modes)
        _check(self.lib.tdeq_grid_advance_stages(grid.data_ptr(), dtype_code(grid.dtype), grid.numel(),
                                                 counter.data_ptr(), 1 if perturb else 0, sign, fr, md, n,
                                                 times_out.data_ptr(), dt_out.data_ptr(), dtype_code(times_out.dtype),
                                                 self._stream()), "tdeq_grid_advance_stages")

    def fixed_stage_dev(self, mode: int, out, y0, ks, ws, dt_dev) -> None:
        """fixed_stage with the step size read from device memory (hipGraph mode)."""
        ptrs, cf, n = self._terms(ks, ws)
        _check(self.lib.tdeq_fixed_stage_dev(mode, out.data_ptr(), y0.data_ptr(), ptrs, cf, n, dt_dev.data_ptr(),
                                             y0.numel(), dtype_code(y0.dtype), self._stream()), "tdeq_fixed_stage_dev")

    def grid_commit(self, solution, y_cur, y_new, counter) -> None:
        _check(self.lib.tdeq_grid_commit(solution.data_ptr(), solution.stride(0), y_cur.data_ptr(), y_new.data_ptr(),
                                         counter.data_ptr(), y_cur.numel(), dtype_code(y_cur.dtype), self._stream()),
               "tdeq_grid_commit")

    def lerp(self, out, y0, y1, slope: float) -> None:
        _check(self.lib.tdeq_lerp(out.data_ptr(), y0.data_ptr(), y1.data_ptr(), slope, y0.numel(),
                                  dtype_code(y0.dtype), self._stream()), "tdeq_lerp")

    def fixed_stage(self, mode: int, out, y0, ks, ws, dt: float) -> None:
        """mode 0: out = y0 + dt*(sum_j k_j*w_j); mode 1: out = y0 + (dt*k_0)*w_0 (low-order fixed-grid steps)."""
        ptrs, cf, n = self._terms(ks, ws)
        _check(self.lib.tdeq_fixed_stage(mode, out.data_ptr(), y0.data_ptr(), ptrs, cf, n, dt, y0.numel(),
                                         dtype_code(y0.dtype), self._stream()), "tdeq_fixed_stage")

    def weighted_sum(self, out, xs, ws) -> None:
        """out = (x_0*w_0 + x_1*w_1) + ... (left to right)."""
        ptrs, cf, n = self._terms(xs, ws)
        _check(self.lib.tdeq_weighted_sum(out.data_ptr(), ptrs, cf, n, out.numel(), dtype_code(out.dtype),
                                          self._stream()), "tdeq_weighted_sum")

    def scale_many(self, outs, g, ws) -> None:
        """outs[m] = ws[m] * g (backward of the linear kernels: g is read once)."""
        ptrs, cf, n = self._terms(outs, ws)
        _check(self.lib.tdeq_scale_many(ptrs, g.data_ptr(), cf, n, g.numel(), dtype_code(g.dtype), self._stream()),
               "tdeq_scale_many")

    def multi_dot(self, g, xs) -> torch.Tensor:
        """fp64 device tensor [len(xs)] of <g, x_m> (no host read-back)."""
        n = len(xs)
        ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
        out = torch.empty(n, dtype=torch.float64, device=g.device)
        nbytes = self.lib.tdeq_dots_workspace_bytes(g.numel(), n)
        ws = torch.empty(max(1, nbytes // 8), dtype=torch.float64, device=g.device)
        _check(self.lib.tdeq_multi_dot(g.data_ptr(), ptrs, n, g.numel(), out.data_ptr(), ws.data_ptr(), nbytes,
                                       dtype_code(g.dtype), self._stream()), "tdeq_multi_dot")
        return out

    # -- Adams–Bashforth(–Moulton) -----------------------------------------------------------------
    def adams_predict(self, y_out, y0, hist, cb, cm=None, dt: float = 0.0, dy_out=None, delta_out=None) -> None:
        """y_out = y0 + sum_j T(cb_j) f_j [, dy_out = that sum, delta_out = T(dt) * sum_j T(cm_j) f_j] in one pass over
        the history `hist` (newest first) — tdeq_adams_predict."""
        ptrs, cbf, n = self._terms(hist, cb)
        cmf = None if cm is None else (ctypes.c_double * n)(*cm)
        p = lambda t: None if t is None else t.data_ptr()
        _check(self.lib.tdeq_adams_predict(y_out.data_ptr(), p(dy_out), p(delta_out), y0.data_ptr(), ptrs, cbf, cmf, n,
                                           dt, y0.numel(), dtype_code(y0.dtype), self._stream()), "tdeq_adams_predict")

    def adams_correct(self, plan: NormPlan, dy_out, dy_old, y_out=None, f=None, delta=None, y0=None, c: float = 0.0,
                      compute: bool = True) -> None:
        """One corrector iteration (dy_out = T(c) f + delta, y_out = y0 + dy_out) fused with its convergence census;
        compute=False: census of (dy_old, dy_out) only.  Read the per-segment counts with `read_norms(plan)`."""
        p = lambda t: None if t is None else t.data_ptr()
        dev = plan.segs_dev.data_ptr() if plan.segs_dev is not None else None
        self._arm(plan, 1)
        _check(self.lib.tdeq_adams_correct(p(y_out), dy_out.data_ptr(), p(f), p(delta), dy_old.data_ptr(), p(y0), c,
                                           1 if compute else 0, plan.segs, dev, plan.n_seg, plan.chunk, plan.n_chunks,
                                           dy_out.numel(), plan.out_ptr, plan.bad_ptr, plan.workspace.data_ptr(),
                                           plan.workspace_bytes, dtype_code(dy_out.dtype), self._stream()),
               "tdeq_adams_correct")

    def pack_segments(self, out, srcs, chunk_starts: Sequence[int], numels: Sequence[int], scales: Sequence[float],
                      chunk: int) -> None:
        """out (a whole number of chunks) <- the pieces `srcs` (contiguous tensors of out's dtype, or None = zeros) at
        their chunk-aligned segment starts, times +-1; padding zero-filled.  One launch (tdeq_pack_segments)."""
        n = len(srcs)
        ptrs = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in srcs])
        # the three constant tables of a layout are converted once (keyed by identity-stable tuples)
        key = (tuple(chunk_starts), tuple(numels), tuple(scales))
        tabs = self._PACK_TABLES.get(key)
        if tabs is None:
            if len(self._PACK_TABLES) > 1024:
                self._PACK_TABLES.clear()
            tabs = self._PACK_TABLES[key] = ((ctypes.c_int64 * n)(*chunk_starts), (ctypes.c_int64 * n)(*numels),
                                             (ctypes.c_double * n)(*scales))
        cs, nm, sc = tabs
        _check(self.lib.tdeq_pack_segments(out.data_ptr(), ptrs, cs, nm, sc, n, chunk, out.numel() // chunk,
                                           dtype_code(out.dtype), self._stream()), "tdeq_pack_segments")

    def fill_scalars(self, dst, vals: Sequence[float]) -> None:
        n = len(vals)
        arr = (ctypes.c_double * n)(*vals)
        _check(self.lib.tdeq_fill_scalars(dst.data_ptr(), arr, n, dtype_code(dst.dtype), self._stream()),
               "tdeq_fill_scalars")


class ComplexHipKernels:
    """complex64 / complex128 states on the HIP kernels (r04; r03 ran them as torch ops).

    Every LINEAR operation of a Runge–Kutta step has real coefficients (rk_common.py:79,89,201-205, interp.py,
    rk_common.py:110-157), and ATen itself evaluates `complex * real` per component — so those calls go to the real
    kernels on the state's interleaved (re, im) view: zero copy, 2n elements of the real type, bit-identical to the
    torch-op path.  The tolerance-scaled norms (misc.py:80-82, 50-56 with |.| the complex modulus) have kernels of their
    own (csrc/tdeq_kernels_complex.hpp, dtype codes TDEQ_C64 / TDEQ_C128): those calls pass the complex tensors through,
    with segment tables, chunks and counts in complex elements."""

    name = "hip-complex"
    _NORM_CALLS = frozenset(("error_norm", "error_norm_partial", "error_norm_partial_ctrl", "error_scaled", "init_norms",
                             "init_scaled"))
    _AS_IS = frozenset(("make_plan", "read_norms", "read_ctrl", "arm_readback", "multi_spec", "_arm", "_read_out",
                        "_stream", "_terms"))

    def __init__(self, inner: "HipKernels"):
        self._inner = inner
        self.lib = inner.lib
        self._host = None

    @staticmethod
    def _real(x):
        if isinstance(x, torch.Tensor):
            if not x.is_complex():
                return x
            r = torch.view_as_real(x)
            return r.reshape(-1) if x.dim() <= 1 else r.reshape(x.shape[0], -1)
        if isinstance(x, (list, tuple)) and any(isinstance(t, torch.Tensor) for t in x):
            return type(x)(ComplexHipKernels._real(t) for t in x)
        return x

    def __getattr__(self, name):
        attr = getattr(self._inner, name)          # AttributeError for what HipKernels does not have either
        if not callable(attr) or name in self._AS_IS or name in self._NORM_CALLS:
            return attr
        real = self._real

        def on_real_views(*args, **kwargs):
            return attr(*[real(a) for a in args], **{k: real(v) for k, v in kwargs.items()})
        on_real_views.__name__ = name
        self.__dict__[name] = on_real_views         # next lookup skips __getattr__
        return on_real_views

    # -- the calls whose COUNTS are in complex elements ------------------------------------------------------------
    def pack_segments(self, out, srcs, chunk_starts, numels, scales, chunk: int) -> None:
        self._inner.pack_segments(self._real(out), [None if t is None else self._real(t.reshape(-1)) for t in srcs],
                                  chunk_starts, [2 * int(n) for n in numels], scales, 2 * chunk)

    def step_controller(self, plan, sums_plan, numel_plan, ctrl, next_times, dtype, state_in_dev: bool = False) -> None:
        real = torch.float32 if dtype in (torch.complex64, torch.float32) else torch.float64
        self._inner.step_controller(plan, sums_plan, numel_plan, ctrl, next_times, real, state_in_dev)

    # -- Adams–Moulton corrector: its convergence census needs |.| (fixed_adams.py:189-192) and has no complex kernel;
    #    SURVEY.md §2 puts the multistep methods out of scope, so this one step stays a torch-op evaluation on the device
    def adams_correct(self, plan, dy_out, dy_old, y_out=None, f=None, delta=None, y0=None, c: float = 0.0,
                      compute: bool = True) -> None:
        from . import _fallback
        if self._host is None:
            self._host = _fallback.HostKernels()
        hp = getattr(plan, "_host_plan", None)
        if hp is None:
            segs = [(int(sg.chunk_start) * plan.chunk, int(sg.numel), float(sg.rtol), float(sg.atol)) for sg in plan.segs]
            hp = plan._host_plan = _fallback.HostPlan(segs, plan.n_chunks * plan.chunk, plan.chunk)
        self._host.adams_correct(hp, dy_out, dy_old, y_out, f, delta, y0, c, compute)
        n = plan.n_seg
        if plan.pinned:
            plan.out_np[:n] = hp.sums0
            plan.out_np[2 * n:3 * n] = hp.bad
        else:
            plan.out[:n] = torch.tensor(hp.sums0, dtype=torch.float64)
            plan.out[2 * n:3 * n] = torch.tensor(hp.bad, dtype=torch.float64)
        plan.expect = ()


class _GuardProbe:
    """Test hook for the device guard on a one-GPU box (tests/test_dist_gpu.py): `pretend_current` = the index the guard
    should take for the caller's current device — a state on cuda:0 then looks like a state on a NON-current device
    and the switching branch runs for real (to device 0); `switched` counts how often it did."""
    pretend_current = None
    switched = 0


def device_guard(device):
    """Context that makes `device` the current HIP device.  Every launch goes to `torch.cuda.current_stream()` — the
    CURRENT device's stream — and a kernel cannot be launched into another device's stream, so each entry point of the
    package runs under this guard: a state on `cuda:1` is integrated on cuda:1's stream whatever the caller's current
    device is.  Free when the device already is the current one (and for the CPU tensors of the host-logic tests)."""
    device = torch.device(device)
    if device.type != "cuda" or device.index is None:
        return contextlib.nullcontext()
    current = torch.cuda.current_device() if _GuardProbe.pretend_current is None else _GuardProbe.pretend_current
    if device.index == current:
        return contextlib.nullcontext()
    _GuardProbe.switched += 1
    return torch.cuda.device(device)


def on_state_device(method):
    """Decorator for solver entry points (`integrate*`): run under `device_guard(self.y0.device)`."""
    @functools.wraps(method)
    def guarded(self, *args, **kwargs):
        with device_guard(self.y0.device):
            return method(self, *args, **kwargs)
    return guarded


_KERNELS: Optional[HipKernels] = None
_HOST_KERNELS = None
_LOW_KERNELS = None
_LOW_HIP_KERNELS = None
_COMPLEX_KERNELS = None


def get_kernels(device: torch.device, dtype: Optional[torch.dtype] = None):
    """The compute backend for a state of `dtype` on `device`.

    A real or complex fp32 / fp64 state on a ROCm device -> the HIP kernels, and ONLY those: a missing or stale
    libtdeq_hip.so raises `NativeLibraryError` here (no silent substitute on the GPU); complex states through
    `ComplexHipKernels` (real kernels on the (re, im) view + the complex norm kernels).  A state that the kernels do
    not take — not on a ROCm device (BASELINE.json configs[0] is a CPU case; the reference runs wherever its tensors
    live, odeint.py:49-108) — -> `_fallback.HostKernels` / `LowPrecisionHostKernels`, the same interface in torch ops,
    with one `HostPathWarning` per process.  bf16 / fp16 states on a ROCm device -> `_lowp.LowPrecisionHipKernels`."""
    global _KERNELS, _HOST_KERNELS, _LOW_KERNELS, _LOW_HIP_KERNELS, _COMPLEX_KERNELS
    device = torch.device(device)
    is_complex = dtype is not None and dtype.is_complex
    if dtype in (torch.bfloat16, torch.float16):
        # states below fp32: integrated in their own precision like the reference's (misc.py:185-187), every operation
        # rounded to the storage type as ATen rounds it — on a ROCm device by the HIP kernels of csrc/tdeq_kernels_lp.hpp
        # (`_lowp.LowPrecisionHipKernels`: a missing library raises here, as for fp32), elsewhere by torch ops
        if device.type == "cuda":
            if _LOW_HIP_KERNELS is None:
                if _KERNELS is None:
                    _KERNELS = HipKernels(load_library())
                from . import _lowp
                _LOW_HIP_KERNELS = _lowp.LowPrecisionHipKernels(_KERNELS)
            return _LOW_HIP_KERNELS
        from . import _fallback
        _fallback.warn_once(f"the state is {dtype} on '{device}'")
        if _LOW_KERNELS is None:
            _LOW_KERNELS = _fallback.LowPrecisionHostKernels()
        return _LOW_KERNELS
    if device.type != "cuda":
        from . import _fallback
        _fallback.warn_once(f"the state lives on '{device}'")
        if _HOST_KERNELS is None:
            _HOST_KERNELS = _fallback.HostKernels()
        return _HOST_KERNELS
    if _KERNELS is None:
        _KERNELS = HipKernels(load_library())
    if is_complex:
        if _COMPLEX_KERNELS is None:
            _COMPLEX_KERNELS = ComplexHipKernels(_KERNELS)
        return _COMPLEX_KERNELS
    return _KERNELS
