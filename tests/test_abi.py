"""The C-ABI library builds for gfx950, loads without a GPU and exports what include/tdeq_hip.h declares."""
import contextlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tdeq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tdeq_[a-z0-9_]+)\s*\(", text)))


def test_build_and_symbols():
    from torchdiffeq_amd import _native, build
    path = build.build()
    assert os.path.exists(path)
    lib = _native.load_library(path)          # binds every symbol of ABI_SIGNATURES, checks the version
    declared = _declared_symbols()
    assert declared, "header parse failed"
    assert sorted(_native.ABI_SIGNATURES) == declared
    for name in declared:
        assert hasattr(lib, name)
    assert lib.tdeq_abi_version() == _native.TDEQ_ABI_VERSION
    assert lib.tdeq_workspace_bytes(10) == 10 * 3 * 8


def test_argument_errors_without_gpu():
    """Validation happens before any launch, so bad arguments are reported without a GPU."""
    import ctypes
    from torchdiffeq_amd import _native
    lib = _native.load_library()
    buf = (ctypes.c_double * 4)()
    ptrs = (ctypes.c_void_p * 1)(ctypes.addressof(buf))
    p = ctypes.addressof(buf)
    assert lib.tdeq_stage_combine(None, p, ptrs, buf, 1, 0.1, 4, 1, None) == -1       # null out
    assert lib.tdeq_stage_combine(p, p, ptrs, buf, 0, 0.1, 4, 1, None) == -1          # n_terms < 1
    assert lib.tdeq_stage_combine(p, p, ptrs, buf, 15, 0.1, 4, 1, None) == -1         # n_terms > 14
    assert lib.tdeq_stage_combine(p, p, ptrs, buf, 1, 0.1, 4, 7, None) == -1          # bad dtype
    assert lib.tdeq_stage_combine(p, p, ptrs, buf, 1, 0.1, 0, 1, None) == 0           # empty state: no-op
    assert lib.tdeq_rk4_38_stage(5, p, p, p, p, p, p, 0.1, 4, 1, None) == -1
    assert lib.tdeq_fill_scalars(p, buf, 17, 1, None) == -1
    assert lib.tdeq_fixed_stage(2, p, p, ptrs, buf, 1, 0.1, 4, 1, None) == -1         # bad mode
    assert lib.tdeq_fixed_stage(1, p, p, ptrs, buf, 2, 0.1, 4, 1, None) == -1         # mode 1 takes one term
    assert lib.tdeq_fixed_stage(0, p, p, ptrs, buf, 5, 0.1, 4, 1, None) == -1         # n_terms > 4
    assert lib.tdeq_weighted_sum(p, ptrs, buf, 9, 4, 1, None) == -1                   # n_terms > 8
    assert lib.tdeq_weighted_sum(p, ptrs, buf, 1, 0, 1, None) == 0
    assert lib.tdeq_stage_combine_fill(p, p, ptrs, buf, 3, 0.1, 4, 1, p, buf, 2, None) == -1   # n_terms > 2
    assert lib.tdeq_stage_combine_fill(p, p, ptrs, buf, 1, 0.1, 4, 1, p, buf, 17, None) == -1  # n_fill > 16
    assert lib.tdeq_stage_combine_err(p, None, p, ptrs, buf, buf, 1, 0.1, 4, 1, None) == -1    # null err_out
    assert lib.tdeq_error_norm_partial(p, p, p, ptrs, buf, 3, 0.1, None, None, 1, 1024, 1, p, p, p, 24, 1, None) == -1
    assert lib.tdeq_scale_many(ptrs, p, buf, 15, 4, 1, None) == -1                    # n_out > 14
    assert lib.tdeq_scale_many(ptrs, p, buf, 1, 0, 1, None) == 0
    assert lib.tdeq_dots_workspace_bytes(4096 * 3 + 1, 5) == 4 * 5 * 8
    assert lib.tdeq_multi_dot(p, ptrs, 1, 4, p, p, 0, 1, None) == -2                  # workspace too small
    # Adams entry points (ABI v13)
    assert lib.tdeq_adams_predict(None, None, None, p, ptrs, buf, None, 1, 0.1, 4, 1, None) == -1     # null y_out
    assert lib.tdeq_adams_predict(p, p, None, p, ptrs, buf, buf, 1, 0.1, 4, 1, None) == -1            # dy without delta
    assert lib.tdeq_adams_predict(p, p, p, p, ptrs, buf, None, 1, 0.1, 4, 1, None) == -1              # implicit without cm
    assert lib.tdeq_adams_predict(p, None, None, p, ptrs, buf, None, 15, 0.1, 4, 1, None) == -1       # n_terms > 14
    assert lib.tdeq_adams_predict(p, None, None, p, ptrs, buf, None, 1, 0.1, 0, 1, None) == 0         # empty state
    seg = (_native.Segment * 1)(_native.Segment(0, 4, 1e-3, 1e-4))
    assert lib.tdeq_adams_correct(None, p, None, None, p, None, 0.1, 1, seg, None, 1, 1024, 1, 4, p, p, p, 24, 1,
                                  None) == -1                                                         # compute needs f, delta, y0
    assert lib.tdeq_adams_correct(None, p, None, None, p, None, 0.1, 0, seg, None, 1, 1024, 1, 4, p, p, p, 8, 1,
                                  None) == -2                                                         # workspace too small
    assert lib.tdeq_adams_correct(None, p, None, None, p, None, 0.1, 0, seg, None, 1, 1000, 1, 4, p, p, p, 24, 1,
                                  None) == -1                                                         # chunk not a multiple of 1024


def test_gpu_state_without_the_library_is_rejected_loudly(monkeypatch, tmp_path):
    """A real state on a ROCm device has exactly one backend: libtdeq_hip.so.  Missing library -> NativeLibraryError,
    never a substitute (the torch-op host path serves CPU / complex states only: tests/test_hostpath.py)."""
    import torch
    from torchdiffeq_amd import _fallback, _native
    monkeypatch.setattr(_native, "_KERNELS", None)
    monkeypatch.setattr(_native, "_LIB_PATH", str(tmp_path / "no_such_lib.so"))
    with pytest.raises(_native.NativeLibraryError):
        _native.get_kernels(torch.device("cuda", 0), torch.float32)
    monkeypatch.setattr(_native, "_LOW_HIP_KERNELS", None)
    with pytest.raises(_native.NativeLibraryError):           # r05: reduced-precision states too (csrc/tdeq_kernels_lp.hpp)
        _native.get_kernels(torch.device("cuda", 0), torch.bfloat16)
    with pytest.warns(_fallback.HostPathWarning) if not _fallback._warned else contextlib.nullcontext():
        assert _native.get_kernels(torch.device("cpu"), torch.float32).name == "host"


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "torchdiffeq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") or f.endswith(".hip") or f.endswith(".hpp"):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "rk_oracle" not in text, f


def test_no_kernel_uses_scratch_memory(tmp_path):
    """Cross-compile the device code (no GPU needed) and check that NO kernel needs private (scratch) memory: a
    spilled argument block once cost the step-controller kernel +10 us (an address select between a device pointer
    and a kernel-argument field made the compiler copy 840 B of arguments to scratch)."""
    import re
    import shutil
    import subprocess
    from torchdiffeq_amd import build as tbuild
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    asm = tmp_path / "tdeq.s"
    flags = [f for f in tbuild.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([hipcc] + flags + ["-S", "--cuda-device-only", tbuild.SOURCES[0], "-o", str(asm)],
                          stderr=subprocess.DEVNULL)
    text = asm.read_text()
    names = re.findall(r"\.amdhsa_kernel (\S+)", text)
    sizes = [int(v) for v in re.findall(r"\.amdhsa_private_segment_fixed_size (\d+)", text)]
    assert len(names) == len(sizes) and len(names) > 100
    spilled = {n: s for n, s in zip(names, sizes) if s}
    assert not spilled, spilled
