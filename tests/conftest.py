import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("TDEQ_TEST_POISON") == "1":
        _poison_uninitialised_memory()


def _poison_uninitialised_memory():
    """TDEQ_TEST_POISON=1: every floating-point `torch.empty` / `empty_like` / `new_empty` comes back filled with NaN —
    a read of memory the package never wrote turns into a wrong or non-finite result instead of passing by luck
    (how the uninitialised flat-state padding of r03 was provoked; the whole suite is run this way once per round)."""
    import torch

    def poisoned(fn, factory=False):
        def wrapper(*args, **kw):
            if factory and kw.get("device") is None:
                # torch.set_default_device works through a function mode that recognises the ORIGINAL torch.empty object
                kw["device"] = torch.get_default_device()
            out = fn(*args, **kw)
            if isinstance(out, torch.Tensor) and (out.is_floating_point() or out.is_complex()) and out.numel():
                with torch.no_grad():
                    out.fill_(float("nan"))
            return out
        return wrapper
    torch.empty = poisoned(torch.empty, factory=True)
    torch.empty_like = poisoned(torch.empty_like)
    torch.Tensor.new_empty = poisoned(torch.Tensor.new_empty)


# Methods SURVEY.md §2 marks OUT OF SCOPE (rows 11, 12, 14: Adams multistep, implicit Runge-Kutta, the SciPy bridge).
# They are covered in full on the "cpu" half of every parametrised test; on the GPU box ONE representative case per (test
# function, method) runs — the driver's `-m gpu` budget belongs to the hot path.  TDEQ_FULL_GPU_MATRIX=1
# runs everything.  Likewise a few in-scope cases that are slow without adding GPU coverage (order-2 pairs at tight
# tolerances: hundreds of launch-bound steps of kernels the order-5 cases already exercise).
_OUT_OF_SCOPE_METHODS = {"explicit_adams", "implicit_adams", "fixed_adams", "implicit_euler", "implicit_midpoint", "trapezoid",
                         "radauIIA3", "gl4", "radauIIA5", "gl6", "sdirk2", "trbdf2", "scipy_solver"}
# adaptive_heun (order 2: hundreds of launch-bound steps per solve at the suites' tolerances) exercises the same kernel
# templates as every other pair with <= 2 terms per row; on the GPU box it runs where the METHOD is the subject
# (tests/test_methods_golden.py, test_backprop_golden.py, test_lookahead.py, test_graph_mode_gpu.py, test_brow_golden.py)
# and is left to the cpu half of the reference-suite / event / drop-in matrices.
_GPU_ELSEWHERE_METHODS = {"adaptive_heun"}
_GPU_ELSEWHERE_FILES = ("test_reference_suite.py", "test_events_golden.py", "test_dropin_golden.py", "test_detest_golden.py")


def pytest_collection_modifyitems(config, items):
    # (advisor r05) a plain `pytest tests/` on a box without a GPU: everything that carries the `gpu` marker is skipped
    # instead of failing with "No HIP GPUs are available"; an explicit `-m gpu` still runs (and fails) them
    markexpr = (config.getoption("markexpr", "") or "").replace(" ", "")
    asked_for_gpu = "gpu" in markexpr and "notgpu" not in markexpr      # `-m gpu`: a box without a GPU must fail loudly
    if not torch.cuda.is_available() and not asked_for_gpu:
        no_gpu = pytest.mark.skip(reason="needs a real MI355X (no GPU visible here)")
        for item in items:
            if item.get_closest_marker("gpu") is not None:
                item.add_marker(no_gpu)
    if os.environ.get("TDEQ_FULL_GPU_MATRIX") == "1":
        return
    seen = set()
    for item in items:
        cs = getattr(item, "callspec", None)
        if cs is None or cs.params.get("dev") != "cuda":
            continue
        if item.fspath.basename in _GPU_ELSEWHERE_FILES and \
                any(isinstance(v, str) and v in _GPU_ELSEWHERE_METHODS for v in cs.params.values()):
            item.add_marker(pytest.mark.skip(reason="slow on the GPU box without adding kernel coverage: this method's cuda "
                                                    "cases live in the method-level tests, the matrix on the cpu half"))
            continue
        out = sorted(v for v in cs.params.values() if isinstance(v, str) and v in _OUT_OF_SCOPE_METHODS)
        if not out:
            continue
        # (advisor r04: one representative per STATE KIND — a complex-state case exercises GPU-only code of its own, the complex
        #  norm kernels and ComplexHipKernels' host-side pieces, so it is kept next to the real one)
        is_complex = any(isinstance(v, str) and ("c64" in v or "c128" in v or "complex" in v.lower()) for v in cs.params.values()) \
            or any(getattr(v, "is_complex", False) is True for v in cs.params.values() if isinstance(v, torch.dtype))
        key = (item.function.__module__, item.function.__name__, tuple(out), is_complex)
        if key in seen:
            item.add_marker(pytest.mark.skip(reason="out-of-scope method (SURVEY.md §2): one representative cuda case per "
                                                    "test and method, the full matrix on the cpu half"))
        seen.add(key)


@pytest.fixture(scope="session")
def oracle_kernels():
    """CPU oracle with the HipKernels interface (test infrastructure, never used by the product)."""
    from oracle.kernels import OracleKernels
    return OracleKernels()


@pytest.fixture()
def cpu_backend(monkeypatch, oracle_kernels):
    """Run the product's HOST logic on CPU tensors by substituting the oracle for the HIP kernels.

    The product has no CPU path (get_kernels raises for non-ROCm devices); this patch exists only so
    the solver / adjoint / sharding control flow can be tested in the GPU-less container."""
    from torchdiffeq_amd import _native
    monkeypatch.setattr(_native, "get_kernels", lambda device, dtype=None: oracle_kernels)
    return oracle_kernels


@pytest.fixture(params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def dev(request, monkeypatch, oracle_kernels):
    """Device of a parity test.  "cuda": the real product path (HIP kernels).  "cpu": host-logic run with
    the oracle substituted for the kernels (see cpu_backend).  Also the default device of the test body."""
    import torch
    if request.param == "cpu":
        from torchdiffeq_amd import _native
        monkeypatch.setattr(_native, "get_kernels", lambda device, dtype=None: oracle_kernels)
    else:
        assert torch.cuda.is_available(), "gpu test on a box without a GPU"
    prev = torch.get_default_device()
    torch.set_default_device(request.param)
    if request.param == "cuda":
        # a "cuda" parity test whose state silently stayed on the CPU would exercise the host path, not the HIP
        # kernels: the package's HostPathWarning is an error here — for real AND (r04) complex states; only states below
        # fp32 are host-path by design, and their tests do not use this fixture
        import warnings
        from torchdiffeq_amd import _fallback
        monkeypatch.setattr(_fallback, "_warned", False)
        with warnings.catch_warnings():
            warnings.filterwarnings("error", category=_fallback.HostPathWarning)
            yield request.param
    else:
        yield request.param
    torch.set_default_device(prev)


@pytest.fixture(scope="session")
def hip_kernels():
    import torch
    from torchdiffeq_amd import _native
    assert torch.cuda.is_available(), "gpu test on a box without a GPU"
    return _native.get_kernels(torch.device("cuda:0"))
