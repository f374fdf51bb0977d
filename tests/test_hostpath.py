"""States the HIP kernels do not take (r03): tensors that are not on a ROCm device and complex states run on the
package's torch-op host path (torchdiffeq_amd/_fallback.py) — BASELINE.json configs[0] is "rk4 ... fp32 on CPU" and the
reference runs wherever its tensors live (odeint.py:49-108, misc.py:185).  Pinned to outputs of the reference itself
(tests/golden/solves.npz, tests/golden/hostpath.npz <- make_golden.py hostpath).  And the other half of the contract: a
real state on a ROCm device NEVER takes this path — with the library missing it raises."""
import warnings

import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from torchdiffeq_amd import _fallback, _native
from _cases import T, load, rel_err


@pytest.fixture()
def quiet():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", _fallback.HostPathWarning)
        yield


def test_cfg1_on_the_cpu_as_baseline_json_writes_it_is_bit_identical(quiet):
    """configs[0]: spiral, rk4, y0 in R^2, fp32 on the CPU — no GPU, no test backend substituted."""
    z = load("solves.npz")
    A, y0, t = T(z["cfg1_A"]), T(z["cfg1_y0"]), T(z["cfg1_t"])
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: (y_ ** 3) @ A, y0, t, method="rk4")
    assert y.device.type == "cpu" and torch.equal(y, T(z["cfg1_y"]))
    assert y[-1, 0].tolist() == [-0.4436032772064209, 0.27951884269714355]


def test_host_path_warns_once_and_says_why(monkeypatch):
    monkeypatch.setattr(_fallback, "_warned", False)
    y0, t = torch.ones(3), torch.tensor([0.0, 0.1])
    with pytest.warns(_fallback.HostPathWarning, match="lives on 'cpu'"):
        tda.odeint(lambda t_, y_: -y_, y0, t, method="euler")
    with warnings.catch_warnings():
        warnings.simplefilter("error", _fallback.HostPathWarning)
        tda.odeint(lambda t_, y_: -y_, y0, t, method="euler")      # second use: silent


def test_selection_is_by_the_state_alone(monkeypatch, tmp_path):
    """cuda + real dtype -> HipKernels or an error, never the host path (needs no GPU: only the selection runs)."""
    monkeypatch.setattr(_native, "_KERNELS", None)
    monkeypatch.setattr(_native, "_LIB_PATH", str(tmp_path / "missing.so"))
    for dtype in (torch.float32, torch.float64, None):
        with pytest.raises(_native.NativeLibraryError):
            _native.get_kernels(torch.device("cuda:0"), dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert isinstance(_native.get_kernels(torch.device("cpu"), torch.float32), _fallback.HostKernels)
        assert isinstance(_native.get_kernels(torch.device("cpu"), torch.complex64), _fallback.HostKernels)
    # r04: complex states on a ROCm device take the HIP kernels too — or raise; never the host path
    with pytest.raises(_native.NativeLibraryError):
        _native.get_kernels(torch.device("cuda:0"), torch.complex64)


@pytest.mark.gpu
def test_a_real_state_on_the_gpu_never_takes_the_host_path(monkeypatch):
    """Kernel-call counter: every HostKernels method is replaced by one that counts; a cuda solve must leave it at 0
    while the HIP library's launches are seen."""
    calls = {"host": 0, "hip": 0}
    for name in [n for n in vars(_fallback.HostKernels) if not n.startswith("_") and callable(getattr(_fallback.HostKernels, n))]:
        monkeypatch.setattr(_fallback.HostKernels, name, lambda *a, **k: calls.__setitem__("host", calls["host"] + 1))
    hip = _native.get_kernels(torch.device("cuda:0"), torch.float32)
    assert isinstance(hip, _native.HipKernels)
    orig = hip.stage_combine

    def counted(*a, **k):
        calls["hip"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(hip, "stage_combine", counted)
    y0 = torch.randn(64, 8, device="cuda")
    lin = torch.nn.Linear(8, 8).cuda()
    with torch.no_grad():
        tda.odeint(lambda t_, y_: -y_, y0, torch.tensor([0.0, 1.0], device="cuda"), method="dopri5")
        tda.odeint(lambda t_, y_: -y_, y0.double(), torch.tensor([0.0, 1.0], device="cuda"), method="rk4")
    x = y0.clone().requires_grad_(True)
    tda.odeint_adjoint(lambda t_, y_: lin(y_), x, torch.tensor([0.0, 0.5], device="cuda"),
                       adjoint_params=tuple(lin.parameters()))[-1].sum().backward()
    assert calls["host"] == 0 and calls["hip"] > 10


CASES = [(tag, method, d) for tag in ("c64", "c128") for method in ("dopri5", "dopri8", "rk4", "bosh3")
         for d in ("fwd", "rev")]


def _complex_case(z, tag, method, d, device):
    A, y0, t = T(z[f"{tag}_A"], device), T(z[f"{tag}_y0"], device), T(z[f"{tag}_{method}_{d}_t"], device)
    kw = {"dopri5": dict(rtol=1e-5, atol=1e-7), "dopri8": dict(rtol=1e-6, atol=1e-8),
          "rk4": dict(options=dict(step_size=0.05)), "bosh3": dict(rtol=1e-4, atol=1e-6)}[method]
    nfe = [0]

    def f(t_, y_):
        assert not t_.is_complex() and t_.dtype == (torch.float32 if tag == "c64" else torch.float64)
        nfe[0] += 1
        return y_ @ A.T
    with torch.no_grad():
        y = tda.odeint(f, y0, t, method=method, **kw)
    return y, nfe[0]


@pytest.mark.parametrize("tag,method,d", CASES)
def test_complex_states_vs_reference_cpu(quiet, tag, method, d):
    z = load("hostpath.npz")
    y, nfe = _complex_case(z, tag, method, d, "cpu")
    ref = z[f"{tag}_{method}_{d}_y"]
    assert y.dtype == (torch.complex64 if tag == "c64" else torch.complex128) and tuple(y.shape) == ref.shape
    # r04: the host path hands row sums and norms to ATen as the reference does (_fallback.py) — every method, adaptive
    # ones included, is the reference bit for bit on the CPU, with its evaluation count
    assert torch.equal(torch.view_as_real(y), torch.view_as_real(T(ref)))
    if method != "rk4":
        assert nfe == int(z[f"{tag}_{method}_{d}_nfe"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,method,d", [c for c in CASES if c[1] in ("dopri5", "rk4")])
def test_complex_states_vs_reference_gpu(quiet, tag, method, d):
    z = load("hostpath.npz")
    y, nfe = _complex_case(z, tag, method, d, "cuda")
    ref = z[f"{tag}_{method}_{d}_y"]
    assert y.is_cuda
    err = float((y.cpu() - T(ref)).abs().max() / np.abs(ref).max())
    assert err < (2e-5 if tag == "c64" else 1e-9), err


def test_gradient_of_a_real_loss_through_a_complex_solve(quiet):
    z = load("hostpath.npz")
    A, y0 = T(z["c128_A"]), T(z["c128_y0"]).requires_grad_(True)
    y = tda.odeint(lambda t_, y_: y_ @ A.T, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), method="dopri5",
                   rtol=1e-7, atol=1e-9)
    (y[-1].abs() ** 2).sum().backward()
    assert rel_err(torch.view_as_real(y0.grad), torch.view_as_real(T(z["c128_grad_y0"]))) < 1e-8


@pytest.mark.parametrize("method,kw", [("rk4", dict(options=dict(step_size=0.1))), ("dopri5", dict(rtol=1e-8, atol=1e-10)),
                                       ("midpoint", dict(options=dict(step_size=0.1)))])
def test_second_order_gradients_vs_reference(dev, method, kw):
    """Hessian-type quantities through plain odeint (the reference's op graph is differentiable to any order,
    rk_common.py:31-40): d/d(y0, W) of |dL/dy0|^2 + |dL/dW|^2, on the HIP path ("cuda") and on the host logic with the
    test backend ("cpu")."""
    z = load("hostpath.npz")
    W = T(z["hess_W"], dev).requires_grad_(True)
    x = T(z["hess_y0"], dev).requires_grad_(True)
    t = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64, device=dev)
    y = tda.odeint(lambda t_, y_: torch.tanh(y_ @ W.T), x, t, method=method, **kw)
    loss = (y[-1] ** 2).sum() + (y[1] ** 3).sum()
    gx, gW = torch.autograd.grad(loss, (x, W), create_graph=True)
    assert rel_err(gx, z[f"hess_{method}_gx"]) < 1e-9 and rel_err(gW, z[f"hess_{method}_gW"]) < 1e-9
    hx, hW = torch.autograd.grad((gx ** 2).sum() + (gW ** 2).sum(), (x, W))
    tol = 1e-9 if method != "dopri5" else 1e-6
    assert rel_err(hx, z[f"hess_{method}_hx"]) < tol, rel_err(hx, z[f"hess_{method}_hx"])
    assert rel_err(hW, z[f"hess_{method}_hW"]) < tol, rel_err(hW, z[f"hess_{method}_hW"])


def test_second_order_gradients_on_the_host_path(quiet):
    z = load("hostpath.npz")
    W = T(z["hess_W"]).requires_grad_(True)
    x = T(z["hess_y0"]).requires_grad_(True)
    y = tda.odeint(lambda t_, y_: torch.tanh(y_ @ W.T), x, torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64), method="rk4",
                   options=dict(step_size=0.1))
    loss = (y[-1] ** 2).sum() + (y[1] ** 3).sum()
    gx, gW = torch.autograd.grad(loss, (x, W), create_graph=True)
    hx, hW = torch.autograd.grad((gx ** 2).sum() + (gW ** 2).sum(), (x, W))
    assert rel_err(hx, z["hess_rk4_hx"]) < 1e-9 and rel_err(hW, z["hess_rk4_hW"]) < 1e-9


def test_host_path_adaptive_solves_match_the_reference(quiet):
    """The same golden cases the HIP path is held to, on plain CPU tensors (no backend substituted)."""
    z = load("solves.npz")
    for prefix, method, tol in (("cfg2_tight", "dopri5", 2e-5), ("cfg2_loose", "dopri5", 2e-5), ("cfg2_rev", "dopri5", 2e-5)):
        A, y0, t = T(z["cfg2_A"]), T(z["cfg2_y0"]), T(z[f"{prefix}_t"])
        rtol, atol = [float(v) for v in z[f"{prefix}_tol"]]
        nfe = [0]

        def f(t_, y_):
            nfe[0] += 1
            return y_ @ A.T
        with torch.no_grad():
            y = tda.odeint(f, y0, t, rtol=rtol, atol=atol, method=method)
        assert torch.equal(y, T(z[f"{prefix}_y"])) and nfe[0] == int(z[f"{prefix}_nfe"])      # r04: bit for bit
    A, y0, t = T(z["cfg4_A"]), T(z["cfg4_y0"]), T(z["cfg4_t"])
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: y_ @ A.T, y0, t, rtol=1e-9, atol=1e-11, method="dopri8")
    assert torch.equal(y, T(z["cfg4_y"]))


def test_host_path_adjoint_and_tuple_state(quiet):
    """odeint_adjoint with a tuple state on CPU tensors: gradients equal backprop through the solver."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(3, 3).double()

    class F(torch.nn.Module):
        def forward(self, t_, s):
            a, b = s
            return torch.tanh(lin(a)) - 0.1 * a, -b * a.pow(2).sum(-1, keepdim=True)
    f = F()
    f.lin = lin
    a0 = torch.randn(5, 3, dtype=torch.float64)
    b0 = torch.ones(5, 1, dtype=torch.float64)
    t = torch.tensor([0.0, 0.7], dtype=torch.float64)
    grads = []
    for fn in (tda.odeint_adjoint, tda.odeint):
        lin.zero_grad()
        x = a0.clone().requires_grad_(True)
        ya, yb = fn(f, (x, b0), t, rtol=1e-9, atol=1e-11, method="dopri5")
        (ya[-1].sum() + yb[-1].pow(2).sum()).backward()
        grads.append((x.grad.clone(), lin.weight.grad.clone()))
    assert rel_err(grads[0][0], grads[1][0]) < 1e-6 and rel_err(grads[0][1], grads[1][1]) < 1e-6


@pytest.mark.parametrize("method,kw", [("rk4", dict(options=dict(step_size=0.1))), ("dopri5", dict(rtol=1e-8, atol=1e-10))])
def test_second_order_gradients_with_times_in_the_graph(dev, method, kw):
    """Second-order quantities when `t` requires grad too: the fixed-grid interpolation / the dense output's weights
    are functions of the output times (polynomial in the interpolation point for the adaptive methods), so their
    higher derivatives matter (autodiff._Spec.w_fn)."""
    z = load("hostpath.npz")
    W = T(z["hess_W"], dev).requires_grad_(True)
    x = T(z["hess_y0"], dev).requires_grad_(True)
    tt = torch.tensor([0.0, 0.43, 1.0], dtype=torch.float64, device=dev, requires_grad=True)
    y = tda.odeint(lambda t_, y_: torch.tanh(y_ @ W.T) * torch.cos(t_), x, tt, method=method, **kw)
    loss = (y[-1] ** 2).sum() + (y[1] ** 3).sum()
    gx, gt = torch.autograd.grad(loss, (x, tt), create_graph=True)
    assert rel_err(gt, z[f"hesst_{method}_gt"]) < 1e-8
    hx, ht, hW = torch.autograd.grad((gx ** 2).sum() + (gt ** 2).sum(), (x, tt, W))
    tol = 1e-8 if method == "rk4" else 1e-5
    for got, key in ((hx, "hx"), (ht, "ht"), (hW, "hW")):
        assert rel_err(got, z[f"hesst_{method}_{key}"]) < tol, (key, rel_err(got, z[f"hesst_{method}_{key}"]))


@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("tag", ["fwd", "rev"])
def test_host_path_tuple_tolerances_are_the_references_flat_vectors(quiet, dname, tag):
    """misc.py:115-123: a tupled rtol / atol becomes ONE flat vector of the time dtype, so the reference's error ratio
    of an fp32 state is an fp64 number there (the HIP kernels take the entries per segment, in fp32: rounding-level
    agreement, tests/test_parity_golden.py).  The host path keeps the reference's forms — a 0-dim tensor or a vector per
    tolerance, `misc.vector_tolerances` — and reproduces tests/golden/tuple_tol.npz bit for bit, evaluation counts
    included.  (Found by `tools/fuzz_vs_reference.py vectol` under TDEQ_FUZZ_BACKEND=host, r04.)"""
    z = load("tuple_tol.npz")
    A, ya, yb = (T(z[f"tt_{dname}_{k}"]) for k in ("A", "y0a", "y0b"))
    t = torch.tensor([0.0, 0.7, 2.0], dtype=torch.float64)
    if tag == "rev":
        t = t.flip(0)
    count = [0]

    def f(t_, y_):
        count[0] += 1
        return y_[0] @ A.T * torch.cos(t_), -y_[1] * 0.5

    with torch.no_grad():
        sa, sb = tda.odeint(f, (ya, yb), t, rtol=(1e-5, 1e-3), atol=(1e-7, 1e-4), method="dopri5")
    assert count[0] == int(z[f"tt_{dname}_{tag}_nfe"])
    assert torch.equal(sa, T(z[f"tt_{dname}_{tag}_ya"])) and torch.equal(sb, T(z[f"tt_{dname}_{tag}_yb"]))
