"""odeint / odeint_adjoint against the reference's golden outputs (tests/golden/*.npz).

Every test runs twice through the `dev` fixture (conftest.py):
  dev = "cuda"  (-m gpu)      the product end to end on the MI355X through libtdeq_hip.so;
  dev = "cpu"   (-m "not gpu") the product's HOST logic (check_inputs, solver drivers, adjoint) on CPU
                               tensors with the oracle substituted for the HIP kernels — test-only."""
import math
import warnings

import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from _cases import SOLVE_CASES, TUPLE_TOL_LIST_FORMS, PlanarCNF, StatFunc, T, linear_case, load, make_mlp, rel_err



def test_cfg1_bit_exact(dev):
    """cfg1 (spiral, rk4): same torch func as the reference -> bit-identical trajectory."""
    z = load("solves.npz")
    A, y0, t = T(z["cfg1_A"], dev), T(z["cfg1_y0"], dev), T(z["cfg1_t"], dev)
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: (y_ ** 3) @ A, y0, t, method="rk4")
    assert y.shape == (1000, 1, 2) and y.dtype == torch.float32
    if dev == "cpu":      # same torch CPU func as the reference -> bit-identical over all 999 steps
        assert torch.equal(y, T(z["cfg1_y"], dev))
        assert y[-1, 0].tolist() == [-0.4436032772064209, 0.27951884269714355]
    else:                 # y**3 @ A is evaluated by the GPU; the solver arithmetic itself is bit-exact
        assert rel_err(y, z["cfg1_y"]) < 1e-5


@pytest.mark.parametrize("prefix,method,tol", SOLVE_CASES)
def test_cfg2_reduced(dev, prefix, method, tol):
    z = load("solves.npz")
    A, y0, t, rtol, atol = linear_case(z, prefix, dev)
    f = StatFunc(lambda t_, y_: y_ @ A.T)
    with torch.no_grad():
        y = tda.odeint(f, y0, t, rtol=rtol, atol=atol, method=method)
    assert torch.equal(y[0], y0)
    assert rel_err(y, z[f"{prefix}_y"]) < tol
    assert f.nfe == int(z[f"{prefix}_nfe"])
    assert len(f.accept) == len(z[f"{prefix}_accept_dt"]) and len(f.reject) == len(z[f"{prefix}_reject_dt"])
    assert len(f.steps) == len(f.accept) + len(f.reject)
    np.testing.assert_allclose(f.accept, z[f"{prefix}_accept_dt"], rtol=5e-2)


def test_cfg4_reduced_dopri8(dev):
    z = load("solves.npz")
    A, y0, t = T(z["cfg4_A"], dev), T(z["cfg4_y0"], dev), T(z["cfg4_t"], dev)
    f = StatFunc(lambda t_, y_: y_ @ A.T)
    with torch.no_grad():
        y = tda.odeint(f, y0, t, rtol=1e-9, atol=1e-11, method="dopri8")
    assert rel_err(y, z["cfg4_y"]) < 1e-7       # see tests/test_oracle_golden.py for the noise-floor note
    assert f.nfe == int(z["cfg4_nfe"])


def test_time_dependent_field_with_rejections(dev):
    """202 accepted + 22 rejected steps in the reference; fp64 so the step sequence must match."""
    z = load("solves.npz")
    A, y0, t = T(z["tdep_A"], dev), T(z["tdep_y0"], dev), T(z["tdep_t"], dev)
    f = StatFunc(lambda t_, y_: torch.sin(3 * t_) * (y_ @ A.T) * 4 - y_ ** 3)
    with torch.no_grad():
        y = tda.odeint(f, y0, t, rtol=1e-8, atol=1e-10, method="dopri5")
    assert rel_err(y, z["tdep_y"]) < 1e-10
    assert (len(f.accept), len(f.reject), f.nfe) == (len(z["tdep_accept_dt"]), len(z["tdep_reject_dt"]), int(z["tdep_nfe"]))
    np.testing.assert_allclose(f.accept, z["tdep_accept_dt"], rtol=1e-6)
    np.testing.assert_allclose(f.reject, z["tdep_reject_dt"], rtol=1e-6)


def test_tuple_state(dev):
    z = load("solves.npz")
    A, ya, yb, t = T(z["tuple_A"], dev), T(z["tuple_ya"], dev), T(z["tuple_yb"], dev), T(z["tuple_t"], dev)
    with torch.no_grad():
        out = tda.odeint(lambda t_, y_: (y_[0] @ A.T, 2 * (y_[1] @ A.T)), (ya, yb), t, rtol=1e-6, atol=1e-8)
    assert isinstance(out, tuple) and len(out) == 2
    assert out[0].shape == (2, 10, 8) and out[1].shape == (2, 20, 8)
    assert rel_err(out[0], z["tuple_out_a"]) < 2e-6 and rel_err(out[1], z["tuple_out_b"]) < 2e-6


@pytest.mark.parametrize("tag,opts", [
    ("first_step", dict(first_step=0.01)),
    ("step_t", dict(step_t=torch.tensor([0.25, 1.5]))),
    ("jump_t", dict(jump_t=torch.tensor([0.7]))),
    ("max_step", dict(max_step=0.05)),
    ("min_step", dict(min_step=0.2)),
])
def test_adaptive_options(dev, tag, opts):
    """first_step / step_t / jump_t / max_step / min_step (rk_common.py:166-177, 293-308, 324-330)."""
    z = load("solves.npz")
    A, y0, t = T(z["opt_A"], dev), T(z["opt_y0"], dev), T(z["opt_t"], dev)
    f = StatFunc(lambda t_, y_: y_ @ A.T)
    with torch.no_grad():
        y = tda.odeint(f, y0, t, rtol=1e-6, atol=1e-8, method="dopri5", options=opts)
    assert rel_err(y, z[f"opt_{tag}_y"]) < 1e-9
    assert f.nfe == int(z[f"opt_{tag}_nfe"])
    np.testing.assert_allclose(f.accept, z[f"opt_{tag}_accept_dt"], rtol=1e-6)
    np.testing.assert_allclose(f.reject, z[f"opt_{tag}_reject_dt"], rtol=1e-6)


def test_rk4_step_size_and_perturb(dev):
    """Fixed grid from `step_size` with linear interpolation of the outputs, and `perturb`
    (solvers.py:86-96, 117-125; misc.py:185-196): no reductions -> bit-exact."""
    z = load("solves.npz")
    A, y0, t = T(z["rk4s_A"], dev), T(z["rk4s_y0"], dev), T(z["rk4s_t"], dev)
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: y_ @ A.T, y0, t, method="rk4", options=dict(step_size=0.1))
        yp = tda.odeint(lambda t_, y_: torch.cos(t_) * (y_ @ A.T), y0, t, method="rk4",
                        options=dict(step_size=0.1, perturb=True))
    if dev == "cpu":      # same torch CPU func as the reference -> bit-identical
        assert torch.equal(y, T(z["rk4s_y"], dev))
        assert torch.equal(yp, T(z["rk4s_y_perturb"], dev))
    else:                 # the field's GEMM runs on the GPU (rocBLAS accumulation order differs from the CPU's)
        assert rel_err(y, z["rk4s_y"]) < 1e-6 and rel_err(yp, z["rk4s_y_perturb"]) < 1e-6


@pytest.mark.parametrize("tag,tol", [("f32", 1e-5), ("f64", 1e-12)])
@pytest.mark.parametrize("norm_tag,aopts", [("default", None), ("seminorm", dict(norm="seminorm"))])
def test_adjoint_cfg3_reduced(dev, tag, tol, norm_tag, aopts):
    z = load("adjoint.npz")
    f = make_mlp(z, tag, dev)
    y0 = T(z[f"adj_{tag}_y0"], dev).requires_grad_(True)
    t = T(z[f"adj_{tag}_t"], dev)
    rtol, atol = [float(v) for v in z[f"adj_{tag}_tol"]]
    y = tda.odeint_adjoint(f, y0, t, rtol=rtol, atol=atol, method="dopri5", adjoint_options=aopts)
    loss = y[-1].pow(2).sum() + (y[1:].sum() if len(t) > 2 else 0.0)
    loss.backward()
    assert rel_err(y.detach(), z[f"adj_{tag}_{norm_tag}_y"]) < tol
    assert rel_err(y0.grad, z[f"adj_{tag}_{norm_tag}_grad_y0"]) < tol
    for i, p in enumerate(f.parameters()):
        assert rel_err(p.grad, z[f"adj_{tag}_{norm_tag}_grad_p{i}"]) < tol, i


def test_cnf_cfg5_reduced(dev):
    """cfg5 (reduced batch): CNF, tuple state (z, logp), decreasing time 10 -> 0, dopri5 + adjoint at
    rtol = atol = 1e-5, vs the reference run on examples/cnf.py's own model (golden)."""
    z = load("cnf.npz")
    f = PlanarCNF(z, dev)
    z0 = T(z["cnf_z0"], dev).requires_grad_(True)
    logp0 = torch.zeros(z0.shape[0], 1)
    t = T(z["cnf_t"], dev)
    z_t, logp_t = tda.odeint_adjoint(f, (z0, logp0), t, atol=1e-5, rtol=1e-5, method="dopri5")
    (logp_t[-1].mean() - z_t[-1].pow(2).sum() / 100).backward()
    # rtol = 1e-5: two correct solvers agree to ~10*rtol on the solution and on its gradients
    assert rel_err(z_t.detach(), z["cnf_z"]) < 1e-4
    assert rel_err(logp_t.detach(), z["cnf_logp"]) < 1e-4
    assert rel_err(z0.grad, z["cnf_grad_z0"]) < 1e-3
    for i, p in enumerate(f.parameters()):
        assert rel_err(p.grad, z[f"cnf_grad_p{i}"]) < 1e-3, i


def test_adjoint_unused_parameter_gets_exact_zero(dev):
    """gradient_tests.py:89-135 behaviour: parameters that do not influence f get exactly 0."""
    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.used = torch.nn.Linear(4, 4).double()
            self.unused = torch.nn.Linear(4, 4).double()

        def forward(self, t, y):
            return torch.tanh(self.used(y))

    torch.manual_seed(0)
    f = F()
    y0 = torch.randn(6, 4, dtype=torch.float64, requires_grad=True)
    y = tda.odeint_adjoint(f, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), rtol=1e-8, atol=1e-10)
    y[-1].sum().backward()
    assert f.used.weight.grad.abs().max() > 0
    assert torch.equal(f.unused.weight.grad, torch.zeros(4, 4, dtype=torch.float64))
    assert torch.equal(f.unused.bias.grad, torch.zeros(4, dtype=torch.float64))


def test_adjoint_matches_finite_differences_and_time_grad(dev):
    """dL/dy0, dL/dθ and dL/dt (t.requires_grad) against central finite differences, fp64."""
    torch.manual_seed(1)
    W = torch.nn.Parameter(torch.randn(3, 3, dtype=torch.float64) * 0.5)

    def field(t_, y_, W_=None):
        W_ = W if W_ is None else W_
        return torch.sin(t_) * torch.tanh(y_ @ W_.T) - 0.1 * y_

    y0 = torch.randn(2, 3, dtype=torch.float64, requires_grad=True)
    t = torch.tensor([0.0, 0.7, 1.3], dtype=torch.float64, requires_grad=True)

    def loss_of(y0_, t_, W_):
        with torch.no_grad():
            y = tda.odeint(lambda a, b: field(a, b, W_), y0_, t_, rtol=1e-11, atol=1e-12)
        return float((y[1].sum() + y[2].pow(2).sum()))

    y = tda.odeint_adjoint(field, y0, t, rtol=1e-10, atol=1e-12, adjoint_params=(W,))
    (y[1].sum() + y[2].pow(2).sum()).backward()
    eps = 1e-6
    for idx in [(0, 0), (1, 2)]:
        d = torch.zeros_like(y0)
        d[idx] = eps
        fd = (loss_of((y0 + d).detach(), t.detach(), W.detach()) - loss_of((y0 - d).detach(), t.detach(), W.detach())) / (2 * eps)
        assert y0.grad[idx].item() == pytest.approx(fd, rel=1e-5, abs=1e-7)
    d = torch.zeros_like(W)
    d[1, 2] = eps
    fd = (loss_of(y0.detach(), t.detach(), (W + d).detach()) - loss_of(y0.detach(), t.detach(), (W - d).detach())) / (2 * eps)
    assert W.grad[1, 2].item() == pytest.approx(fd, rel=1e-5, abs=1e-7)
    for i in range(3):
        d = torch.zeros(3, dtype=torch.float64)
        d[i] = eps
        fd = (loss_of(y0.detach(), t.detach() + d, W.detach()) - loss_of(y0.detach(), t.detach() - d, W.detach())) / (2 * eps)
        assert t.grad[i].item() == pytest.approx(fd, rel=1e-4, abs=1e-6), i


def test_adjoint_tuple_state_and_reverse_time(dev):
    torch.manual_seed(2)
    lin = torch.nn.Linear(3, 3).double()

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = lin

        def forward(self, t, y):
            a, b = y
            return torch.tanh(self.lin(a)), -b * a.sum(-1, keepdim=True)

    f = F()
    a0 = torch.randn(5, 3, dtype=torch.float64, requires_grad=True)
    b0 = torch.randn(5, 1, dtype=torch.float64, requires_grad=True)
    t = torch.tensor([1.0, 0.2], dtype=torch.float64)       # decreasing time
    ya, yb = tda.odeint_adjoint(f, (a0, b0), t, rtol=1e-10, atol=1e-12)
    (ya[-1].pow(2).sum() + yb[-1].sum()).backward()

    def loss_of(a_, b_):
        with torch.no_grad():
            oa, ob = tda.odeint(f, (a_, b_), t, rtol=1e-11, atol=1e-12)
        return float(oa[-1].pow(2).sum() + ob[-1].sum())

    eps = 1e-6
    d = torch.zeros_like(a0)
    d[2, 1] = eps
    fd = (loss_of((a0 + d).detach(), b0.detach()) - loss_of((a0 - d).detach(), b0.detach())) / (2 * eps)
    assert a0.grad[2, 1].item() == pytest.approx(fd, rel=1e-5, abs=1e-8)
    d = torch.zeros_like(b0)
    d[3, 0] = eps
    fd = (loss_of(a0.detach(), (b0 + d).detach()) - loss_of(a0.detach(), (b0 - d).detach())) / (2 * eps)
    assert b0.grad[3, 0].item() == pytest.approx(fd, rel=1e-5, abs=1e-8)
    assert lin.weight.grad.abs().max() > 0


def test_adjoint_rk4_and_custom_adjoint_norm(dev):
    torch.manual_seed(3)
    lin = torch.nn.Linear(3, 3).double()
    f = lambda t_, y_: torch.tanh(lin(y_))
    params = tuple(lin.parameters())
    y0 = torch.randn(4, 3, dtype=torch.float64, requires_grad=True)
    t = torch.linspace(0, 1, 21, dtype=torch.float64)
    y = tda.odeint_adjoint(f, y0, t, method="rk4", adjoint_params=params)
    y[-1].sum().backward()
    g_rk4 = y0.grad.clone()
    y0.grad = None
    seen = []

    def my_norm(tensors):
        seen.append(tuple(x.shape for x in tensors))
        return max(x.abs().max() for x in tensors if x.numel())

    y = tda.odeint_adjoint(f, y0, t[[0, -1]], rtol=1e-9, atol=1e-11, adjoint_params=params,
                           adjoint_options=dict(norm=my_norm))
    y[-1].sum().backward()
    assert seen and seen[0] == ((), (4, 3), (4, 3), (3, 3), (3,))      # (t, y, adj_y, *params) — adjoint.py:247
    assert rel_err(y0.grad, g_rk4) < 1e-5
    assert isinstance(y.grad_fn.adjoint_options, dict) and "norm" in y.grad_fn.adjoint_options   # norm_tests.py:128


def test_api_errors_and_warnings(dev):
    y0 = torch.ones(3)
    t = torch.tensor([0.0, 1.0])
    f = lambda t_, y_: -y_
    with pytest.raises(ValueError, match="Invalid method"):
        tda.odeint(f, y0, t, method="no_such_method")
    with pytest.raises(AssertionError):
        tda.odeint(f, y0, torch.tensor([0.0, 1.0, 0.5]))                # not monotone
    with pytest.raises(TypeError):
        tda.odeint(f, y0, torch.tensor([0, 1]))                         # integer t
    with pytest.raises(ValueError, match="nn.Module"):
        tda.odeint_adjoint(f, y0, t)                                    # no adjoint_params for a lambda
    with pytest.raises(ValueError, match="adjoint_options"):
        tda.odeint_adjoint(f, y0, t, method="dopri5", options={}, adjoint_method="rk4", adjoint_params=())
    with pytest.warns(UserWarning, match="Unexpected arguments"):
        with torch.no_grad():
            tda.odeint(f, y0, t, method="dopri5", options=dict(no_such_option=1))
    with pytest.raises(AssertionError, match="max_num_steps"):
        with torch.no_grad():
            tda.odeint(f, y0, t, method="dopri5", options=dict(max_num_steps=2, first_step=1e-4))
    with pytest.raises(AssertionError, match="non-finite"):
        with torch.no_grad():
            tda.odeint(f, torch.tensor([1.0, float("nan")]), t, options=dict(first_step=0.1))
    with pytest.raises(AssertionError, match="underflow"):      # NaN state -> NaN first step -> dt = min_step = 0
        with torch.no_grad():
            tda.odeint(f, torch.tensor([1.0, float("nan")]), t)
    with pytest.raises(AssertionError, match="underflow"):
        with torch.no_grad():
            tda.odeint(lambda t_, y_: y_ * float("nan"), y0, t)
    # len(t) == 1 (odeint_tests.py:98-111)
    with torch.no_grad():
        y = tda.odeint(f, y0, torch.tensor([0.5]))
    assert y.shape == (1, 3) and torch.equal(y[0], y0)


def test_t_on_other_device_warns_and_default_method(dev):
    y0 = torch.ones(2, dtype=torch.float64)
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: -y_, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), rtol=1e-9, atol=1e-12)
    assert y[-1, 0].item() == pytest.approx(math.exp(-1), rel=1e-8)      # default method = dopri5


@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("tag", ["fwd", "rev"])
def test_tuple_state_with_per_component_tolerances(dev, dname, tag):
    """misc.py:115-123: tupled rtol / atol apply per component (= per segment of the flat state).  fp64: the same
    steps as the reference (1e-12).  fp32: the reference promotes the ratio to fp64 for vector tolerances, this
    path forms it in fp32 — same NFE, agreement at fp32 rounding level."""
    z = load("tuple_tol.npz")
    A, ya, yb = (T(z[f"tt_{dname}_{k}"], dev) for k in ("A", "y0a", "y0b"))
    t = torch.tensor([0.0, 0.7, 2.0], dtype=torch.float64, device=dev)
    if tag == "rev":
        t = t.flip(0)
    count = [0]

    def f(t_, y_):
        count[0] += 1
        return y_[0] @ A.T * torch.cos(t_), -y_[1] * 0.5

    with torch.no_grad():
        sa, sb = tda.odeint(f, (ya, yb), t, rtol=(1e-5, 1e-3), atol=(1e-7, 1e-4), method="dopri5")
    assert count[0] == int(z[f"tt_{dname}_{tag}_nfe"])
    tol = 1e-12 if dname == "f64" else 2e-6
    assert rel_err(sa, z[f"tt_{dname}_{tag}_ya"]) < tol
    assert rel_err(sb, z[f"tt_{dname}_{tag}_yb"]) < tol


@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("form", sorted(TUPLE_TOL_LIST_FORMS))
def test_tuple_tolerance_entries_that_are_lists_or_arrays(dev, dname, form):
    """misc.py:113-123 (`torch.as_tensor(tol_).expand(shape.numel())`): an entry of a tuple tolerance may be a Python list,
    a tuple of numbers or a numpy array — r05's kernel path recognised only tensors and raised ValueError (VERDICT r05,
    `tools/fuzz_vs_reference.py vectol 9157`).  Same evaluation count and accepted steps as the reference."""
    z = load("tuple_tol.npz")
    dtype = torch.float32 if dname == "f32" else torch.float64
    x0, b0 = T(z[f"ttl_{dname}_x0"], dev), T(z[f"ttl_{dname}_b0"], dev)
    w = torch.tensor([1.0, 3.0, 0.3], dtype=dtype, device=dev)
    t = torch.tensor([0.0, 0.4, 1.1], dtype=dtype, device=dev)
    rtol, atol = TUPLE_TOL_LIST_FORMS[form]
    f = StatFunc(lambda t_, y_: (-y_[0] * w * (1 + 0.2 * t_) + 0.1 * torch.sin(y_[0]), -0.4 * y_[1]))
    with torch.no_grad():
        sa, sb = tda.odeint(f, (x0, b0), t, rtol=rtol, atol=atol, method="dopri5")
    key = f"ttl_{dname}_{form}"
    tol = 1e-11 if dname == "f64" else 2e-5
    assert rel_err(sa, z[f"{key}_ya"]) < tol and rel_err(sb, z[f"{key}_yb"]) < tol
    if dname == "f64":
        assert f.nfe == int(z[f"{key}_nfe"])
        np.testing.assert_allclose(f.accept, z[f"{key}_accept_dt"], rtol=1e-9)


def test_tuple_tolerance_list_entries_through_odeint_adjoint(dev):
    """The same forms through `odeint_adjoint`'s forward solve (adjoint.py:156-223 hands rtol / atol on untouched), with
    the gradients of the reference's backward solve."""
    z = load("tuple_tol.npz")
    x = T(z["ttl_f64_x0"], dev).requires_grad_(True)
    wp = torch.tensor([1.0, 3.0, 0.3], dtype=torch.float64, device=dev, requires_grad=True)
    rtol, atol = TUPLE_TOL_LIST_FORMS["both"]
    out = tda.odeint_adjoint(lambda t_, y_: (-y_[0] * wp * (1 + 0.2 * t_) + 0.1 * torch.sin(y_[0]), -0.4 * y_[1]),
                             (x, T(z["ttl_f64_b0"], dev)), torch.tensor([0.0, 0.4, 1.1], dtype=torch.float64, device=dev),
                             rtol=rtol, atol=atol, adjoint_rtol=1e-8, adjoint_atol=1e-10, adjoint_params=(wp,))
    out[0][-1].pow(2).sum().backward()
    assert rel_err(out[0], z["ttl_adj_ya"]) < 1e-11
    assert rel_err(x.grad, z["ttl_adj_gx"]) < 1e-7 and rel_err(wp.grad, z["ttl_adj_gw"]) < 1e-7


@pytest.mark.parametrize("tup", [False, True], ids=["tensor", "tuple"])
@pytest.mark.parametrize("norm_tag", ["default", "usernorm", "usernorm_semi", "semi"])
def test_adjoint_time_dependent_field_and_user_norms(dev, tup, norm_tag):
    """Adjoint of a time-dependent field against the reference (tests/golden/adjoint_tdep.npz): the backward solve
    takes the reference's steps — same accepted / rejected step sizes, same NFE — because (i) the time VJP is formed
    in every backward evaluation and enters the norms even when `t` needs no gradient, as in the reference, and
    (ii) a user forward `norm` is used by the initial-step heuristic and on (y, adj_y) of the backward solve."""
    z = load("adjoint_tdep.npz")
    lin = torch.nn.Linear(4, 4).double()
    with torch.no_grad():
        lin.weight.copy_(T(z["adjt_W"], dev))
        lin.bias.copy_(T(z["adjt_b"], dev))
    acc, rej, nfe = [], [], [0]

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = lin

        def forward(self, t_, y_):
            nfe[0] += 1
            if tup:
                return torch.tanh(self.lin(y_[0])) * torch.cos(t_), -y_[1] * 0.3 * t_
            return torch.tanh(self.lin(y_)) * torch.cos(t_)

        def callback_accept_step_adjoint(self, t0, y0, dt):
            acc.append(float(dt))

        def callback_reject_step_adjoint(self, t0, y0, dt):
            rej.append(float(dt))

    f = F()
    a0 = T(z["adjt_ya"], dev).clone().requires_grad_(True)
    b0 = T(z["adjt_yb"], dev).clone().requires_grad_(True)
    t = T(z["adjt_t"], dev)
    opts = aopts = None
    if norm_tag.startswith("usernorm"):
        opts = dict(norm=(lambda y: max(y[0].abs().max(), y[1].abs().max())) if tup else (lambda y: y.abs().max()))
    if norm_tag.endswith("semi"):
        aopts = dict(norm="seminorm")
    out = tda.odeint_adjoint(f, (a0, b0) if tup else a0, t, rtol=1e-6, atol=1e-8, method="dopri5", options=opts,
                             adjoint_options=aopts)
    o = out[0] if tup else out
    nfe_fwd = nfe[0]
    (o[-1].pow(2).sum() + o[1].sum() + (out[1][-1].sum() if tup else 0.0)).backward()
    key = f"adjt_{'tup' if tup else 'ten'}_{norm_tag}"
    assert [nfe_fwd, nfe[0] - nfe_fwd] == z[f"{key}_nfe"].tolist()
    assert len(acc) == len(z[f"{key}_acc"]) and len(rej) == len(z[f"{key}_rej"])
    # the embedded error is a cancelling sum (|err| ~ 1e-8 |k|): its association moves the ratio by ~1e-8 relative
    assert np.allclose(acc, z[f"{key}_acc"], rtol=1e-6, atol=0)
    assert rel_err(o.detach(), z[f"{key}_y"]) < 1e-12
    assert rel_err(a0.grad, z[f"{key}_g_y0"]) < 1e-11
    assert rel_err(lin.weight.grad, z[f"{key}_g_W"]) < 1e-11
    assert rel_err(lin.bias.grad, z[f"{key}_g_b"]) < 1e-11
