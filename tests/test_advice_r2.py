"""Round-2 advisor findings, pinned."""
import functools
import warnings

import pytest
import torch

import torchdiffeq_amd as tda


# -- 1. odeint_event: no forward-pass cost / failure for event functions without a graph -------------------------------
def test_event_fn_without_a_graph_and_forward_only_calls(cpu_backend):
    """`event_fn=lambda t, y: y[0].detach()`-style functions (detached, integer / boolean based) used to raise
    'element 0 of tensors does not require grad' in odeint_event's forward pass; forward-only and no_grad callers paid
    an extra func evaluation.  The reference differentiates event_fn only in backward (odeint.py:195-231)."""
    nfe = [0]

    def f(t, y):
        nfe[0] += 1
        return torch.stack([torch.ones_like(y[0]), -0.7 * torch.ones_like(y[1])])
    y0 = torch.tensor([0.0, 1.0], dtype=torch.float64)
    t0 = torch.tensor(0.0, dtype=torch.float64)
    # y[1] = 1 - 0.7 t crosses zero at t = 1/0.7
    with torch.no_grad():
        et, sol = tda.odeint_event(f, y0, t0, event_fn=lambda t, y: y[1].detach(), method="dopri5", atol=1e-9, rtol=1e-7)
    n_plain = nfe[0]
    assert abs(float(et) - 1.0 / 0.7) < 1e-6
    nfe[0] = 0
    with torch.no_grad():
        et2, _ = tda.odeint(f, y0, torch.stack([t0, t0 + 1.0]), event_fn=lambda t, y: y[1].detach(), method="dopri5",
                            atol=1e-9, rtol=1e-7)
    assert nfe[0] == n_plain and float(et2) == float(et)          # odeint_event adds no evaluation of func
    # grad mode, state requires grad, event function WITHOUT a graph: no failure, gradient of the state flows
    y0g = y0.clone().requires_grad_(True)
    et3, sol3 = tda.odeint_event(f, y0g, t0, event_fn=lambda t, y: y[1].detach(), method="dopri5", atol=1e-9, rtol=1e-7)
    sol3[-1].sum().backward()
    assert float(et3.detach()) == pytest.approx(float(et), abs=1e-12) and torch.isfinite(y0g.grad).all()


# -- 2. captured-step cache key: callable objects, containers ----------------------------------------------------------
def test_cache_key_sees_callable_objects_and_container_attributes():
    from torchdiffeq_amd.solvers import _held_tensor_ptrs, _reusable_across_solves
    W, V, U = torch.randn(3, 3), torch.randn(3), torch.randn(2)

    class Field:                                   # not an nn.Module: a class with __call__ holding tensors
        def __init__(self):
            self.W, self.parts, self.table = W, [V], {"u": U}

        def __call__(self, t, y):
            return y @ self.W + self.parts[0]
    f = Field()
    key = _held_tensor_ptrs(f)
    assert {W.data_ptr(), V.data_ptr(), U.data_ptr()} <= set(key) and _reusable_across_solves(f)
    f.W = torch.randn(3, 3)                        # re-bound attribute: the key changes
    assert _held_tensor_ptrs(f) != key
    key = _held_tensor_ptrs(f)
    f.parts[0] = torch.randn(3)                    # ... also inside a list
    assert _held_tensor_ptrs(f) != key

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ws = [torch.ones(3)]              # a plain list attribute of a Module

        def forward(self, t, y):
            return y * self.ws[0]
    m = M()
    assert m.ws[0].data_ptr() in _held_tensor_ptrs(m)

    class Opaque:                                  # nothing discoverable: no reuse across solves (captured per solve)
        __slots__ = ("_w",)

        def __init__(self):
            self._w = W

        def __call__(self, t, y):
            return y @ self._w
    o = Opaque()
    assert _held_tensor_ptrs(o) == () and not _reusable_across_solves(o)
    assert _reusable_across_solves(lambda t, y: y) and _reusable_across_solves(functools.partial(lambda t, y, a: y, a=1))
    assert _reusable_across_solves(torch.tanh) and _held_tensor_ptrs(torch.tanh) == ()


# -- 3. functional_call proxy: verified before it is trusted -----------------------------------------------------------
class _AliasField(torch.nn.Module):
    """Reaches its second layer through a Python list holding the SAME Parameter objects: functional_call does not
    re-route that path."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(4, 4).double()
        self.b = torch.nn.Linear(4, 4).double()
        self.alias = [self.b.weight, self.b.bias]

    def forward(self, t, y):
        h = torch.tanh(self.a(y))
        return torch.nn.functional.linear(h, self.alias[0], self.alias[1])


def test_proxy_is_checked_against_the_direct_vjps(cpu_backend):
    from torchdiffeq_amd.adjoint import _AugmentedDynamics
    from torchdiffeq_amd.misc import OdeFunc, StateLayout
    torch.manual_seed(0)
    for field, expect in ((_AliasField(), False), (torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Tanh()).double(), True)):
        func = field if isinstance(field, _AliasField) else (lambda m: type("F", (torch.nn.Module,), {
            "__init__": lambda self: (torch.nn.Module.__init__(self), setattr(self, "m", m))[0],
            "forward": lambda self, t, y: self.m(y)})())(field)
        params = tuple(func.parameters())
        lay = StateLayout([torch.Size((5, 4))], False)
        fwd = OdeFunc(func, lay, 1.0, torch.float64, torch.device("cpu"))
        shapes = [torch.Size(())] + lay.shapes + lay.shapes + [p.shape for p in params]
        aug_lay = StateLayout(shapes, True, chunk=lay.chunk)
        aug = torch.randn(aug_lay.total, dtype=torch.float64)
        dyn = _AugmentedDynamics(fwd, aug_lay, params, False)
        assert dyn.proxy_names is not None
        assert dyn.proxy_is_faithful(torch.tensor(0.3, dtype=torch.float64), aug) is expect


def test_alias_mode_reroutes_the_parameters_of_any_callable(cpu_backend):
    """r05: a func that is not an nn.Module owning its parameters (a closure, explicit `adjoint_params`) is evaluated
    under `adjoint._AliasParams` for a captured backward step: every torch call that would receive a parameter gets its
    leaf alias, so the aliased VJPs exist and equal the direct ones — unless func holds a pre-computed VIEW of a
    parameter, which the probe finds out."""
    from torchdiffeq_amd.adjoint import _AugmentedDynamics
    from torchdiffeq_amd.misc import OdeFunc, StateLayout
    torch.manual_seed(0)
    W1 = torch.randn(4, 8, dtype=torch.float64, requires_grad=True)
    W2 = torch.randn(8, 4, dtype=torch.float64, requires_grad=True)
    held = {"w": [W2]}
    W2t = W2.t()
    for field, expect in ((lambda t, y: torch.tanh(y @ W1) @ held["w"][0] * torch.cos(t), True),
                          (lambda t, y: torch.tanh(torch.matmul(y, W1)).matmul(W2) + t, True),
                          (lambda t, y: torch.tanh(y @ W1) @ W2t.t(), False)):
        lay = StateLayout([torch.Size((5, 4))], False)
        fwd = OdeFunc(field, lay, 1.0, torch.float64, torch.device("cpu"))
        shapes = [torch.Size(())] + lay.shapes + lay.shapes + [W1.shape, W2.shape]
        aug_lay = StateLayout(shapes, True, chunk=lay.chunk)
        aug = torch.randn(aug_lay.total, dtype=torch.float64)
        dyn = _AugmentedDynamics(fwd, aug_lay, (W1, W2), False)
        assert dyn.proxy_names is None
        assert dyn.proxy_is_faithful(torch.tensor(0.3, dtype=torch.float64), aug) is expect
        if expect:      # and the aliased evaluation leaves the parameters' own graph alone: nothing accumulates into .grad
            _, grads = dyn._vjps(torch.tensor(0.3, dtype=torch.float64), aug, True)
            assert W1.grad is None and W2.grad is None and all(g is not None for g in grads[2:])


@pytest.mark.gpu
def test_unfaithful_proxy_falls_back_to_eager_with_correct_gradients():
    torch.manual_seed(0)
    f = _AliasField().cuda()
    y0 = torch.randn(16, 4, dtype=torch.float64, device="cuda")
    t = torch.tensor([0.0, 1.0], dtype=torch.float64, device="cuda")
    grads = []
    for opts in (None, dict(hip_graph=True)):
        f.zero_grad()
        x = y0.clone().requires_grad_(True)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            tda.odeint_adjoint(f, x, t, options=opts, rtol=1e-8, atol=1e-10)[-1].pow(2).sum().backward()
        if opts:
            assert any("cannot be re-routed to" in str(m.message) for m in w)
        grads.append([p.grad.clone() for p in f.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b) and a.abs().max() > 0          # in particular: the aliased layer's gradients are not zero


# -- 4. a 0-dim output on another device -------------------------------------------------------------------------------
@pytest.mark.gpu
def test_zero_dim_cpu_output_for_a_gpu_state():
    y0 = torch.ones(5, 3, device="cuda")
    t = torch.tensor([0.0, 1.0], device="cuda")
    with torch.no_grad():
        y = tda.odeint(lambda tt, yy: torch.tensor(0.5), y0, t, method="rk4")       # 0-dim CPU tensor: broadcasts
        ref = tda.odeint(lambda tt, yy: torch.full_like(yy, 0.5), y0, t, method="rk4")
    assert torch.equal(y, ref)


# -- 5. hip_graph warning: names lock-step sharding, silent for the environment default --------------------------------
def test_hip_graph_warning_text_and_env_default(cpu_backend, monkeypatch):
    from torchdiffeq_amd.misc import OdeFunc, StateLayout, rms_norm
    from torchdiffeq_amd.solvers import Dopri5Solver
    y0 = torch.ones(8, dtype=torch.float64)
    func = OdeFunc(lambda t, y: -y, StateLayout([y0.shape], False), 1.0, y0.dtype, y0.device)
    with pytest.warns(UserWarning, match="lock-step sharding"):
        Dopri5Solver(func=func, y0=y0, rtol=1e-6, atol=1e-8, norm=rms_norm, hip_graph=True)
    monkeypatch.setenv("TDEQ_HIP_GRAPH", "1")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        Dopri5Solver(func=func, y0=y0, rtol=1e-6, atol=1e-8, norm=rms_norm)          # env default: applies where it can, silently
