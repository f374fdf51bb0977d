"""The `SOLVERS` plugin protocol (reference odeint.py:19-46, :92): a solver class that is NOT this package's
— written against the reference's protocol `cls(func=, y0=, rtol=, atol=, **options).integrate(t)`, doing its own
arithmetic with torch ops, calling `func(t, y, perturb=<its own enum>)`, `norm(y)` and `func.callback_step` — can be
registered in `torchdiffeq_amd.SOLVERS` and used through `odeint` / `odeint_adjoint`.  (In the build container the
reference's own 24 classes were registered this way and reproduce the reference bit for bit for tensor and tuple states,
both time directions and per-component tolerances — r03; here a small class written for the test stands in for them.)

What the package has to get right for such a class: the flat state of a TUPLE is padded per component, so the default
norm must be handed over as a callable over the components and per-component tolerances per element (misc.py:237-254 in
the reference), the foreign `Perturb` enum is matched by name, time reversal stays inside `func`."""
import enum

import pytest
import torch

import torchdiffeq_amd as tda


class _ForeignPerturb(enum.Enum):       # a copy of the enum, as a third-party module importing another package has
    NONE = 0
    PREV = 1
    NEXT = 2


class ForeignHeunEuler:
    """Embedded Heun(2)/Euler(1) pair, steps clipped to the output times; the reference's protocol and nothing else."""
    order = 2
    seen = []

    def __init__(self, func, y0, rtol, atol, norm, first_step=0.05, **unused):
        self.func, self.y0, self.norm, self.first_step = func, y0, norm, first_step
        self.rtol = torch.as_tensor(rtol, dtype=y0.dtype, device=y0.device)
        self.atol = torch.as_tensor(atol, dtype=y0.dtype, device=y0.device)
        type(self).seen.append(dict(norm=norm, rtol=rtol, atol=atol, n=y0.numel()))

    @classmethod
    def valid_callbacks(cls):
        return {"callback_step"}

    def integrate(self, t):
        func, y = self.func, self.y0
        sol = [y]
        dt = torch.as_tensor(self.first_step, dtype=t.dtype, device=t.device)
        for t0, t1 in zip(t[:-1], t[1:]):
            cur = t0
            while cur < t1:
                h = torch.minimum(dt, t1 - cur)
                func.callback_step(cur, y, h)
                k1 = func(cur, y, perturb=_ForeignPerturb.NEXT)
                k2 = func(cur + h, y + h * k1, perturb=_ForeignPerturb.PREV)
                y1 = y + 0.5 * h * (k1 + k2)
                err = 0.5 * h * (k2 - k1)
                ratio = self.norm(err / (self.atol + self.rtol * torch.maximum(y.abs(), y1.abs())))
                if ratio <= 1:
                    cur, y = cur + h, y1
                dt = h * torch.clamp(0.9 / torch.sqrt(ratio + 1e-30), 0.2, 5.0)
            sol.append(y)
        return torch.stack(sol)


@pytest.fixture()
def plugin():
    tda.SOLVERS["foreign_heun_euler"] = ForeignHeunEuler
    ForeignHeunEuler.seen.clear()
    yield "foreign_heun_euler"
    del tda.SOLVERS["foreign_heun_euler"]


def _field(A):
    return lambda t_, s: (torch.tanh(s[0] @ A) * torch.cos(t_), -0.5 * s[1] * (1 + t_))


def _standalone(A, y0a, y0b, t, rtol, atol):
    """The same class driven by hand the way the REFERENCE's odeint would drive it: unpadded concatenation, mixed norm
    over the components, tolerances per element, time negated for a decreasing grid."""
    na = y0a.numel()
    sign = -1.0 if t[0] > t[1] else 1.0

    class Flat:
        callback_step = staticmethod(lambda *a: None)

        def __call__(self, t_, y, perturb=None):
            if perturb is _ForeignPerturb.NEXT:         # misc.py:185-196: the evaluation time moves by one ulp
                t_ = torch.nextafter(t_, t_ + 1)
            elif perturb is _ForeignPerturb.PREV:
                t_ = torch.nextafter(t_, t_ - 1)
            fa, fb = _field(A)(t_ * sign, (y[:na].view(y0a.shape), y[na:].view(y0b.shape)))
            return torch.cat([fa.reshape(-1), fb.reshape(-1)]) * sign
    norm = lambda y: max(y[:na].abs().pow(2).mean().sqrt(), y[na:].abs().pow(2).mean().sqrt())
    # misc.py:115-123 `_tuple_tol`: torch.as_tensor(<python float>) is fp32 — the reference's per-component tolerances
    # pass through single precision, and so do this package's
    expand = lambda tol: tol if not isinstance(tol, tuple) else torch.cat(
        [torch.as_tensor(tol[0]).expand(na), torch.as_tensor(tol[1]).expand(y0b.numel())]).to(y0a.device)
    out = ForeignHeunEuler(Flat(), torch.cat([y0a.reshape(-1), y0b.reshape(-1)]), expand(rtol), expand(atol), norm).integrate(t * sign)
    return out[:, :na].view(len(t), *y0a.shape), out[:, na:].view(len(t), *y0b.shape)


@pytest.mark.parametrize("tols", [(1e-4, 1e-6), ((1e-3, 1e-6), (1e-5, 1e-8))], ids=["scalar-tol", "tuple-tol"])
@pytest.mark.parametrize("reverse", [False, True], ids=["fwd", "rev"])
def test_foreign_solver_class_on_a_tuple_state(dev, plugin, reverse, tols):
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)
    y0a = torch.tensor([[2.0, 0.0], [1.0, 0.5], [0.3, -0.7]], dtype=torch.float64)
    y0b = torch.tensor([0.4, -1.1, 2.0, 0.1, 0.9], dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 0.7, 1.0], dtype=torch.float64)
    if reverse:
        t = t.flip(0)
    rtol, atol = tols
    steps = []

    class F(torch.nn.Module):
        def forward(self, t_, s):
            return _field(A)(t_, s)

        def callback_step(self, t0, y0_, dt):
            assert isinstance(y0_, tuple) and y0_[0].shape == y0a.shape and y0_[1].shape == y0b.shape
            steps.append(float(t0))
    with torch.no_grad():
        got = tda.odeint(F(), (y0a, y0b), t, rtol=rtol, atol=atol, method=plugin)
        want = _standalone(A, y0a, y0b, t, rtol, atol)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # the class got a norm it can call and tolerances it can broadcast against the (padded) flat state
    seen = ForeignHeunEuler.seen[0]
    assert callable(seen["norm"]) and not isinstance(seen["norm"], tda.misc.BuiltinNorm)
    if isinstance(rtol, tuple):
        assert torch.is_tensor(seen["rtol"]) and seen["rtol"].numel() == seen["n"]
    assert steps and (steps == sorted(steps, reverse=reverse))        # callbacks see user time


def test_foreign_solver_class_in_the_adjoint(dev, plugin):
    """Forward and backward solve of odeint_adjoint through the foreign class; gradients against the package's own
    solver at tight tolerances."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(2, 2).double()

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l = lin

        def forward(self, t_, s):
            return torch.tanh(self.l(s[0])) * torch.cos(t_), -0.3 * s[1]
    y0b = torch.tensor([0.4, -1.1, 2.0], dtype=torch.float64)
    t = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64)
    res = []
    for method, kw in ((plugin, dict(rtol=1e-7, atol=1e-9)), ("dopri5", dict(rtol=1e-9, atol=1e-11))):
        for p in lin.parameters():
            p.grad = None
        x = torch.tensor([[2.0, 0.0], [1.0, 0.5]], dtype=torch.float64, requires_grad=True)
        out = tda.odeint_adjoint(F(), (x, y0b), t, method=method, **kw)
        (out[0][-1].pow(2).sum() + out[1][-1].sum()).backward()
        res.append([out[0].detach(), x.grad.clone()] + [p.grad.clone() for p in lin.parameters()])
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    assert len(ForeignHeunEuler.seen) >= 3            # one forward solve + one backward solve per output interval


def test_foreign_solver_class_on_a_tensor_state_reversed(dev, plugin):
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)
    y0 = torch.tensor([[2.0, 0.0], [1.0, 0.5]], dtype=torch.float64)
    t = torch.tensor([1.0, 0.4, 0.0], dtype=torch.float64)
    with torch.no_grad():
        got = tda.odeint(lambda t_, y: torch.tanh(y @ A) * torch.cos(t_), y0, t, rtol=1e-5, atol=1e-7, method=plugin)
        ref = tda.odeint(lambda t_, y: torch.tanh(y @ A) * torch.cos(t_), y0, t, rtol=1e-10, atol=1e-12, method="dopri5")
    assert got.shape == (3, 2, 2) and torch.equal(got[0], y0)
    assert torch.allclose(got, ref, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("direction", ["fwd", "rev"])
def test_user_defined_tableau_on_the_native_solver(dev, direction):
    """Adding an explicit embedded Runge–Kutta method is adding a coefficient table: `tableaus.Tableau` + a two-line
    subclass of `RKAdaptiveStepsizeODESolver` — the same generic kernels, look-ahead controller and dense output run it.
    Cash–Karp 5(4) (in neither library's table) against the reference's adaptive machinery given the same table
    (tests/golden/dropin.npz <- make_golden.py dropin: rk_common.py:15, :153-211): same evaluations, same accepted
    steps, same solution."""
    from _cases import T, load
    from torchdiffeq_amd.solvers import RKAdaptiveStepsizeODESolver
    from torchdiffeq_amd.tableaus import Tableau
    z = load("dropin.npz")
    beta = tuple(tuple(float(v) for v in z[f"cashkarp_beta{i}"]) for i in range(6))
    tab = Tableau("cashkarp", 5, tuple(z["cashkarp_alpha"].tolist()), beta, tuple(z["cashkarp_c_sol"].tolist()),
                  tuple(z["cashkarp_c_err"].tolist()), tuple(z["cashkarp_mid"].tolist()))

    class CashKarp(RKAdaptiveStepsizeODESolver):
        order = 5
        tableau = tab
    tda.SOLVERS["cashkarp"] = CashKarp
    try:
        A, y0, t = T(z["cashkarp_A"]).to(dev), T(z["cashkarp_y0"]).to(dev), T(z[f"cashkarp_{direction}_t"]).to(dev)
        accepted = []

        class F(torch.nn.Module):
            nfe = 0

            def forward(self, t_, y):
                self.nfe += 1
                return torch.tanh(y @ A) * torch.cos(t_)

            def callback_accept_step(self, t0, y_, dt):
                accepted.append(float(dt))
        f = F()
        with torch.no_grad():
            y = tda.odeint(f, y0, t, method="cashkarp", rtol=1e-8, atol=1e-10)
        assert f.nfe == int(z[f"cashkarp_{direction}_nfe"])
        assert len(accepted) == len(z[f"cashkarp_{direction}_accept_dt"])
        assert torch.allclose(torch.tensor(accepted, dtype=torch.float64),
                              torch.as_tensor(z[f"cashkarp_{direction}_accept_dt"], dtype=torch.float64), rtol=1e-6)
        assert float((y.cpu() - T(z[f"cashkarp_{direction}_y"])).abs().max()) < 1e-11
    finally:
        del tda.SOLVERS["cashkarp"]


def test_user_tableau_with_an_all_zero_error_row(dev):
    """Found by the random-tableau fuzz (tools/fuzz_vs_reference.py tableau): an error row that estimates nothing is a
    legal table — the reference's dense sum gives 0, every step is accepted and grows by `ifactor` (misc.py:86-95).  The
    sparse rows keep one explicit zero term so that the kernels, which take >= 1 term, compute the same 0 * k_0."""
    from torchdiffeq_amd.solvers import RKAdaptiveStepsizeODESolver
    from torchdiffeq_amd.tableaus import SparseRow, Tableau
    assert SparseRow.from_dense((0.0, 0.0, 0.0)) == SparseRow((0,), (0.0,))
    b = (0.34516902151778417, 0.6548309784822158)
    tab = Tableau("zeroerr", 2, (0.3, 1.0), ((0.3,), b), b + (0.0,), (0.0, 0.0, 0.0), (0.2975845107588921, 0.3274154892411079, -0.125))

    class ZeroErr(RKAdaptiveStepsizeODESolver):
        order = 2
        tableau = tab
    tda.SOLVERS["zeroerr"] = ZeroErr
    try:
        steps = []

        class F(torch.nn.Module):
            def forward(self, t_, y):
                return -y * (1 + 0.3 * t_)

            def callback_accept_step(self, t0, y_, dt):
                steps.append(float(dt))

            def callback_reject_step(self, t0, y_, dt):
                raise AssertionError("a zero error estimate never rejects")
        with torch.no_grad():
            y = tda.odeint(F(), torch.linspace(0.5, 2.0, 17, dtype=torch.float64), torch.tensor([0.0, 0.3, 1.0], dtype=torch.float64),
                           method="zeroerr", options=dict(first_step=0.02))
        assert torch.isfinite(y).all() and y.shape == (3, 17)
        assert steps[0] == pytest.approx(0.02) and all(b_ == pytest.approx(10 * a_) for a_, b_ in zip(steps, steps[1:]))
    finally:
        del tda.SOLVERS["zeroerr"]
