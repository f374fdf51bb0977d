"""Second family of program-level differential cases against the reference (build container only).

  PYTHONDONTWRITEBYTECODE=1 python tools/fuzz_api_programs_vs_reference.py [seed] [cases] [only]

What tools/fuzz_programs_vs_reference.py leaves out: step / accept / reject callbacks (forward and adjoint) logged with
their arguments, event solves differentiated through `odeint_adjoint` (event time AND end state in the loss), plain
callables with explicit `adjoint_params`, tuple states of unequal shapes with tupled tolerances, complex and bf16
states, `odeint_dense`.  Everything on CPU tensors (the package's host path), compared with `torch.equal`."""
import copy
import random
import sys
import warnings

import torch
from torch import nn

import os  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torchdiffeq_amd as tda  # noqa: E402

# TDEQ_FUZZ_DEVICE=cuda (GPU box, no reference there): the package's CPU host path (= the reference's arithmetic, which
# the default mode of this tool establishes) against the HIP kernels on the device, to tolerance.
DEVICE = os.environ.get("TDEQ_FUZZ_DEVICE")
if DEVICE:
    ref = tda
else:
    sys.path.insert(0, "/root/reference")
    import torchdiffeq as ref  # noqa: E402
TARGET = "cpu"      # where a program puts its tensors (switched by the device mode)


def D(x):
    return x.to(TARGET)

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
rng = random.Random(seed)
ADAPTIVE = ["dopri5", "dopri8", "bosh3", "tsit5", "fehlberg2", "adaptive_heun"]
FIXED = ["euler", "midpoint", "heun2", "heun3", "rk4"]


def digest(x):
    if isinstance(x, (tuple, list)):
        return tuple(digest(c) for c in x)
    if torch.is_tensor(x):
        return x.detach().clone()
    return x


class CallbackField(nn.Module):
    def __init__(self, dim, g, dtype, names):
        super().__init__()
        self.lin = nn.Linear(dim, dim)
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.6
        self.to(dtype)
        self.log = []
        for n in names:
            setattr(self, n, self._make(n))

    def _make(self, name):
        def cb(*args):
            self.log.append((name, digest(args)))
        return cb

    def forward(self, t, y):
        self.log.append(("f", digest(t)))
        return torch.tanh(self.lin(y)) * (1.5 - t)


def gen(rng):
    return torch.Generator().manual_seed(rng.randrange(10 ** 6))


def times(rng, g, npts, dtype):
    t = torch.sort(torch.rand(npts, generator=g, dtype=torch.float64) * rng.choice([0.5, 1.0, 2.0])).values
    if float((t[1:] - t[:-1]).min()) < 5e-3:
        t = torch.linspace(0, 1, npts, dtype=torch.float64)
    t = t.to(dtype)
    return t.flip(0) if rng.random() < 0.35 else t


def tol(rng):
    return dict(rtol=rng.choice([1e-3, 1e-5, 1e-7]), atol=rng.choice([1e-4, 1e-6, 1e-9])) if rng.random() < 0.8 else {}


# ---- case families: each returns (description, program) with program(lib) -> list of (name, value) ----------------
def callbacks_case(rng):
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    names = [n for n in ("callback_step", "callback_accept_step", "callback_reject_step",
                         "callback_step_adjoint", "callback_accept_step_adjoint", "callback_reject_step_adjoint")
             if rng.random() < 0.6]
    dim = rng.choice([1, 3])
    field = CallbackField(dim, g, dtype, names)
    method = rng.choice(ADAPTIVE + FIXED)
    y0 = torch.randn(rng.choice([1, 3]), dim, generator=g, dtype=torch.float64).to(dtype)
    t = times(rng, g, rng.choice([2, 4]), dtype)
    kw = tol(rng)
    if method in FIXED and rng.random() < 0.6:
        kw["options"] = dict(step_size=rng.choice([0.07, 0.2]))
    api = rng.choice(["odeint", "odeint_adjoint"])
    tup = rng.random() < 0.3

    def program(lib):
        f = copy.deepcopy(field).to(TARGET)
        f.log = []
        for n in names:
            setattr(f, n, f._make(n))
        if tup:
            ff = lambda t_, y_: (f(t_, y_[0]), -y_[1] * 0.3)  # noqa: E731
            wrap = nn.Module()
            wrap.inner = f
            wrap.forward = ff
            for n in names:
                setattr(wrap, n, getattr(f, n))
            state = (D(y0).clone().requires_grad_(True), torch.ones(2, dtype=dtype, device=TARGET))
            sol = getattr(lib, api)(wrap, state, D(t), method=method, **kw)[0]
        else:
            sol = getattr(lib, api)(f, D(y0).clone().requires_grad_(True), D(t), method=method, **kw)
        out = [("sol", sol.detach().clone())]
        if api == "odeint_adjoint":
            sol[-1].sum().backward()
            out += [("g:" + n, p.grad.clone()) for n, p in f.named_parameters()]
        out.append(("calls", len(f.log)))
        out += [(f"log{i}:{e[0]}", e[1]) for i, e in enumerate(f.log)]
        return out
    return f"callbacks {api} {method} {str(dtype)[6:]} {names} tuple={tup} {kw}", program


def event_grad_case(rng):
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    method = rng.choice(ADAPTIVE + ["rk4", "midpoint"])
    grav = nn.Parameter(torch.tensor([9.8 * rng.choice([0.5, 1.0])], dtype=dtype))
    drag = nn.Parameter(torch.tensor([rng.choice([0.0, 0.1])], dtype=dtype))
    pos0 = torch.tensor([rng.choice([5.0, 10.0])], dtype=dtype)
    vel0 = torch.tensor([rng.choice([0.0, 1.0, -1.0])], dtype=dtype)
    tupled = rng.random() < 0.5
    interface = rng.choice(["odeint", "odeint_adjoint"])
    kw = tol(rng)
    opts = dict(step_size=0.01) if method in ("rk4", "midpoint") else {}
    grad_t0 = rng.random() < 0.4

    class Ball(nn.Module):
        def __init__(self):
            super().__init__()
            self.g = nn.Parameter(grav.detach().clone())
            self.k = nn.Parameter(drag.detach().clone())

        def forward(self, t, s):
            if tupled:
                p, v = s
                return v, -self.g - self.k * v
            return torch.stack([s[1], (-self.g - self.k * s[1:2]).squeeze(0)])

    def program(lib):
        f = Ball().to(TARGET)
        p0 = D(pos0).clone().requires_grad_(True)
        v0 = D(vel0).clone().requires_grad_(True)
        t0 = torch.tensor(0.0, dtype=dtype, device=TARGET, requires_grad=grad_t0)
        state = (p0, v0) if tupled else torch.cat([p0, v0])
        ev = (lambda t, s: s[0]) if tupled else (lambda t, s: s[0:1])
        et, es = lib.odeint_event(f, state, t0, event_fn=ev, method=method, options=opts or None,
                                  odeint_interface=getattr(lib, interface), **kw)
        end = es[0][-1] if tupled else es[-1]
        endv = es[1][-1] if tupled else es[-1]
        loss = et * 1.5 + endv.sum() * 0.1 + end.sum()
        loss.backward()
        out = [("event_t", et.detach().clone()), ("end", digest(es))]
        out += [("g:" + n, None if p.grad is None else p.grad.clone()) for n, p in f.named_parameters()]
        out += [("g:p0", p0.grad), ("g:v0", v0.grad)]
        if grad_t0:
            out.append(("g:t0", t0.grad))
        return out
    exact = interface == "odeint_adjoint"
    return f"event_grad {interface} {method} {str(dtype)[6:]} tupled={tupled} t0grad={grad_t0} {kw}", program, exact


def explicit_params_case(rng):
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    dim = rng.choice([2, 4])
    W = (torch.randn(dim, dim, generator=g) * 0.5).to(dtype)
    b = (torch.randn(dim, generator=g) * 0.2).to(dtype)
    unused = torch.randn(3, generator=g).to(dtype)
    method = rng.choice(ADAPTIVE + FIXED)
    y0 = torch.randn(3, dim, generator=g, dtype=torch.float64).to(dtype)
    t = times(rng, g, rng.choice([2, 3, 5]), dtype)
    kw = tol(rng)
    which = rng.choice(["both", "W", "none", "with_unused", "frozen"])
    norm = rng.choice([None, "seminorm"])

    def program(lib):
        w_ = D(W).clone().requires_grad_(True)
        b_ = D(b).clone().requires_grad_(which != "frozen")
        u_ = D(unused).clone().requires_grad_(True)
        params = {"both": (w_, b_), "W": (w_,), "none": (), "with_unused": (w_, u_, b_), "frozen": (w_, b_)}[which]
        f = lambda t_, y_: torch.sin(y_ @ w_.t() + b_) * torch.exp(-t_)  # noqa: E731
        k = dict(kw)
        if norm:
            k["adjoint_options"] = dict(norm=norm)
        y = D(y0).clone().requires_grad_(True)
        sol = lib.odeint_adjoint(f, y, D(t), method=method, adjoint_params=params, **k)
        (sol ** 2).sum().backward()
        return [("sol", sol.detach().clone()), ("gW", w_.grad), ("gb", b_.grad), ("gu", u_.grad), ("gy", y.grad)]
    return f"explicit_params {which} {method} {str(dtype)[6:]} norm={norm} {kw}", program


def ragged_tuple_case(rng):
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    method = rng.choice(ADAPTIVE + FIXED)
    a0 = torch.randn(2, 3, generator=g, dtype=torch.float64).to(dtype)
    b0 = torch.randn(5, generator=g, dtype=torch.float64).to(dtype)
    c0 = torch.randn((), generator=g, dtype=torch.float64).to(dtype)
    t = times(rng, g, rng.choice([2, 4]), dtype)
    kw = {}
    r = rng.random()
    if r < 0.4:
        kw = dict(rtol=(1e-4, 1e-6, 1e-5), atol=(1e-6, 1e-8, 1e-7))
    elif r < 0.7:
        kw = tol(rng)
    api = rng.choice(["odeint", "odeint_adjoint"])

    class F(nn.Module):
        def __init__(self):
            super().__init__()
            self.m = nn.Parameter((torch.randn(3, 3, generator=gen(random.Random(7))) * 0.4).to(dtype))
            self.s = nn.Parameter(torch.tensor(0.7, dtype=dtype))

        def forward(self, t, y):
            a, b, c = y
            return torch.tanh(a @ self.m) + c, -b * self.s + a.sum() * 0.05, torch.cos(t) * self.s - c

    def program(lib):
        f = F().to(TARGET)
        st = tuple(D(x).clone().requires_grad_(True) for x in (a0, b0, c0))
        sol = getattr(lib, api)(f, st, D(t), method=method, **kw)
        loss = sum((s[-1] ** 2).sum() for s in sol)
        loss.backward()
        out = [(f"sol{i}", s.detach().clone()) for i, s in enumerate(sol)]
        if api == "odeint_adjoint":
            out += [("g:" + n, p.grad.clone()) for n, p in f.named_parameters()]
            out += [(f"gy{i}", x.grad) for i, x in enumerate(st)]
        return out
    return f"ragged_tuple {api} {method} {str(dtype)[6:]} {kw}", program


def odd_dtype_case(rng):
    g = gen(rng)
    dtype = rng.choice([torch.complex64, torch.complex128, torch.bfloat16])
    method = rng.choice(["dopri5", "dopri8", "bosh3", "rk4", "tsit5", "midpoint"])
    real = torch.float32 if dtype in (torch.complex64, torch.bfloat16) else torch.float64
    if dtype.is_complex:
        y0 = torch.complex(torch.randn(2, 3, generator=g), torch.randn(2, 3, generator=g)).to(dtype)
        A = torch.complex(torch.randn(3, 3, generator=g) * 0.4, torch.randn(3, 3, generator=g) * 0.4).to(dtype)
    else:
        y0 = torch.randn(2, 3, generator=g).to(dtype)
        A = (torch.randn(3, 3, generator=g) * 0.4).to(dtype)
    t = times(rng, g, rng.choice([2, 4]), real if dtype != torch.bfloat16 else dtype)
    kw = dict(rtol=1e-2, atol=1e-2) if dtype == torch.bfloat16 else tol(rng)
    api = rng.choice(["odeint", "odeint_adjoint"]) if dtype.is_complex else "odeint"

    def program(lib):
        a_ = D(A).clone().requires_grad_(True)
        f = lambda t_, y_: y_ @ a_ - y_ * 0.5  # noqa: E731
        y = D(y0).clone().requires_grad_(True)
        extra = dict(adjoint_params=(a_,)) if api == "odeint_adjoint" else {}
        sol = getattr(lib, api)(f, y, D(t), method=method, **kw, **extra)
        out = [("sol", sol.detach().clone())]
        if api == "odeint_adjoint":
            sol[-1].abs().sum().backward()
            out += [("gA", a_.grad), ("gy", y.grad)]
        return out
    return f"odd_dtype {api} {method} {str(dtype)[6:]} {kw}", program


def dense_case(rng):
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    y0 = torch.randn(rng.choice([1, 4]), generator=g, dtype=torch.float64).to(dtype)
    A = (torch.randn(len(y0), len(y0), generator=g) * 0.7).to(dtype)
    t0 = torch.tensor(rng.choice([0.0, 0.3]), dtype=dtype)
    t1 = torch.tensor(rng.choice([1.0, 2.5]), dtype=dtype)
    kw = tol(rng)
    qs = [float(t0) + (float(t1) - float(t0)) * rng.random() for _ in range(4)]

    def program(lib):
        A_ = D(A)
        f = lambda t_, y_: torch.sin(A_ @ y_) - 0.1 * y_ * t_  # noqa: E731
        fn = lib.odeint_dense(f, D(y0), D(t0), D(t1), **kw)
        out = []
        for q in qs:
            out.append((f"q{q:.3f}", fn(torch.tensor(q, dtype=dtype, device=TARGET)).detach().clone()))
        return out
    return f"dense {str(dtype)[6:]} {kw}", program


def stochastic_case(rng):
    """A field that draws from torch's global RNG at every evaluation: the results agree bit for bit only if both
    libraries evaluate func the same number of times in the same order — forward, initial-step heuristic, rejected
    steps, interpolation, and every evaluation of the backward solve."""
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    method = rng.choice(ADAPTIVE + FIXED)
    dim = rng.choice([1, 3])
    y0 = torch.randn(2, dim, generator=g, dtype=torch.float64).to(dtype)
    W = (torch.randn(dim, dim, generator=g) * 0.5).to(dtype)
    t = times(rng, g, rng.choice([2, 4]), dtype)
    kw = dict(rtol=rng.choice([1e-2, 1e-3]), atol=rng.choice([1e-3, 1e-4]))
    if method in FIXED:
        kw["options"] = dict(step_size=rng.choice([0.1, 0.26]), interp=rng.choice(["linear", "cubic"]))
    api = rng.choice(["odeint", "odeint_adjoint"])
    noise = rng.choice([1e-3, 1e-2])
    seed_ = rng.randrange(10 ** 6)
    tup = rng.random() < 0.3

    def program(lib):
        torch.manual_seed(seed_)
        w_ = W.clone().requires_grad_(True)
        calls = [0]

        def f(t_, y_):
            calls[0] += 1
            if tup:
                a, b = y_
                return torch.tanh(a @ w_) + noise * torch.randn_like(a), -b + noise * torch.randn_like(b)
            return torch.tanh(y_ @ w_) + noise * torch.randn_like(y_)
        y = y0.clone().requires_grad_(True)
        state = (y, torch.ones(3, dtype=dtype)) if tup else y
        extra = dict(adjoint_params=(w_,)) if api == "odeint_adjoint" else {}
        with torch.no_grad() if api == "odeint" else torch.enable_grad():
            sol = getattr(lib, api)(f, state, t, method=method, **kw, **extra)
        main = sol[0] if tup else sol
        out = [("sol", main.detach().clone()), ("calls_fwd", calls[0])]
        if api == "odeint_adjoint":
            main[-1].sum().backward()
            out += [("gW", w_.grad), ("gy", y.grad), ("calls", calls[0])]
        out.append(("rng_after", torch.rand(1)))
        return out
    return f"stochastic {api} {method} {str(dtype)[6:]} tuple={tup} {kw}", program


def options_case(rng):
    """The adaptive solvers' option pool, several at once (rk_common.py:160-214, 266-361): step-size clamps and factors,
    an evaluation budget that may run out, a user norm, step_t / jump_t (also together, also outside the interval, also
    with decreasing time), through odeint or odeint_adjoint."""
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    method = rng.choice(ADAPTIVE)
    dim = rng.choice([2, 3])
    y0 = torch.randn(2, dim, generator=g, dtype=torch.float64).to(dtype)
    W = (torch.randn(dim, dim, generator=g) * 0.8).to(dtype)
    t = times(rng, g, rng.choice([2, 3, 5]), dtype)
    lo, hi = float(t.min()), float(t.max())
    kw = tol(rng)
    opts = {}
    if rng.random() < 0.3:
        opts["min_step"] = rng.choice([1e-4, 1e-2, 0.2])
    if rng.random() < 0.3:
        opts["max_step"] = rng.choice([0.01, 0.05, 0.3])
    if rng.random() < 0.3:
        opts["ifactor"] = rng.choice([2.0, 5.0])
    if rng.random() < 0.3:
        opts["dfactor"] = rng.choice([0.1, 0.5])
    if rng.random() < 0.3:
        opts["safety"] = rng.choice([0.7, 0.95])
    if rng.random() < 0.2:
        opts["max_num_steps"] = rng.choice([3, 10, 40])
    if rng.random() < 0.25:
        opts["first_step"] = rng.choice([1e-3, 0.05, 5.0])
    if rng.random() < 0.25:
        opts["norm"] = rng.choice(["linf", "l1"])
    pts = lambda k: torch.tensor(sorted(lo - 0.1 + (hi - lo + 0.2) * rng.random() for _ in range(k)), dtype=dtype)  # noqa: E731
    if rng.random() < 0.35:
        opts["step_t"] = pts(rng.choice([1, 3]))
    if rng.random() < 0.35:
        opts["jump_t"] = pts(rng.choice([1, 2]))
    if rng.random() < 0.1:
        opts["dtype"] = rng.choice([torch.float32, torch.float64])
    api = rng.choice(["odeint", "odeint_adjoint"])
    adj_inherits = rng.random() < 0.5

    def program(lib):
        w_ = D(W).clone().requires_grad_(True)
        calls = []

        def f(t_, y_):
            calls.append(float(t_))
            # (discontinuous in t at the jump points, as jump_t is meant for)
            kick = sum((t_ > float(j)).to(y_.dtype) for j in opts.get("jump_t", []))
            return torch.tanh(y_ @ w_) * (1.0 + 0.5 * kick) - 0.2 * y_
        o = {k: (D(v) if torch.is_tensor(v) else v) for k, v in opts.items()}
        if "norm" in o:
            o["norm"] = {"linf": lambda x: x.abs().max(), "l1": lambda x: x.abs().mean()}[o["norm"]]
        y = D(y0).clone().requires_grad_(True)
        extra = {}
        if api == "odeint_adjoint":
            extra["adjoint_params"] = (w_,)
            if not adj_inherits:
                extra["adjoint_options"] = {k: v for k, v in o.items() if k in ("max_num_steps", "safety")}
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            sol = getattr(lib, api)(f, y, D(t), method=method, options=o, **kw, **extra)
        out = [("sol", sol.detach().clone()), ("calls_fwd", len(calls)),
               ("warnings", sorted(str(w.message)[:60] for w in caught
                                   # (not the libraries' own: the package's host-path notice, and a PyTorch notice the reference's
                                   #  `float(tensor)` conversions trigger under autograd)
                                   if "host path" not in str(w.message) and "Converting a tensor" not in str(w.message)))]
        if api == "odeint_adjoint":
            sol[-1].sum().backward()
            out += [("gW", w_.grad), ("gy", y.grad), ("calls", len(calls))]
        out.append(("times", torch.tensor(calls, dtype=torch.float64)))
        return out
    shown = {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in opts.items()}
    return f"options {api} {method} {str(dtype)[6:]} rev={bool(t[0] > t[-1])} {shown} {kw} inherit={adj_inherits}", program


def direct_event_case(rng):
    """`odeint(..., event_fn=...)` called directly (odeint.py:87-101): len(t) must be 2, the result is (event_t, solution);
    a vector-valued event function (combined by `min` of the sign-normalised components); grids for the fixed-grid
    methods from a user `grid_constructor`."""
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    method = rng.choice(ADAPTIVE + FIXED)
    y0 = torch.tensor([rng.choice([1.0, 2.0]), rng.choice([0.0, 0.5])], dtype=dtype)
    rev = rng.random() < 0.3
    npts = rng.choice([2, 2, 3])
    t = torch.tensor([0.0, 10.0, 20.0][:npts], dtype=dtype)
    if rev:
        t = -t
    thr = rng.choice([0.2, 0.5])
    vector_event = rng.random() < 0.5
    opts = {}
    if method in FIXED:
        opts["step_size"] = rng.choice([0.05, 0.02])
        if rng.random() < 0.4:
            opts["interp"] = "cubic"
    kw = tol(rng)
    api = rng.choice(["odeint", "odeint_adjoint"])

    def program(lib):
        k_ = torch.tensor(1.3, dtype=dtype, device=TARGET, requires_grad=True)

        def f(t_, y_):
            return torch.stack([y_[1], -k_ * y_[0] - 0.1 * y_[1]])
        if vector_event:
            ev = lambda t_, y_: torch.stack([y_[0] - thr, y_[0] + 5.0])  # noqa: E731
        else:
            ev = lambda t_, y_: y_[0] - thr  # noqa: E731
        extra = dict(adjoint_params=(k_,)) if api == "odeint_adjoint" else {}
        y = D(y0).clone().requires_grad_(True)
        res = getattr(lib, api)(f, y, D(t), event_fn=ev, method=method, options=dict(opts) or None, **kw, **extra)
        et, sol = res
        out = [("event_t", et.detach().clone()), ("sol", sol.detach().clone())]
        if api == "odeint_adjoint":
            (et + sol[-1].sum()).backward()
            out += [("gk", k_.grad), ("gy", y.grad)]
        return out
    return f"direct_event {api} {method} {str(dtype)[6:]} rev={rev} npts={npts} vector={vector_event} {opts} {kw}", program


def grid_constructor_case(rng):
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    method = rng.choice(FIXED)
    y0 = torch.randn(3, generator=g, dtype=torch.float64).to(dtype)
    t = times(rng, g, rng.choice([2, 4]), dtype)
    extra_pts = rng.choice([1, 4])
    interp = rng.choice(["linear", "cubic"])
    api = rng.choice(["odeint", "odeint_adjoint"])
    bad_grid = rng.random() < 0.15

    def program(lib):
        w_ = torch.tensor([0.5, -0.3, 0.8], dtype=dtype, device=TARGET, requires_grad=True)
        seen = []

        def grid(func, y, tt):
            d = func(tt[0], y)               # a grid constructor may look at the field: func and y0 in the reference's form
            seen.append((tuple(y.shape), tt.detach().clone(), d.detach().clone()))
            inner = [tt[:-1] + (tt[1:] - tt[:-1]) * (i + 1) / (extra_pts + 1) for i in range(extra_pts)]
            full = torch.sort(torch.cat([tt] + inner), descending=bool(tt[0] > tt[-1])).values
            return full[1:] if bad_grid else full
        f = lambda t_, y_: -y_ * w_ * (1 + t_) + torch.sin(y_)  # noqa: E731
        y = D(y0).clone().requires_grad_(True)
        extra = dict(adjoint_params=(w_,)) if api == "odeint_adjoint" else {}
        sol = getattr(lib, api)(f, y, D(t), method=method, options=dict(grid_constructor=grid, interp=interp), **extra)
        out = [("sol", sol.detach().clone())]
        if api == "odeint_adjoint":
            sol[-1].sum().backward()
            out += [("gw", w_.grad), ("gy", y.grad)]
        out.append(("grid_calls", len(seen)))
        out += [(f"grid{i}", s) for i, s in enumerate(seen)]
        return out
    return f"grid_constructor {api} {method} {str(dtype)[6:]} extra={extra_pts} {interp} bad={bad_grid}", program


def blowup_case(rng):
    """Fields that leave the finite range: y' = y^2 (finite-time blow-up), a NaN switched on after some time, an inf
    derivative — what is raised (class and text, rk_common.py:286-290 `non-finite values in state`, `underflow in dt`,
    `max_num_steps exceeded`), after how many evaluations, and what a fixed-grid method returns (inf / nan rows)."""
    g = gen(rng)
    dtype = rng.choice([torch.float32, torch.float64])
    method = rng.choice(ADAPTIVE + FIXED)
    kind = rng.choice(["square", "nan_after", "inf_field", "huge"])
    y0 = (torch.rand(3, generator=g, dtype=torch.float64) + 0.5).to(dtype)
    t = torch.tensor([0.0, 1.0, 3.0], dtype=dtype)
    if rng.random() < 0.3:
        t = -t
    kw = tol(rng)
    opts = {}
    if method in FIXED:
        opts["step_size"] = rng.choice([0.05, 0.25])
    elif rng.random() < 0.4 or kind == "huge":
        opts["max_num_steps"] = rng.choice([20, 200])       # ("huge" without a budget walks 1e-30-sized steps for minutes)
    tup = rng.random() < 0.3
    api = rng.choice(["odeint", "odeint", "odeint_adjoint"])

    def program(lib):
        calls = [0]
        w_ = torch.tensor(1.0, dtype=dtype, device=TARGET, requires_grad=True)
        sgn = -1.0 if float(t[-1]) < 0 else 1.0

        def core(t_, y_):
            calls[0] += 1
            if kind == "square":
                return sgn * y_ * y_ * w_
            if kind == "nan_after":
                return torch.where(sgn * t_ > 0.7, torch.full_like(y_, float("nan")), -y_ * w_)
            if kind == "inf_field":
                return torch.where(sgn * t_ > 0.4, torch.full_like(y_, float("inf")), -y_ * w_)
            return sgn * y_ * 1e30 * w_
        f = (lambda t_, y_: (core(t_, y_[0]), -y_[1])) if tup else core
        y = D(y0).clone().requires_grad_(api == "odeint_adjoint")
        state = (y, torch.ones(2, dtype=dtype, device=TARGET)) if tup else y
        extra = dict(adjoint_params=(w_,)) if api == "odeint_adjoint" else {}
        out = []
        try:
            with torch.no_grad() if api == "odeint" else torch.enable_grad():
                sol = getattr(lib, api)(f, state, D(t), method=method, options=dict(opts) or None, **kw, **extra)
            main = sol[0] if tup else sol
            out.append(("sol", main.detach().clone()))
            if api == "odeint_adjoint":
                main[-1].sum().backward()
                out += [("gw", w_.grad), ("gy", y.grad)]
        except Exception as e:  # noqa: BLE001
            out.append(("raised", f"{type(e).__name__}: {str(e)[:70]}"))
        out.append(("calls", calls[0]))
        return out
    return f"blowup {kind} {api} {method} {str(dtype)[6:]} tuple={tup} rev={float(t[-1]) < 0} {opts} {kw}", program


FAMILIES = [blowup_case, options_case, direct_event_case, grid_constructor_case, stochastic_case, callbacks_case, event_grad_case, explicit_params_case, ragged_tuple_case, odd_dtype_case, dense_case]


if os.environ.get("TDEQ_FUZZ_FAMILY"):         # e.g. TDEQ_FUZZ_FAMILY=odd_dtype,ragged_tuple
    FAMILIES = [f for f in FAMILIES if f.__name__[:-5] in os.environ["TDEQ_FUZZ_FAMILY"].split(",")]


def same(a, b, exact=True):
    if isinstance(a, tuple) and isinstance(b, tuple):
        return len(a) == len(b) and all(same(x, y, exact) for x, y in zip(a, b))
    if torch.is_tensor(a) and torch.is_tensor(b):
        if a.dtype != b.dtype or a.shape != b.shape:
            return False
        if exact:
            return torch.equal(a, b) or (bool((a.isnan() == b.isnan()).all())
                                         and torch.equal(a.nan_to_num(), b.nan_to_num()))
        work = torch.complex128 if a.is_complex() else torch.float64
        rt = 1e-11 if a.dtype in (torch.float64, torch.complex128) else 2e-3
        return bool(((a.to(work) - b.to(work)).abs() <= rt * (float(a.to(work).abs().max()) + 1e-300)).all())
    return a == b


def attempt(lib, program):
    try:
        return program(lib)
    except Exception as e:  # noqa: BLE001
        return [("raised", f"{type(e).__name__}: {str(e)[:200]}")]


def rel_diff(a, b):
    """Largest relative difference between two logged values (tensors / tuples / numbers), None if incomparable."""
    if isinstance(a, tuple) and isinstance(b, tuple):
        if len(a) != len(b):
            return None
        ds = [rel_diff(x, y) for x, y in zip(a, b)]
        return None if any(d is None for d in ds) else max(ds, default=0.0)
    if torch.is_tensor(a) and torch.is_tensor(b):
        b = b.cpu()
        if a.shape != b.shape or a.dtype != b.dtype:
            return None
        work = torch.complex128 if a.is_complex() else torch.float64
        return float((a.to(work) - b.to(work)).abs().max() / (a.to(work).abs().max() + 1e-30)) if a.numel() else 0.0
    if a is None and b is None:
        return 0.0
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return 0.0 if a == b else float("inf")
    return 0.0 if a == b else None


class kernel_backend:
    """TDEQ_FUZZ_DEVICE=oracle (build container, no GPU): the second run keeps its tensors on the CPU but takes the HIP
    path's host logic — padded segment layout, segmented norms, fused packing, carried partial sums — over the C oracle's
    kernels (bit-identical to the HIP kernels, tests/test_kernels_gpu.py), exactly what the `dev="cpu"` tests do."""

    def __enter__(self):
        if DEVICE == "oracle":
            from torchdiffeq_amd import _native
            from oracle.kernels import OracleKernels
            self.prev, ok = _native.get_kernels, OracleKernels()
            _native.get_kernels = lambda d, dtype=None: ok

    def __exit__(self, *exc):
        if DEVICE == "oracle":
            from torchdiffeq_amd import _native
            _native.get_kernels = self.prev


def main_device():
    """Host path (CPU) vs HIP kernels (TDEQ_FUZZ_DEVICE) on the same programs, to tolerance.  The stochastic family
    (device RNG differs) and bf16 states (host path by design) are left out."""
    global TARGET
    from torchdiffeq_amd import _fallback
    bad = ran = counts = 0
    worst = {}
    for case_no in range(n_cases):
        made = rng.choice(FAMILIES)(rng)
        desc, program = made[0], made[1]
        if desc.startswith("stochastic") or "bfloat16" in desc or (only is not None and case_no != only):
            continue
        if DEVICE == "oracle" and "complex" in desc:
            continue            # (the oracle object is the real kernels' twin; complex states have their own, oracle/complex_norms.py)
        wide = "float64" in desc or "complex128" in desc
        TARGET = "cpu"
        la = attempt(tda, program)
        TARGET = "cpu" if DEVICE == "oracle" else DEVICE
        with warnings.catch_warnings(), kernel_backend():
            warnings.simplefilter("error", _fallback.HostPathWarning)       # the device run must be on the kernels
            _fallback._warned = False
            lb = attempt(tda, program)
        TARGET = "cpu"
        ran += 1
        msgs = []
        if len(la) != len(lb):
            counts += 1                     # another number of steps / evaluations: fp32 noise; an fp64 case is reported
            if wide:
                msgs.append(f"log length {len(la)} vs {len(lb)}: {str(la[-1])[:120]} | {str(lb[-1])[:120]}")
        else:
            for (na, va), (nb, vb) in zip(la, lb):
                d = rel_diff(va, vb) if na == nb else None
                kind = ("grad" if na.startswith("g") else "log" if na.startswith("log") else "sol") + ("64" if wide else "32")
                if d is None:
                    msgs.append(f"{na}/{nb}: {str(va)[:100]} | {str(vb)[:100]}")
                    continue
                if d == float("inf") and not wide:
                    counts += 1
                    continue
                worst[kind] = max(worst.get(kind, 0.0), d)
                limit = {"sol64": 1e-6, "grad64": 1e-5, "log64": 1e-6, "sol32": 2e-3, "grad32": 5e-2, "log32": 0.3}[kind]
                if not d <= limit:
                    msgs.append(f"{na}: rel {d:.2e} (limit {limit:.0e})")
        if msgs:
            bad += 1
            print(f"case {case_no}: {desc}")
            for m in msgs[:5]:
                print("    ", m)
        if (case_no + 1) % 25 == 0:
            print(f"... {case_no + 1} cases, {bad} beyond tolerance", flush=True)
    print(f"seed {seed}: {ran} programs host path vs {DEVICE}, {bad} beyond tolerance, {counts} fp32 cases with another step / "
          f"evaluation count, worst relative differences {({k: float(f'{v:.2e}') for k, v in sorted(worst.items())})}")


def main():
    torch.set_num_threads(1)               # (process-wide settings only when run as a tool: tests import this module)
    warnings.simplefilter("ignore")
    if DEVICE:
        return main_device()
    bad = 0
    for case_no in range(n_cases):
        made = rng.choice(FAMILIES)(rng)
        desc, program = made[0], made[1]
        # plain odeint under autograd: the hand-written backward agrees to rounding (DESIGN header (9))
        exact_grads = made[2] if len(made) > 2 else True
        if only is not None and case_no != only:
            continue
        la, lb = attempt(ref, program), attempt(tda, program)
        msgs = []
        if len(la) != len(lb):
            msgs.append(f"log length {len(la)} vs {len(lb)}: {str(la[-1])[:200]} | {str(lb[-1])[:200]}")
        for (na, va), (nb, vb) in zip(la, lb):
            exact = exact_grads or not na.startswith("g")
            if " odeint " in desc and na.startswith("g"):
                exact = False
            if na != nb or not same(va, vb, exact):
                if torch.is_tensor(va) and torch.is_tensor(vb) and va.shape == vb.shape and va.dtype == vb.dtype:
                    work = torch.complex128 if va.is_complex() else torch.float64
                    d = float((va.to(work) - vb.to(work)).abs().max() / (va.to(work).abs().max() + 1e-300))
                    msgs.append(f"{na}: rel {d:.2e} {va.dtype}")
                else:
                    msgs.append(f"{na}/{nb}: {str(va)[:160]} | {str(vb)[:160]}")
        if msgs:
            bad += 1
            print(f"case {case_no}: {desc}")
            for m in msgs[:5]:
                print("    ", m)
        if (case_no + 1) % 25 == 0:
            print(f"... {case_no + 1} cases, {bad} with differences", flush=True)
    print(f"seed {seed}: {n_cases} programs, {bad} with differences")


if __name__ == "__main__":
    main()
