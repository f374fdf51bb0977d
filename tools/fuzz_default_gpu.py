"""r06: differential fuzzing of the BUILT-IN DEFAULT (hip_graph='auto', no option, no environment variable) on the GPU box.

Every case is a small "training loop": the same func object solved REPS times in default mode — first sight eager, second
sight captured, later solves replayed with the per-solve re-check — next to the same solves with `hip_graph=False`.  Whatever
the default decides (capture, refuse, stay eager), the user-visible results must be those of the eager path:
    * solutions bit-identical (`torch.equal`), gradients of `odeint_adjoint` / backprop bit-identical;
    * a field that COUNTS its evaluations sees exactly the eager counts in every repetition;
    * a Python number the field reads — through an attribute, a module-level name, a nested config object, or a
      `__slots__` object no key can see — and that changes between repetitions takes effect at once;
    * no warning is ever emitted (the default is silent).
Random: method (all explicit adaptive + fixed-grid ones), fp32 / fp64, tensor / tuple state, time direction, output grid,
plain odeint under no_grad / odeint_adjoint / backprop through odeint, kind of func.

    python tools/fuzz_default_gpu.py [seed] [cases]"""
import os
import random
import sys
import types
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.pop("TDEQ_HIP_GRAPH", None)
import torchdiffeq_amd as tda  # noqa: E402
from torchdiffeq_amd import _graph  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = random.Random(seed)
dev = torch.device("cuda:0")
ADAPTIVE = ["dopri5", "dopri5", "dopri8", "tsit5", "bosh3", "fehlberg2"]
FIXED = ["rk4", "euler", "midpoint", "heun3"]
REPS = 4
GLOBAL_CFG = types.SimpleNamespace(scale=1.0)
ALPHA = 1.0


class Hidden:
    __slots__ = ("v",)

    def __init__(self):
        self.v = 1.0


def make_func(kind, d, dtype, g, is_tuple):
    lin = torch.nn.Linear(d, d).to(dtype).to(dev)
    with torch.no_grad():
        lin.weight.mul_(0.5)
    hidden = Hidden()
    state = {"set": lambda v: None, "params": list(lin.parameters()), "counter": None}

    def rhs(t, y, scale):
        if is_tuple:
            return torch.tanh(lin(y[0])) * torch.cos(t) * scale, -0.4 * y[1] * scale
        return torch.tanh(lin(y)) * torch.cos(t) * scale

    if kind in ("module", "module_attr_scale", "module_counter"):
        class F(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.lin, self.scale, self.nfe = lin, 1.0, 0

            def forward(self, t, y):
                if kind == "module_counter":
                    self.nfe += 1
                return rhs(t, y, self.scale)
        f = F()
        if kind == "module_attr_scale":
            state["set"] = lambda v: setattr(f, "scale", v)
        if kind == "module_counter":
            state["counter"] = lambda: f.nfe
        return f, state
    if kind == "closure":
        return (lambda t, y: rhs(t, y, 1.0)), state
    if kind == "closure_counter":
        n = [0]

        def f(t, y):
            n[0] += 1
            return rhs(t, y, 1.0)
        state["counter"] = lambda: n[0]
        return f, state
    if kind == "global_scale":
        def f(t, y):
            return rhs(t, y, ALPHA)

        def setter(v):
            global ALPHA
            ALPHA = v
        state["set"] = setter
        return f, state
    if kind == "config_scale":
        def f(t, y):
            return rhs(t, y, GLOBAL_CFG.scale)
        state["set"] = lambda v: setattr(GLOBAL_CFG, "scale", v)
        return f, state
    if kind == "hidden_scale":          # invisible to every key: only the per-solve re-check can notice
        def f(t, y):
            return rhs(t, y, hidden.v)
        state["set"] = lambda v: setattr(hidden, "v", v)
        return f, state
    if kind == "callable_object":
        class Obj:
            def __init__(self):
                self.net = lin

            def __call__(self, t, y):
                return rhs(t, y, 1.0)
        return Obj(), state
    raise ValueError(kind)


KINDS = ["module", "module", "module_attr_scale", "module_counter", "closure", "closure_counter", "global_scale", "config_scale",
         "hidden_scale", "callable_object"]
bad = 0
captured = refused = 0
for case in range(n_cases):
    g = torch.Generator().manual_seed(rng.randrange(10 ** 6))
    kind = rng.choice(KINDS)
    dtype = rng.choice([torch.float32, torch.float64])
    fixed = rng.random() < 0.25
    method = rng.choice(FIXED if fixed else ADAPTIVE)
    is_tuple = rng.random() < 0.25
    api = rng.choice(["nograd", "nograd", "adjoint", "backprop"])
    if kind == "hidden_scale" and api == "backprop":
        api = "nograd"
    d = rng.choice([3, 8])
    n = rng.choice([1, 17, 300])
    y0 = torch.randn(n, d, generator=g, dtype=torch.float64).to(dtype).to(dev)
    yb = torch.randn(5, generator=g, dtype=torch.float64).to(dtype).to(dev)
    n_t = rng.choice([2, 3, 40]) if fixed else rng.choice([2, 3])
    t = torch.linspace(0.0, rng.uniform(0.5, 2.0), n_t, dtype=dtype, device=dev)
    if rng.random() < 0.3:
        t = t.flip(0)
    tol = dict(rtol=1e-5, atol=1e-7) if dtype == torch.float32 else dict(rtol=1e-7, atol=1e-9)
    GLOBAL_CFG.scale = 1.0
    ALPHA = 1.0
    f, st = make_func(kind, d, dtype, g, is_tuple)
    scales = [1.0, 1.0, 1.0, 0.5] if rng.random() < 0.5 else [1.0, 0.5, 0.5, 1.0]
    desc = (case, kind, method, str(dtype)[6:], is_tuple, api, n, d, n_t)

    def solve(opts):
        x = y0.clone().requires_grad_(api != "nograd")
        state0 = (x, yb) if is_tuple else x
        for p in st["params"]:
            p.grad = None
        c0 = st["counter"]() if st["counter"] else 0
        if api == "nograd":
            with torch.no_grad():
                out = tda.odeint(f, state0, t, method=method, options=opts, **({} if fixed else tol))
        elif api == "adjoint":
            out = tda.odeint_adjoint(f, state0, t, method=method, options=opts, adjoint_params=tuple(st["params"]),
                                     **({} if fixed else tol))
        else:
            out = tda.odeint(f, state0, t, method=method, options=opts, **({} if fixed else tol))
        o = out[0] if is_tuple else out
        grads = []
        if api != "nograd":
            o[-1].pow(2).sum().backward()
            grads = [x.grad.clone()] + [p.grad.clone() if p.grad is not None else torch.zeros(()) for p in st["params"]]
        cnt = (st["counter"]() - c0) if st["counter"] else None
        return o.detach().clone(), grads, cnt
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for rep in range(REPS):
                st["set"](scales[rep])
                got = solve(None)
                want = solve(dict(hip_graph=False))
                ok = torch.equal(got[0], want[0]) and len(got[1]) == len(want[1]) and \
                    all(torch.equal(a, b) for a, b in zip(got[1], want[1])) and got[2] == want[2]
                if not ok:
                    bad += 1
                    dy = float((got[0] - want[0]).abs().max())
                    print("MISMATCH", desc, "rep", rep, "max|dy|", dy, "counts", got[2], want[2])
                    break
        msgs = [str(x.message)[:80] for x in w if "hip_graph" in str(x.message)]
        if msgs:
            bad += 1
            print("WARNED", desc, msgs[:2])
    except Exception as exc:      # noqa: BLE001
        bad += 1
        print("ERROR", desc, type(exc).__name__, str(exc)[:160])
    try:
        captured += int(f in _graph._GraphStep._cache and bool(_graph._GraphStep._cache[f]))
        refused += int(f in _graph._GraphStep._refused)
    except TypeError:
        pass
print("done", n_cases, "bad", bad, "| funcs with a cached captured step:", captured, "| refused:", refused, "| seed", seed)
