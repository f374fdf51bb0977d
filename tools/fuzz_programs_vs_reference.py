"""Program-level differential fuzzing against the reference (build container only: imports /root/reference).

  PYTHONDONTWRITEBYTECODE=1 python tools/fuzz_programs_vs_reference.py [seed] [cases]

Where tools/fuzz_vs_reference.py draws one solve per case, a case here is a small USER PROGRAM on CPU tensors (the
package's host path): an nn.Module vector field (time-concatenated MLP, concat-squash layer, a CNF-style field that
differentiates inside `forward`, a plain linear field), a few steps of a training loop (forward solve -> loss ->
backward -> SGD update) through `odeint` or `odeint_adjoint`, optionally an event solve at the end.  Both libraries get
deep copies of the same module and the same data; compared with `torch.equal` after every iteration: the solution, the
loss, every parameter gradient, the gradient wrt y0 (and wrt t when it takes part), the parameters after the update and
the number of field evaluations.  Test infrastructure, like oracle/."""
import copy
import random
import sys
import warnings

import torch
from torch import nn

import os  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torchdiffeq_amd as tda  # noqa: E402

# TDEQ_FUZZ_DEVICE=cuda (GPU box, no reference there): the same programs, the package's CPU host path — which IS the
# reference's arithmetic bit for bit (the default mode of this tool establishes that) — against the HIP kernels on the
# device.  Compared to tolerance: the kernels sum tableau rows left to right and the device's GEMM / tanh round differently.
DEVICE = os.environ.get("TDEQ_FUZZ_DEVICE")
if DEVICE:
    ref = tda
else:
    sys.path.insert(0, "/root/reference")
    import torchdiffeq as ref  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
rng = random.Random(seed)

ADAPTIVE = ["dopri5", "dopri8", "bosh3", "tsit5", "fehlberg2", "adaptive_heun"]
FIXED = ["euler", "midpoint", "heun2", "heun3", "rk4"]


class Counted(nn.Module):
    def __init__(self):
        super().__init__()
        self.nfe = 0


class TimeMLP(Counted):
    def __init__(self, dim, hidden, act, g):
        super().__init__()
        self.l1 = nn.Linear(dim + 1, hidden)
        self.l2 = nn.Linear(hidden, dim)
        self.act = act
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.4

    def forward(self, t, y):
        self.nfe += 1
        tt = torch.ones_like(y[..., :1]) * t
        return self.l2(self.act(self.l1(torch.cat([tt, y], -1))))


class ConcatSquash(Counted):
    def __init__(self, dim, hidden, act, g):
        super().__init__()
        self.lin = nn.Linear(dim, dim)
        self.gate = nn.Linear(1, dim)
        self.bias = nn.Linear(1, dim, bias=False)
        self.frozen = nn.Parameter(torch.randn(dim, generator=g) * 0.1, requires_grad=False)
        self.act = act
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.5

    def forward(self, t, y):
        self.nfe += 1
        tv = t.reshape(1, 1).to(y.dtype)
        return self.act(self.lin(y) * torch.sigmoid(self.gate(tv)) + self.bias(tv)) + self.frozen


class LinearField(Counted):
    def __init__(self, dim, hidden, act, g):
        super().__init__()
        self.A = nn.Parameter(torch.randn(dim, dim, generator=g) * 0.5 - torch.eye(dim))

    def forward(self, t, y):
        self.nfe += 1
        return y @ self.A.t()


class CNFField(Counted):
    """examples/cnf.py in miniature: state (z, logp); exact trace through autograd inside forward."""

    def __init__(self, dim, hidden, act, g):
        super().__init__()
        self.l1 = nn.Linear(dim, hidden)
        self.l2 = nn.Linear(hidden, dim)
        self.act = act
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.4

    def forward(self, t, state):
        self.nfe += 1
        z = state[0]
        with torch.set_grad_enabled(True):
            z = z.requires_grad_(True) if not z.requires_grad else z
            dz = self.l2(self.act(self.l1(z))) * torch.cos(t)
            tr = 0.0
            for i in range(z.shape[-1]):
                tr = tr + torch.autograd.grad(dz[:, i].sum(), z, create_graph=True)[0][:, i]
        return dz, -tr.reshape(-1, 1)


def make_case(rng):
    g = torch.Generator().manual_seed(rng.randrange(10 ** 6))
    dtype = rng.choice([torch.float32, torch.float64])
    kind = rng.choice([TimeMLP, ConcatSquash, LinearField, CNFField, TimeMLP])
    dim = rng.choice([1, 2, 3, 5])
    batch = rng.choice([1, 2, 4, 7])
    act = rng.choice([torch.tanh, nn.functional.softplus, torch.sin, nn.functional.elu])
    field = kind(dim, rng.choice([4, 8, 16]), act, g).to(dtype)
    api = rng.choice(["odeint", "adjoint", "adjoint"])
    method = rng.choice(ADAPTIVE + ADAPTIVE + FIXED)
    npts = rng.choice([2, 3, 6])
    t = torch.sort(torch.rand(npts, generator=g, dtype=torch.float64) * rng.choice([0.5, 1.0, 2.0])).values
    if float((t[1:] - t[:-1]).min()) < 5e-3:
        t = torch.linspace(0, 1, npts, dtype=torch.float64)
    tdtype = dtype if rng.random() < 0.7 else torch.float64
    t = t.to(tdtype)
    if rng.random() < 0.35:
        t = t.flip(0)
    kw = {}
    if rng.random() < 0.7:
        kw["rtol"] = rng.choice([1e-3, 1e-5, 1e-7])
        kw["atol"] = rng.choice([1e-4, 1e-6, 1e-9])
    opts = {}
    if method in FIXED:
        r = rng.random()
        if r < 0.5:
            opts["step_size"] = rng.choice([0.05, 0.11, 0.3])
        if rng.random() < 0.3:
            opts["interp"] = "cubic"
        if rng.random() < 0.2:
            opts["perturb"] = True
    else:
        if rng.random() < 0.25:
            opts["first_step"] = rng.choice([0.01, 0.1])
        if rng.random() < 0.2:
            opts["max_num_steps"] = 1000
        if rng.random() < 0.2:
            opts["safety"] = 0.8
        if rng.random() < 0.15:
            opts["dtype"] = rng.choice([torch.float32, torch.float64])
        if rng.random() < 0.15 and float(t[0]) < float(t[-1]):
            mid = float(t[0] + (t[-1] - t[0]) * 0.37)
            opts[rng.choice(["step_t", "jump_t"])] = torch.tensor([mid], dtype=tdtype)
    if opts:
        kw["options"] = opts
    if api == "adjoint":
        if rng.random() < 0.3:
            kw["adjoint_options"] = dict(norm="seminorm")
            if method in FIXED:
                kw["adjoint_options"].update({k: v for k, v in opts.items()})
        if rng.random() < 0.2:
            kw["adjoint_rtol"] = 1e-4
            kw["adjoint_atol"] = 1e-6
        if rng.random() < 0.15 and "options" not in kw:
            kw["adjoint_method"] = rng.choice(["dopri5", "rk4", "bosh3"])
            if kw["adjoint_method"] == "rk4":
                kw["adjoint_options"] = dict(step_size=0.1)
    z0 = torch.randn(batch, dim, generator=g, dtype=torch.float64).to(dtype)
    if kind is CNFField:
        y0 = (z0, torch.zeros(batch, 1, dtype=dtype))
    else:
        y0 = z0
    w = torch.randn(npts, batch, dim, generator=g, dtype=torch.float64).to(dtype)
    r = rng.random()
    if r < 0.12:
        w[1:] = 0               # the loss sees the initial row only: every backward interval starts from a zero adjoint
    elif r < 0.24:
        w[-1] = 0               # ... the last output not at all (zero cotangent at the start of the backward solve)
    elif r < 0.36:
        w[:-1] = 0              # ... only the last
    elif r < 0.45:
        w[rng.randrange(npts)] = 0
    return dict(field=field, api=api, method=method, t=t, kw=kw, y0=y0, w=w, kind=kind,
                grad_t=rng.random() < 0.3, lr=rng.choice([0.05, 0.3]),
                # (backprop gradients agree to rounding only, so a second iteration would compare different programs)
                iters=rng.choice([1, 2, 3]) if api == "adjoint" else 1,
                # (the event solve comes after the SGD update: only where the update itself is bit-identical)
                event=rng.random() < 0.3 and kind is not CNFField and api == "adjoint")


def to_device(case, device):
    moved = dict(case)
    moved["field"] = copy.deepcopy(case["field"]).to(device)
    moved["y0"] = tuple(c.to(device) for c in case["y0"]) if isinstance(case["y0"], tuple) else case["y0"].to(device)
    moved["t"], moved["w"] = case["t"].to(device), case["w"].to(device)
    kw = dict(case["kw"])
    if "options" in kw:
        kw["options"] = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in kw["options"].items()}
    moved["kw"] = kw
    return moved


def run(lib, case):
    field = copy.deepcopy(case["field"])
    field.nfe = 0
    y0 = case["y0"]
    istuple = isinstance(y0, tuple)
    # (a CNF-style field under odeint_adjoint with y0 in the graph: the reference hands func a no-grad VIEW of a
    #  requires-grad tensor, and torch.autograd.grad wrt that view raises inside forward — a PyTorch quirk, docs/LAB_NOTEBOOK.md §8)
    y0_grad = not (istuple and case["api"] == "adjoint")
    leaves = [c.clone().requires_grad_(y0_grad) for c in (y0 if istuple else (y0,))]
    t = case["t"].clone().requires_grad_(case["grad_t"])
    solve = lib.odeint if case["api"] == "odeint" else lib.odeint_adjoint
    opt = torch.optim.SGD([p for p in field.parameters() if p.requires_grad], lr=case["lr"])
    log = []
    try:
        for _ in range(case["iters"]):
            opt.zero_grad()
            for leaf in leaves + [t]:
                leaf.grad = None
            sol = solve(field, tuple(leaves) if istuple else leaves[0], t, method=case["method"], **case["kw"])
            main = sol[0] if istuple else sol
            loss = (main * case["w"]).sum()
            if istuple:
                loss = loss + sol[1][-1].mean()
            loss.backward()
            log.append(("sol", main.detach().clone()))
            log.append(("loss", loss.detach().clone()))
            for name, p in field.named_parameters():
                log.append(("g:" + name, None if p.grad is None else p.grad.clone()))
            for i, leaf in enumerate(leaves):
                log.append((f"gy{i}", None if leaf.grad is None else leaf.grad.clone()))
            if case["grad_t"]:
                log.append(("gt", None if t.grad is None else t.grad.clone()))
            opt.step()
            log.append(("nfe", field.nfe))
        if case["event"]:
            lim = float(main.detach()[-1].flatten()[0])
            ev = lambda tt, yy: yy.flatten()[0] - (lim + 0.05)  # noqa: E731
            t0 = case["t"][0].detach()
            et, es = lib.odeint_event(field, leaves[0].detach(), t0, event_fn=ev, method=case["method"],
                                      reverse_time=bool(case["t"][0] > case["t"][-1]),
                                      odeint_interface=solve,
                                      **{k: v for k, v in case["kw"].items() if k in ("rtol", "atol")},
                                      options=dict(max_num_steps=300) if case["method"] in ADAPTIVE else
                                      dict(step_size=0.05))
            log.append(("event_t", et.detach().clone()))
            log.append(("event_y", es.detach().clone()))
    except Exception as e:  # noqa: BLE001
        log.append(("raised", f"{type(e).__name__}: {str(e)[:160]}"))
    return log


def same(a, b, exact=True, state_dtype=None):
    if torch.is_tensor(a) and torch.is_tensor(b) and not exact and a.shape == b.shape and a.dtype == b.dtype:
        # backprop through plain odeint: the package's hand-written backward sums cotangents in its own order
        # (a time gradient is formed from the STATE's arithmetic even when t itself is fp64)
        tol = 1e-11 if (state_dtype or a.dtype) == torch.float64 else 1e-3
        scale = float(a.double().abs().max()) + 1e-300
        return bool(((a.double() - b.double()).abs().nan_to_num() <= tol * scale).all())
    if torch.is_tensor(a) and torch.is_tensor(b):
        return a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b) or \
            (a.shape == b.shape and a.dtype == b.dtype and bool((a.isnan() == b.isnan()).all())
             and torch.equal(a.nan_to_num(), b.nan_to_num()))
    return a == b


def main():
    torch.set_num_threads(1)               # (process-wide settings only when run as a tool: tests import this module)
    warnings.simplefilter("ignore")
    if DEVICE:
        return main_device()
    bad = 0
    for case_no in range(n_cases):
        case = make_case(rng)
        if only is not None and case_no != only:
            continue
        la, lb = run(ref, case), run(tda, case)
        msgs = []
        if len(la) != len(lb):
            msgs.append(f"log length {len(la)} vs {len(lb)}: {la[-1]} | {lb[-1]}")
        for (na, va), (nb, vb) in zip(la, lb):
            exact = case["api"] == "adjoint" or na in ("sol", "loss", "nfe")
            if na != nb or not same(va, vb, exact, case["w"].dtype):
                if torch.is_tensor(va) and torch.is_tensor(vb) and va.shape == vb.shape:
                    d = float((va.double() - vb.double()).abs().max() / (va.double().abs().max() + 1e-300))
                    msgs.append(f"{na}: rel {d:.2e} dtype {va.dtype}/{vb.dtype}")
                else:
                    msgs.append(f"{na}/{nb}: {str(va)[:150]} | {str(vb)[:150]}")
        if msgs:
            bad += 1
            print(f"case {case_no}: {case['kind'].__name__} {case['api']} {case['method']} "
                  f"{str(case['t'].dtype)[6:]}/{str(case['w'].dtype)[6:]} rev={bool(case['t'][0] > case['t'][-1])} "
                  f"kw={ {k: v for k, v in case['kw'].items()} } grad_t={case['grad_t']} event={case['event']}")
            for m in msgs[:6]:
                print("    ", m)
        if (case_no + 1) % 25 == 0:
            print(f"... {case_no + 1} cases, {bad} with differences", flush=True)
    print(f"seed {seed}: {n_cases} programs, {bad} with differences")


class kernel_backend:
    """TDEQ_FUZZ_DEVICE=oracle (build container, no GPU): the second run keeps its tensors on the CPU but takes the HIP
    path's host logic over the C oracle's kernels (bit-identical to the HIP kernels), as the `dev="cpu"` tests do."""

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
    """Host path (CPU) vs HIP kernels (TDEQ_FUZZ_DEVICE), one iteration per program, to tolerance."""
    from torchdiffeq_amd import _fallback
    worst, nfe_diff, bad, ran = {}, 0, 0, 0
    for case_no in range(n_cases):
        case = make_case(rng)
        case["iters"] = 1
        if only is not None and case_no != only:
            continue
        f64 = case["w"].dtype == torch.float64
        la = run(tda, case)
        with warnings.catch_warnings(), kernel_backend():
            warnings.simplefilter("error", _fallback.HostPathWarning)       # the device run must be on the kernels
            _fallback._warned = False
            lb = run(tda, to_device(case, "cpu" if DEVICE == "oracle" else DEVICE))
        ran += 1
        msgs = []
        if len(la) != len(lb):
            msgs.append(f"log length {len(la)} vs {len(lb)}: {str(la[-1])[:150]} | {str(lb[-1])[:150]}")
        for (na, va), (nb, vb) in zip(la, lb):
            if na == "nfe":
                nfe_diff += va != vb
                if f64 and va != vb:
                    msgs.append(f"nfe {va} vs {vb}")
                continue
            if na == "raised" or nb == "raised":
                if va != vb:
                    msgs.append(f"{na}/{nb}: {str(va)[:120]} | {str(vb)[:120]}")
                continue
            if va is None or vb is None:
                if (va is None) != (vb is None):
                    msgs.append(f"{na}: None on one side")
                continue
            vb = vb.cpu()
            d = float((va.double() - vb.double()).abs().max() / (va.double().abs().max() + 1e-30))
            kind = ("sol" if na in ("sol", "loss", "event_t", "event_y") else "gt" if na == "gt" else "grad") + \
                ("64" if f64 else "32")
            worst[kind] = max(worst.get(kind, 0.0), d)
            limit = {"sol64": 1e-6, "grad64": 1e-5, "gt64": 1e-5, "sol32": 2e-3, "grad32": 5e-2, "gt32": 1.0}[kind]
            if not d <= limit:
                msgs.append(f"{na}: rel {d:.2e} (limit {limit:.0e})")
        if msgs:
            bad += 1
            print(f"case {case_no}: {case['kind'].__name__} {case['api']} {case['method']} "
                  f"{str(case['t'].dtype)[6:]}/{str(case['w'].dtype)[6:]} kw={case['kw']} grad_t={case['grad_t']} "
                  f"event={case['event']}")
            for m in msgs[:6]:
                print("    ", m)
        if (case_no + 1) % 25 == 0:
            print(f"... {case_no + 1} cases, {bad} beyond tolerance", flush=True)
    print(f"seed {seed}: {ran} programs host path vs {DEVICE}, {bad} beyond tolerance, {nfe_diff} with another "
          f"evaluation count (fp32 noise), worst relative differences {({k: float(f'{v:.2e}') for k, v in sorted(worst.items())})}")


if __name__ == "__main__":
    main()
