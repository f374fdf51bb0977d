"""Differential fuzzing ON THE GPU BOX: random solves through the HIP path (eager, look-ahead off, hipGraph) against
the same host logic over the CPU oracle kernels (the `dev="cpu"` path of the tests).  Elementwise polynomial fields
only (no GEMM, no transcendentals: the CPU and the GPU then differ by FMA contraction at most), fp64 — so forward
solves must agree to ~1e-11 and evaluation counts exactly; adjoint gradients to 1e-8.

    python tools/fuzz_gpu_vs_oracle.py [seed] [cases]"""
import os
import random
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchdiffeq_amd as tda  # noqa: E402
from torchdiffeq_amd import _native  # noqa: E402
from oracle.kernels import OracleKernels  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = random.Random(seed)
hip = _native.get_kernels(torch.device("cuda:0"))
oracle = OracleKernels()
_orig_get = _native.get_kernels


def backend(device, dtype=None):
    return hip if torch.device(device).type == "cuda" else oracle


_native.get_kernels = backend
ADAPTIVE = ["dopri5", "dopri8", "tsit5", "bosh3", "fehlberg2", "adaptive_heun"]
FIXED = ["euler", "midpoint", "heun2", "heun3", "rk4"]
bad = 0
def random_tableau_method(case):
    """A random explicit embedded RK table (2..9 stages, structural zeros, FSAL or not) registered as a method of the
    native adaptive solver: the kernels, the end-of-step fusion and the dense output are generic in the table."""
    from torchdiffeq_amd.solvers import RKAdaptiveStepsizeODESolver
    from torchdiffeq_amd.tableaus import Tableau
    S = rng.randint(2, 9)
    rnd = lambda: 0.0 if rng.random() < 0.25 else rng.uniform(-0.6, 0.9)
    beta = [[rnd() for _ in range(i + 1)] for i in range(S)]
    for r in beta:
        if all(v == 0.0 for v in r):
            r[0] = 0.3
    alpha = [min(1.0, abs(sum(r))) for r in beta]
    if rng.random() < 0.5:
        alpha[-1] = 1.0
        tot = sum(beta[-1])
        beta[-1] = [v / tot for v in beta[-1]] if abs(tot) > 1e-3 else [1.0 / S] * S
        c_sol = list(beta[-1]) + [0.0]
    else:
        w = [abs(rnd()) + 0.05 for _ in range(S + 1)]
        c_sol = [v / sum(w) for v in w]
    c_err = [rng.uniform(-1, 1) * 1e-2 * (rng.random() < 0.8) for _ in range(S + 1)]
    if rng.random() < 0.3:
        c_err = [c * 1e-2 for c in c_sol[:-1]] + [rng.uniform(-1, 1) * 1e-3]
    c_err[0] -= sum(c_err)               # like a real pair: the two solutions agree to first order
    mid = [0.5 * c for c in c_sol]
    mid[0] += 0.125
    mid[-1] -= 0.125
    name = f"randtab{case}"
    tda.SOLVERS[name] = type("RandTab", (RKAdaptiveStepsizeODESolver,), dict(order=rng.choice([2, 3, 5, 8]), tableau=Tableau(
        name, 5, tuple(alpha), tuple(tuple(r) for r in beta), tuple(c_sol), tuple(c_err), tuple(mid))))
    return name


for case in range(n_cases):
    method = rng.choice(ADAPTIVE + FIXED)
    if rng.random() < float(os.environ.get("FUZZ_RANDTAB", "0.15")):
        method = random_tableau_method(case)
        ADAPTIVE.append(method)
    shape = rng.choice([(), (1,), (5,), (3, 4), (33, 7), (1025,), (2, 3, 5)])
    is_tuple = rng.random() < 0.3
    rev = rng.random() < 0.4
    npts = rng.choice([2, 3, 7, 40])
    adjoint = rng.random() < float(os.environ.get('FUZZ_ADJOINT', '0.25')) and method in ("dopri5", "rk4", "bosh3", "tsit5")
    g = torch.Generator().manual_seed(rng.randrange(10 ** 6))
    y0 = torch.randn(shape, generator=g, dtype=torch.float64)
    y0b = torch.randn(4, generator=g, dtype=torch.float64)
    t = torch.sort(torch.rand(npts, generator=g, dtype=torch.float64) * 2).values
    if float((t[1:] - t[:-1]).min()) < 1e-3:
        continue
    if rev:
        t = t.flip(0)
    lam = rng.choice([0.3, 1.0, 2.5])
    opts = {}
    if method in ADAPTIVE:
        if rng.random() < 0.2:
            opts["first_step"] = 0.05
        if rng.random() < 0.2:
            opts["max_step"] = 0.3
        if rng.random() < 0.15:
            lo, hi = float(t.min()), float(t.max())
            opts[rng.choice(["step_t", "jump_t"])] = torch.tensor([lo + (hi - lo) * 0.41], dtype=torch.float64)
    else:
        if rng.random() < 0.3:
            opts["step_size"] = rng.choice([0.05, 0.11])
        if rng.random() < 0.2:
            opts["perturb"] = True
        if rng.random() < 0.2:
            opts["interp"] = "cubic"
    graph = rng.choice([None, True, "auto"]) if not (opts.get("step_t") is not None or opts.get("jump_t") is not None) else None
    kw = dict(rtol=rng.choice([1e-5, 1e-8]), atol=rng.choice([1e-7, 1e-10]))
    if method.startswith("randtab"):
        kw = dict(rtol=1e-4, atol=1e-5)          # an inconsistent random method converges nowhere: bound its step count
        opts["max_num_steps"] = 2000

    class Field(torch.nn.Module):
        def __init__(self, dev):
            super().__init__()
            self.a = torch.nn.Parameter(torch.tensor(lam, dtype=torch.float64, device=dev))
            self.n = 0

        def forward(self, tt, y):
            self.n += 1
            if is_tuple:
                ya, yb = y
                # (a sum written out: `yb.sum()` adds in a different order on the two devices, and ONE ulp in f moves a
                # step size by 1e-3 wherever an error estimate is rounding noise — dopri8 after a small first step,
                # seed 777 case 86: err/tol 1.5e-8 — docs/LAB_NOTEBOOK.md §12)
                coupling = ((yb[0] + yb[1]) + yb[2]) + yb[3]
                return (ya * (tt * 0.5 - 1.0) * self.a - ya * ya * ya * 0.1 + coupling * 0.01, yb * (-0.5) * (1.0 + tt))
            return y * (tt * 0.5 - 1.0) * self.a - y * y * y * 0.1

    if case % 10 == 0:
        print("case", case, flush=True)        # progress: a timeout then shows where (blow-up cases take minutes)
    results = []
    for dev, hg, look in (("cpu", None, "1"), ("cuda", None, "1"), ("cuda", None, "0"), ("cuda", graph, "1")):
        if dev == "cuda" and hg is None and look == "1" and False:
            continue
        os.environ["TDEQ_LOOKAHEAD"] = look
        f = Field(dev)
        o = dict(opts)
        for k in ("step_t", "jump_t"):
            if k in o:
                o[k] = o[k].to(dev)
        if hg is not None:
            o["hip_graph"] = hg
        x = y0.detach().clone().to(dev).requires_grad_(adjoint)
        state = (x, y0b.to(dev)) if is_tuple else x
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if adjoint:
                    out = tda.odeint_adjoint(f, state, t.to(dev), method=method, options=o or None, **kw)
                    (out[0] if is_tuple else out)[-1].pow(2).sum().backward()
                    res = [v.detach().cpu() for v in (out if is_tuple else (out,))] + [x.grad.cpu(), f.a.grad.cpu()]
                else:
                    with torch.no_grad():
                        out = tda.odeint(f, state, t.to(dev), method=method, options=o or None, **kw)
                    res = [v.cpu() for v in (out if is_tuple else (out,))]
            results.append(("ok", res, f.n if hg is None else None))
        except Exception as e:
            results.append(("err", type(e).__name__ + ": " + str(e)[:100], None))
    os.environ.pop("TDEQ_LOOKAHEAD", None)
    desc = (case, method, shape, is_tuple, rev, npts, adjoint, {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in opts.items()}, graph, kw)
    base = results[0]
    for name, r in zip(("gpu", "gpu-nolook", "gpu-graph"), results[1:]):
        if r[0] != base[0]:
            bad += 1
            print("STATUS", name, desc, base[1] if base[0] == "err" else "ok", r[1] if r[0] == "err" else "ok")
            continue
        if r[0] == "err":
            if r[1].split(":")[0] != base[1].split(":")[0]:
                bad += 1
                print("ERRTYPE", name, desc, base[1], r[1])
            continue
        tol = 1e-7 if adjoint else 1e-10
        for a, b in zip(base[1], r[1]):
            fin = torch.isfinite(a) & torch.isfinite(b)
            same_nonfinite = bool(((torch.isfinite(a) == torch.isfinite(b)).all()))
            d = float(((a - b)[fin]).abs().max() / (a[fin].abs().max() + 1e-30)) if fin.any() else 0.0
            if a.shape != b.shape or not same_nonfinite or not d <= tol:
                bad += 1
                print("VALUE", name, desc, d)
                break
        else:
            if r[2] is not None and base[2] is not None and r[2] != base[2]:
                print("nfe differs", name, desc, base[2], r[2])
print("done", n_cases, "bad", bad)
