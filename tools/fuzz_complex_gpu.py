"""Differential fuzzing of the COMPLEX path on the GPU box (r04): random complex64 / complex128 solves through the HIP
kernels (`_native.ComplexHipKernels`: real kernels on the (re, im) view + complex norm kernels; look-ahead, host-driven
and captured steps) against the package's torch-op path (`_fallback.HostKernels`) forced onto the same device.  Both
evaluate the user's func with the same ATen kernels, so: fixed-grid solves bit-identical; adaptive solves equal
evaluation counts and 1e-13 (complex128) / 2e-5 (complex64: a last-bit difference of a norm sum may move a step size);
adjoint gradients 1e-9 / 1e-4.

    python tools/fuzz_complex_gpu.py [seed] [cases]"""
import os
import random
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchdiffeq_amd as tda  # noqa: E402
from torchdiffeq_amd import _fallback, _native  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = random.Random(seed)
warnings.simplefilter("ignore")
DEV = os.environ.get("FUZZ_DEV", "cuda")
REAL = os.environ.get("FUZZ_REAL", "0") == "1"
BACKPROP = os.environ.get("FUZZ_BACKPROP", "0") == "1"      # half of the non-adjoint cases differentiate through the solver
# FUZZ_LOW=1 (r05): bf16 / fp16 states — `_lowp.LowPrecisionHipKernels` (csrc/tdeq_kernels_lp.hpp) against the same arithmetic
# in torch ops on the same device (`_fallback.KernelOrderLowHostKernels`).  Elementwise the two are bit-identical; the norm
# twin forms sqrt(mean) from the fp64 sum of the rounded squares like the kernels' host side, so EVERY solve — fixed grid and
# adaptive, host-driven and with the device controller + look-ahead — must agree bit for bit with equal evaluation counts.
LOW = os.environ.get("FUZZ_LOW", "0") == "1"
REAL = REAL or LOW
NL = (lambda y: 0.1 * torch.tanh(y) * y.abs()) if REAL else (lambda y: 0.1j * y * y.abs())
host = _fallback.KernelOrderLowHostKernels() if LOW else _fallback.KernelOrderHostKernels()
orig_get = _native.get_kernels
ADAPTIVE = ["dopri5", "dopri8", "tsit5", "bosh3", "fehlberg2", "adaptive_heun"]
FIXED = ["euler", "midpoint", "heun2", "heun3", "rk4", "explicit_adams", "implicit_adams"]
bad = 0
for case in range(n_cases):
    method = rng.choice(ADAPTIVE + ADAPTIVE + FIXED)
    cdt = rng.choice([torch.complex64, torch.complex128])
    rdt = torch.float32 if cdt == torch.complex64 else torch.float64
    if REAL:
        cdt = rdt               # FUZZ_REAL=1: the same comparison for fp32 / fp64 states (transcendental field included)
    if LOW:
        cdt = rdt = torch.bfloat16 if (method in ADAPTIVE or rng.random() < 0.5) else torch.float16
    shape = rng.choice([(), (1,), (5,), (3, 4), (33, 7), (1025,), (2, 3, 5), (70000,)])
    is_tuple = rng.random() < 0.3
    rev = rng.random() < 0.4
    npts = rng.choice([2, 3, 7, 40])
    adjoint = rng.random() < 0.25 and method in ("dopri5", "rk4", "bosh3", "tsit5")
    g = torch.Generator().manual_seed(rng.randrange(10 ** 6))
    def mk(s):
        z = torch.complex(torch.randn(s, generator=g, dtype=torch.float64), torch.randn(s, generator=g, dtype=torch.float64))
        return (z.real if REAL else z).to(cdt).to(DEV)
    y0, yb = mk(shape), torch.randn(3, generator=g, dtype=torch.float64).to(rdt).to(DEV)       # second component REAL
    w = mk(shape if shape else ()) * 0.3
    t = torch.sort(torch.rand(npts, generator=g, dtype=torch.float64) * 2).values.to(torch.float32 if LOW else rdt).to(DEV)
    if float((t[1:] - t[:-1]).min()) < 1e-3:
        continue
    if rev:
        t = t.flip(0)
    opts = {}
    if method in ADAPTIVE:
        r = rng.random()
        if r < 0.15:
            opts["first_step"] = 0.01
        elif r < 0.3:
            opts["max_step"] = 0.2
        elif r < 0.4 and not rev:
            opts["step_t"] = torch.tensor([float(t.min()) + 0.0137], dtype=rdt)
        if rng.random() < 0.3:
            opts["hip_graph"] = True
    else:
        if rng.random() < 0.5:
            opts["step_size"] = 0.05
        if rng.random() < 0.3:
            opts["interp"] = "cubic"
        if rng.random() < 0.3:
            opts["perturb"] = True
    rtol, atol = (1e-5, 1e-7) if cdt == torch.complex64 else (1e-8, 1e-10)
    if LOW:
        rtol, atol = 2e-2, 2e-3
        if "step_t" in opts:
            opts["step_t"] = opts["step_t"].float()
    lookahead = rng.random() < 0.7
    backprop = BACKPROP and not adjoint and rng.random() < 0.5
    if os.environ.get("FUZZ_ONLY_CASE") and case != int(os.environ["FUZZ_ONLY_CASE"]):
        continue                      # (replay of one case: the random stream above is consumed identically)
    res = []
    for which in ("hip", "host"):
        _native.get_kernels = orig_get if which == "hip" else (lambda device, dtype=None: host)
        os.environ["TDEQ_LOOKAHEAD"] = "1" if (lookahead or which == "host") else "0"
        nfe = [0]
        wp = w.clone().requires_grad_(adjoint or backprop)

        def f(tt, y):
            nfe[0] += 1
            if is_tuple:
                return (-y[0] * wp * (1 + 0.2 * tt) + NL(y[0]), -0.4 * y[1] * (1 + y[0].abs().mean().to(y[1].dtype)))
            return -y * wp * (1 + 0.2 * tt) + NL(y)
        x = y0.clone().requires_grad_(adjoint or backprop)
        try:
            if backprop:
                # backprop THROUGH the solver (autodiff._LinearOp nodes over the kernels), output times in the graph too
                tg = t.clone().requires_grad_(True)
                out = tda.odeint(f, (x, yb) if is_tuple else x, tg, method=method, rtol=rtol, atol=atol,
                                 options={k: v for k, v in opts.items() if k != "hip_graph"})
                y = out[0] if is_tuple else out
                (y[-1].abs().pow(2).sum() + y[len(t) // 2].abs().sum()).backward()
                res.append(("ok", [y.detach(), x.grad, wp.grad, tg.grad], nfe[0]))
            elif adjoint:
                # with `hip_graph`: func as an nn.Module, so that the BACKWARD solve is captured as well (forward and
                # augmented trial steps replayed; the torch twin runs both eagerly)
                class Field(torch.nn.Module):
                    def __init__(self):
                        super().__init__()
                        self.w = torch.nn.Parameter(w.clone())

                    def forward(self, tt, y):
                        if is_tuple:
                            return (-y[0] * self.w * (1 + 0.2 * tt) + NL(y[0]),
                                    -0.4 * y[1] * (1 + y[0].abs().mean().to(y[1].dtype)))
                        return -y * self.w * (1 + 0.2 * tt) + NL(y)
                field = Field()
                o = dict(opts) if which == "hip" else {k: v for k, v in opts.items() if k != "hip_graph"}
                out = tda.odeint_adjoint(field, (x, yb) if is_tuple else x, t, method=method, rtol=rtol, atol=atol, options=o)
                y = out[0] if is_tuple else out
                y[-1].abs().pow(2).sum().backward()
                res.append(("ok", [y.detach(), x.grad, field.w.grad], nfe[0]))
            else:
                with torch.no_grad():
                    out = tda.odeint(f, (x, yb) if is_tuple else x, t, method=method, rtol=rtol, atol=atol, options=dict(opts))
                res.append(("ok", [out[0] if is_tuple else out] + ([out[1]] if is_tuple else []), nfe[0]))
        except Exception as e:
            res.append(("err", type(e).__name__ + ": " + str(e)[:90], 0))
    _native.get_kernels = orig_get
    a, b = res
    desc = (case, method, str(cdt)[6:], shape, is_tuple, 'backprop' if backprop else adjoint, rev, lookahead, {k: (v if not torch.is_tensor(v) else "t") for k, v in opts.items()})
    if a[0] != b[0]:
        bad += 1
        print("STATUS", desc, a[1] if a[0] == "err" else "ok", "|", b[1] if b[0] == "err" else "ok")
        continue
    if a[0] == "err":
        continue
    captured = opts.get("hip_graph")        # replays do not run func's Python body: counts differ by construction
    if a[2] != b[2] and not captured and (rdt == torch.float64 or LOW):
        bad += 1
        print("NFE", desc, a[2], b[2])
        continue
    exact = method in FIXED and method != "implicit_adams"
    tol = (1e-9 if (adjoint or backprop) else 1e-12) if rdt == torch.float64 else (2e-4 if (adjoint or backprop) else 3e-5)
    if LOW:
        tol = 0.5 if (adjoint or backprop) else 0.0
        exact = exact or not (adjoint or backprop)
    if method == "dopri8":
        tol = max(tol, 1e-6)    # a noise-dominated 9-term error estimate turns ONE ulp of a norm sum into 1e-3 of a step
                                # size (docs/LAB_NOTEBOOK.md §8; the complex norm kernel adds re^2 + im^2 in double, the torch twin
                                # squares a rounded modulus) — equal evaluation counts are still required above
    for i, (p, q) in enumerate(zip(a[1], b[1])):
        fin_p, fin_q = torch.isfinite(torch.view_as_real(p) if p.is_complex() else p), \
            torch.isfinite(torch.view_as_real(q) if q.is_complex() else q)
        if not bool(fin_p.all()) or not bool(fin_q.all()):
            if not torch.equal(fin_p, fin_q):
                bad += 1
                print("NONFINITE", desc, i, int((~fin_p).sum()), int((~fin_q).sum()))
                break
            p, q = torch.where(torch.isfinite(p.abs()), p, 0), torch.where(torch.isfinite(q.abs()), q, 0)   # both blew up alike
        d = float((p - q).abs().max() / (q.abs().max() + 1e-30))
        if (exact and not adjoint and not backprop and d != 0.0) or not d <= tol:
            bad += 1
            print("VALUE", desc, i, d)
            break
print("done", n_cases, "bad", bad)
