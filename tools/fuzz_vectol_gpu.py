"""Differential fuzzing of PER-ELEMENT tolerances on the GPU box (r05, ABI 20): random adaptive solves whose `rtol` / `atol`
are tensors broadcasting against the state — the fused route (tdeq_error_norm_vec[_ctrl] continuing the partial error row of
the step's last combine, carried partial sums, device controller + look-ahead) against the r04 route of the same package on
the same device (raw error from the kernel, the scaling and the norm as fp64 torch ops, host-driven steps): equal evaluation
and accept / reject counts, solutions to 1e-12 (fp64) / 2e-5 (fp32: one ulp of a norm sum may move a step size).

    python tools/fuzz_vectol_gpu.py [seed] [cases]"""
import os
import random
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchdiffeq_amd as tda  # noqa: E402
from torchdiffeq_amd import solvers  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = random.Random(seed)
warnings.simplefilter("ignore")
METHODS = ["dopri5", "dopri5", "dopri8", "tsit5", "bosh3", "fehlberg2", "adaptive_heun"]
orig_init = solvers.RKAdaptiveStepsizeODESolver.__init__
made = []


def spy(self, *a, **k):
    orig_init(self, *a, **k)
    made.append(self)


def without(self, *a, **k):
    orig_init(self, *a, **k)
    self._vec_fused, self._vec_ctrl, self._lookahead = None, False, False
    made.append(self)


bad = 0
for case in range(n_cases):
    method = rng.choice(METHODS)
    dtype = rng.choice([torch.float32, torch.float64])
    shape = rng.choice([(7,), (33, 5), (1025,), (300, 12), (70000,), (2, 3, 5)])
    g = torch.Generator().manual_seed(rng.randrange(10 ** 6))
    y0 = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype).cuda()
    w = (torch.rand(shape, generator=g, dtype=torch.float64) + 0.3).to(dtype).cuda()
    npts = rng.choice([2, 3, 6])
    t = torch.sort(torch.rand(npts, generator=g, dtype=torch.float64) * 2).values.cuda()
    if float((t[1:] - t[:-1]).min()) < 1e-3:
        continue
    if rng.random() < 0.3:
        t = t.flip(0)
    lo, hi = (-6, -3) if dtype == torch.float32 else (-9, -5)
    form = rng.choice(["rtol", "atol", "both", "last_dim"])
    mk = lambda s: (10.0 ** (torch.rand(s, generator=g, dtype=torch.float64) * (hi - lo) + lo)).cuda()
    rtol = mk(shape) if form in ("rtol", "both") else (mk(shape[-1:]) if form == "last_dim" else 10.0 ** hi)
    atol = mk(shape) * 1e-2 if form in ("atol", "both") else 10.0 ** (lo - 1)
    opts = {}
    if rng.random() < 0.3:
        opts["first_step"] = 0.3            # forces rejections
    lookahead = rng.random() < 0.7
    nfe = [0]

    def f(tt, y):
        nfe[0] += 1
        return -y * w * (1 + 0.3 * torch.sin(3 * tt)) + 0.1 * torch.tanh(y)
    res = []
    for which in ("fused", "torch_ops"):
        solvers.RKAdaptiveStepsizeODESolver.__init__ = spy if which == "fused" else without
        os.environ["TDEQ_LOOKAHEAD"] = "1" if lookahead else "0"
        nfe[0] = 0
        try:
            with torch.no_grad():
                y = tda.odeint(f, y0, t, method=method, rtol=rtol, atol=atol, options=dict(opts))
            s = made[-1]
            res.append(("ok", y, nfe[0], s.n_accepted, s.n_rejected, s._vec_fused is not None))
        except Exception as e:
            res.append(("err", type(e).__name__ + ": " + str(e)[:80]))
    solvers.RKAdaptiveStepsizeODESolver.__init__ = orig_init
    a, b = res
    desc = (case, method, str(dtype)[6:], shape, form, lookahead, opts)
    if a[0] != b[0]:
        bad += 1
        print("STATUS", desc, a[1] if a[0] == "err" else "ok", "|", b[1] if b[0] == "err" else "ok")
        continue
    if a[0] == "err":
        continue
    if not a[5] or b[5]:
        bad += 1
        print("ROUTE", desc, a[5], b[5])
        continue
    if a[2:5] != b[2:5] and dtype == torch.float64 and method != "dopri8":
        bad += 1
        print("COUNTS", desc, a[2:5], b[2:5])
        continue
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    if method == "dopri8":
        tol = max(tol, 1e-6)      # noise-dominated 13-term estimate: one ulp of a norm sum moves a step size (notebook §8)
    d = float((a[1] - b[1]).abs().max() / (b[1].abs().max() + 1e-30))
    if not d <= tol:
        bad += 1
        print("VALUE", desc, d, a[2:5], b[2:5])
print("done", n_cases, "bad", bad)
