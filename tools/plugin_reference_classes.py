"""The reference's OWN solver classes registered as plug-ins in this package's `SOLVERS` table (build container only:
imports /root/reference) — the plugin protocol of SURVEY.md §8(b) exercised with the classes it was written for.

    PYTHONDONTWRITEBYTECODE=1 python tools/plugin_reference_classes.py [method ...]

`tda.odeint(..., method='ref_<name>')` runs torchdiffeq's class inside torchdiffeq_amd's odeint (input checks, tuple
flattening, time reversal, func wrapper, output layout) and must reproduce `torchdiffeq.odeint(..., method='<name>')`
bit for bit: tensor and tuple states, both time directions, per-component tolerances; then odeint_adjoint with the
reference's class in the forward AND the backward solve."""
import json
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference")
import torch  # noqa: E402
import torchdiffeq as ref  # noqa: E402
import torchdiffeq_amd as tda  # noqa: E402
from torchdiffeq._impl.odeint import SOLVERS as REF_SOLVERS  # noqa: E402

warnings.simplefilter("ignore")
torch.set_num_threads(1)
for name, cls in REF_SOLVERS.items():
    tda.SOLVERS["ref_" + name] = cls
A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)
f = lambda t, y: torch.tanh(y @ A) * torch.cos(t)
y0 = torch.tensor([[2.0, 0.0], [1.0, 0.5]], dtype=torch.float64)
t = torch.linspace(0, 2, 7, dtype=torch.float64)
CASES = (("tensor, increasing t", y0, t, {}), ("tensor, decreasing t", y0, t.flip(0), {}),
         ("tuple, increasing t", (y0, y0[0] * 2), t, {}), ("tuple, decreasing t", (y0, y0[0] * 2), t.flip(0), {}),
         ("tuple, per-component tolerances", (y0, y0[0] * 2), t, dict(rtol=(1e-5, 1e-8), atol=(1e-7, 1e-10))))
names = [n for n in REF_SOLVERS if n in sys.argv[1:]] if len(sys.argv) > 1 else [n for n in REF_SOLVERS if n != "scipy_solver"]
report, bad = {}, 0
for name in names:
    report[name] = {}
    for label, state, tt, kw in CASES:
        fn = f if not isinstance(state, tuple) else (lambda t_, s: (f(t_, s[0]), -s[1] * 0.5))
        t0 = time.time()
        try:
            a = ref.odeint(fn, state, tt, method=name, **kw)
            b = tda.odeint(fn, state, tt, method="ref_" + name, **kw)
            flat = lambda v: v if not isinstance(v, tuple) else torch.cat([x.reshape(len(tt), -1) for x in v], 1)
            same = bool(torch.equal(flat(a), flat(b)))
            report[name][label] = "bit-identical" if same else "max abs diff %g" % float((flat(a) - flat(b)).abs().max())
            bad += not same
        except Exception as e:        # noqa: BLE001
            report[name][label] = "ERROR " + type(e).__name__ + ": " + str(e)[:120]
            bad += 1
    print(name, json.dumps(report[name]), flush=True)

lin = torch.nn.Linear(2, 2).double()


class F(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.l = lin

    def forward(self, t_, st):
        return torch.tanh(self.l(st[0])) * torch.cos(t_), -st[1] * 0.3


report["odeint_adjoint"] = {}
for m in ("dopri5", "rk4", "bosh3", "dopri8"):
    if len(sys.argv) > 1 and m not in sys.argv[1:]:
        continue
    res = []
    for lib, mm in ((ref, m), (tda, "ref_" + m)):
        for p in lin.parameters():
            p.grad = None
        x = y0.clone().requires_grad_(True)
        out = lib.odeint_adjoint(F(), (x, y0[0] * 2), t, method=mm, options={"step_size": 0.05} if m == "rk4" else None)
        (out[0][-1].pow(2).sum() + out[1][-1].sum()).backward()
        res.append([out[0].detach(), x.grad.clone()] + [p.grad.clone() for p in lin.parameters()])
    same = all(torch.equal(a, b) for a, b in zip(*res))
    report["odeint_adjoint"][m] = "solution and gradients bit-identical" if same else \
        "max abs diff %g" % max(float((a - b).abs().max()) for a, b in zip(*res))
    bad += not same
print("odeint_adjoint", json.dumps(report["odeint_adjoint"]))
report["mismatches"] = bad
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
json.dump(report, open(os.path.join(out_dir, "plugin_reference_classes.json"), "w"), indent=1)
print("mismatches", bad)
sys.exit(1 if bad else 0)
