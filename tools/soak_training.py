"""Soak of TRAINING loops on the MI355X: device memory after iteration 50 vs after the last iteration (gc collected) for
backprop through `odeint`, `odeint_adjoint` (eager and with captured steps), `odeint_event` and a tuple-state CNF-style
solve — a leak (reference cycle holding state tensors, a cache keyed by something that changes per call) shows up as
growth.  Writes gpurun_out/soak_training.json."""
import gc
import json
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchdiffeq_amd as tda  # noqa: E402

warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
N_ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def mem():
    gc.collect()
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated()


def loop(name, make_loss):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 16)).to(dev)

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, t, y):
            if isinstance(y, tuple):
                return self.net(y[0]) * torch.cos(t), -y[1]
            return self.net(y) * torch.cos(t)
    f = F()
    opt = torch.optim.SGD(f.parameters(), lr=1e-3)
    marks = {}
    for it in range(N_ITERS):
        opt.zero_grad()
        x = torch.randn(512, 16, device=dev)
        loss = make_loss(f, x)
        loss.backward()
        opt.step()
        if it in (49, N_ITERS - 1):
            marks[it + 1] = mem()
    a, b = marks[50], marks[N_ITERS]
    out = {"bytes_after_50": a, f"bytes_after_{N_ITERS}": b, "growth_bytes": b - a, "finite": bool(torch.isfinite(loss))}
    print(name, json.dumps(out), flush=True)
    return out


def _nograd(fn):
    with torch.no_grad():
        return fn()


t = torch.tensor([0.0, 0.5, 1.0], device=dev)
res = {
    "odeint_backprop": loop("odeint_backprop", lambda f, x: tda.odeint(f, x, t, rtol=1e-4, atol=1e-6)[-1].pow(2).mean()),
    "odeint_backprop_rk4": loop("odeint_backprop_rk4", lambda f, x: tda.odeint(f, x, t, method="rk4", options=dict(step_size=0.1))[-1].pow(2).mean()),
    "adjoint_eager": loop("adjoint_eager", lambda f, x: tda.odeint_adjoint(f, x, t, rtol=1e-4, atol=1e-6)[-1].pow(2).mean()),
    "adjoint_hip_graph": loop("adjoint_hip_graph", lambda f, x: tda.odeint_adjoint(f, x, t, rtol=1e-4, atol=1e-6, options=dict(hip_graph=True))[-1].pow(2).mean()),
    # r04: "auto" (lazy capture, side-effect fingerprint, replay-vs-eager probe) in a training loop, and complex states
    "adjoint_hip_graph_auto": loop("adjoint_hip_graph_auto", lambda f, x: tda.odeint_adjoint(f, x, t, rtol=1e-4, atol=1e-6, options=dict(hip_graph="auto"))[-1].pow(2).mean()),
    "odeint_nograd_auto": loop("odeint_nograd_auto", lambda f, x: (lambda y: f.net(y.detach()).pow(2).mean())(
        _nograd(lambda: tda.odeint(f, x, t, rtol=1e-4, atol=1e-6, options=dict(hip_graph="auto"))[-1]))),
    "complex_odeint": loop("complex_odeint", lambda f, x: (lambda y: f.net(y.real).pow(2).mean())(_nograd(lambda: tda.odeint(
        lambda t_, z: torch.complex(f.net(z.real), f.net(z.imag)) * 0.5, torch.complex(x, x.flip(1)), t, rtol=1e-4, atol=1e-6)[-1]))),
    "adjoint_tuple": loop("adjoint_tuple", lambda f, x: tda.odeint_adjoint(f, (x, torch.ones(8, device=dev)), t, rtol=1e-4, atol=1e-6)[0][-1].pow(2).mean()),
    "event": loop("event", lambda f, x: (lambda et, ys: et + ys[-1].pow(2).mean())(*tda.odeint_event(
        f, x, torch.tensor(0.0, device=dev), event_fn=lambda t_, y: 0.7 - t_ + 0.0 * y.sum(), rtol=1e-4, atol=1e-6))),
}
tda.clear_graph_cache()
res["after_clear_graph_cache_bytes"] = mem()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "soak_training.json"), "w"), indent=1)
worst = max(v["growth_bytes"] for v in res.values() if isinstance(v, dict))
print("worst growth", worst)
sys.exit(1 if worst > (1 << 20) else 0)
