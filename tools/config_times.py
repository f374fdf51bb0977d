"""Wall-clock of BASELINE.json's five configurations at full size on one MI355X (run through gpurun):
second call of each (first = warm-up: hipBLASLt heuristics, allocator), with NFE and step counts.
Prints one JSON object; the committed copy is profiles/<tag>_config_times.json."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchdiffeq_amd as tda  # noqa: E402
from _cases import StatFunc  # noqa: E402

dev = torch.device("cuda:0")


def linear(B, D, dtype):
    g = torch.Generator().manual_seed(0)
    G = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    A = 0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64)
    y0 = torch.randn(B, D, generator=g, dtype=torch.float64)
    return A.to(dtype).to(dev), y0.to(dtype).to(dev)


class Counting:
    """Counts evaluations WITHOUT the step callbacks of tests/_cases.StatFunc (callbacks switch the solver's
    look-ahead first stage off, which is not what a plain odeint call runs)."""

    def __init__(self, fn):
        self.fn, self.nfe = fn, 0

    def __call__(self, t, y):
        self.nfe += 1
        return self.fn(t, y)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return best, out


res = {}
# cfg1
A1 = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device=dev)
y01 = torch.tensor([[2.0, 0.0]], device=dev)
t1 = torch.linspace(0.0, 25.0, 1000, device=dev)
with torch.no_grad():
    w, y = timed(lambda: tda.odeint(lambda t, y: (y ** 3) @ A1, y01, t1, method="rk4"))
res["cfg1_rk4_spiral"] = {"wall_s": w, "steps": 999, "stages_per_s": 4 * 999 / w, "y_end": y[-1, 0].tolist()}
with torch.no_grad():
    wg, yg = timed(lambda: tda.odeint(lambda t, y: (y ** 3) @ A1, y01, t1, method="rk4", options=dict(hip_graph=True)))
res["cfg1_rk4_spiral_hip_graph"] = {"wall_s": wg, "steps": 999, "stages_per_s": 4 * 999 / wg,
                                    "bit_identical_to_eager": bool(torch.equal(y, yg))}

# cfg2
A, y0 = linear(65536, 128, torch.float32)
At = A.T.contiguous()
f = Counting(lambda t, y: y @ At)
tt = torch.tensor([0.0, 1.0], device=dev)
with torch.no_grad():
    w, y = timed(lambda: tda.odeint(f, y0, tt, method="dopri5"))
nfe = f.nfe // 4
exact = y0.double() @ torch.linalg.matrix_exp(A.double()).T
res["cfg2_dopri5_linear_fp32"] = {"wall_s": w, "nfe": nfe, "stages_per_s": (nfe - 2) / w,
                                   "rel_err_vs_expm": float((y[-1].double() - exact).abs().max() / exact.abs().max())}

# cfg4
A, y0 = linear(16384, 512, torch.float64)
At = A.T.contiguous()
f = Counting(lambda t, y: y @ At)
tt64 = torch.tensor([0.0, 1.0], dtype=torch.float64, device=dev)
with torch.no_grad():
    w, y = timed(lambda: tda.odeint(f, y0, tt64, method="dopri8", rtol=1e-9, atol=1e-11))
nfe = f.nfe // 4
exact = y0 @ torch.linalg.matrix_exp(A).T
res["cfg4_dopri8_linear_fp64"] = {"wall_s": w, "nfe": nfe, "stages_per_s": (nfe - 2) / w,
                                   "rel_err_vs_expm": float((y[-1] - exact).abs().max() / exact.abs().max())}

# cfg3 — SURVEY.md §8(d) inputs exactly (tests/_fullsize.py; r01 drew y0 from another seed, hence its NFE 86 vs the
# reference's 74 was not comparable — see docs/LAB_NOTEBOOK.md §6), full batch and the 1/8 shard of an 8-GPU strong-scaling run
import _fullsize as fs  # noqa: E402


def adjoint_times(rows, label):
    fm, y03 = fs.cfg3_problem(rows)
    fm, y03 = fm.to(dev), y03.to(dev)
    state = {}

    def fwd():
        fm.nfe = 0
        x = y03.clone().requires_grad_(True)
        y = tda.odeint_adjoint(fm, x, tt, rtol=1e-5, atol=1e-7, method="dopri5")
        state["y"], state["nfe_fwd"] = y, fm.nfe
        return y

    def fwd_bwd():
        for p in fm.parameters():
            p.grad = None
        y = fwd()
        torch.cuda.synchronize()
        t = time.perf_counter()
        fm.nfe = 0
        y[-1].pow(2).sum().backward()
        torch.cuda.synchronize()
        state["bwd_s"], state["nfe_bwd"] = time.perf_counter() - t, fm.nfe

    wf, _ = timed(fwd)
    fwd_bwd()
    fwd_bwd()
    ref = fs.load(label)
    res[label + "_adjoint_mlp"] = {"rows": y03.shape[0], "fwd_wall_s": wf, "bwd_wall_s": state["bwd_s"],
                                   "nfe_fwd": state["nfe_fwd"], "nfe_bwd": state["nfe_bwd"],
                                   "reference_nfe_fwd": int(ref["nfe_fwd"]), "reference_nfe_bwd": int(ref["nfe_bwd"]),
                                   "reference_wall_s_1thread": ref["wall_s_1thread"].tolist()}


adjoint_times(None, "cfg3")
adjoint_times(slice(0, 8192), "cfg3_shard")

# cfg2 / 8: the linear field on the 8192 x 128 shard, eager and captured
A, y0 = linear(65536, 128, torch.float32)
At, y0s = A.T.contiguous(), y0[:8192].contiguous()
f = Counting(lambda t, y: y @ At)
with torch.no_grad():
    w, y = timed(lambda: tda.odeint(f, y0s, tt, method="dopri5"))
    nfe = f.nfe // 4
    wg, yg = timed(lambda: tda.odeint(lambda t, y: y @ At, y0s, tt, method="dopri5", options=dict(hip_graph="auto")))
res["cfg2_shard_dopri5_linear_fp32"] = {"rows": 8192, "wall_s": w, "nfe": nfe, "stages_per_s": (nfe - 2) / w,
                                         "wall_s_hip_graph_auto": wg, "same_result": bool(torch.equal(y, yg))}

# cfg5 (the example's CNF at its seeded init; closed-form trace), SURVEY inputs
z = fs.load("cfg5")
cnf = fs.ExampleCNF([z[f"p{i}"] for i in range(6)], trace="closed").to(dev)
z0 = fs.cfg5_problem()[0].to(dev)
t5 = torch.tensor([10.0, 0.0], device=dev)
state = {}


def cnf_fwd_bwd():
    for p in cnf.parameters():
        p.grad = None
    x = z0.clone().requires_grad_(True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    zt, lp = tda.odeint_adjoint(cnf, (x, torch.zeros(32768, 1, device=dev)), t5, atol=1e-5, rtol=1e-5, method="dopri5")
    torch.cuda.synchronize()
    state["cnf_fwd"] = time.perf_counter() - t
    t = time.perf_counter()
    (lp[-1].mean() - zt[-1].pow(2).sum() / 100).backward()
    torch.cuda.synchronize()
    state["cnf_bwd"] = time.perf_counter() - t


cnf_fwd_bwd()
cnf_fwd_bwd()
res["cfg5_cnf_adjoint"] = {"fwd_wall_s": state["cnf_fwd"], "bwd_wall_s": state["cnf_bwd"]}


def cnf_fwd_graph():
    x = z0.clone().requires_grad_(True)
    return tda.odeint_adjoint(cnf, (x, torch.zeros(32768, 1, device=dev)), t5, atol=1e-5, rtol=1e-5, method="dopri5",
                              options=dict(hip_graph=True))


wg5, _ = timed(cnf_fwd_graph)
res["cfg5_cnf_adjoint"]["fwd_wall_s_hip_graph"] = wg5
print(json.dumps(res, indent=1))
