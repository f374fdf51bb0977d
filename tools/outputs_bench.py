"""Many output times per solve (run on the GPU box): `odeint` wall time with the per-step multi-output dense
evaluation (tdeq_dense_eval_multi) vs one launch per output time (TDEQ_DENSE_MULTI=0), A/B in one process."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    import torchdiffeq_amd as tda
    import bench
    dev = torch.device("cuda:0")
    out = {}
    A, y0 = bench.make_problem(dev)
    At = A.T.contiguous()
    small_A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device=dev)
    cases = {
        "cfg2 state (65536x128 fp32), 101 output times": (lambda t, y: y @ At, y0, torch.linspace(0, 1, 101, device=dev),
                                                         dict(rtol=1e-7, atol=1e-9)),
        "spiral (ode_demo.py: 1x2 state), 1000 output times, dopri5": (lambda t, y: (y ** 3) @ small_A,
                                                                     torch.tensor([[2.0, 0.0]], device=dev),
                                                                     torch.linspace(0.0, 25.0, 1000, device=dev),
                                                                     dict(rtol=1e-7, atol=1e-9)),
    }
    for name, (f, y, t, kw) in cases.items():
        with torch.no_grad():
            tda.odeint(f, y, t, method="dopri5", **kw)
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                sol = tda.odeint(f, y, t, method="dopri5", **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
        out[name] = {"wall_s": best, "checksum": float(sol.double().sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("OB_CHILD") == "1":
        child()
        sys.exit(0)
    res = {}
    for multi in ("0", "1"):
        e = dict(os.environ, OB_CHILD="1", TDEQ_DENSE_MULTI=multi)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        res[f"TDEQ_DENSE_MULTI={multi}"] = json.loads(line[-1]) if line else {"error": r.stderr[-500:]}
        print(multi, res[f"TDEQ_DENSE_MULTI={multi}"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "outputs_bench.json"), "w"), indent=1)
