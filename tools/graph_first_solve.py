"""Cost of the FIRST captured solves of a func object (capture + warm-up) vs later ones (cache hit) vs eager, cfg2/8
shard.  (GPU box)"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torchdiffeq_amd as tda  # noqa: E402
dev = torch.device("cuda:0")
A, y0 = bench.make_problem(dev, rows=slice(0, 8192))
At = A.T.contiguous()
tt = torch.tensor([0.0, 1.0], device=dev)
def solve(f, **o):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        y = tda.odeint(f, y0, tt, method="dopri5", options=o or None)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
res = {}
f = lambda t, y: y @ At
solve(f)
res["eager_ms"] = [solve(f) for _ in range(4)]
res["captured_same_func_ms"] = [solve(f, hip_graph=True) for _ in range(6)]
res["captured_fresh_func_each_time_ms"] = [solve((lambda t, y: y @ At), hip_graph=True) for _ in range(4)]
print(json.dumps(res, indent=1))
