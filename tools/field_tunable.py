"""cfg2's field (y @ A^T, 65536x128x128 fp32) as PyTorch dispatches it by default vs with PyTorch's TunableOp
(torch.cuda.tunable) picking the GEMM solution — the user's side of the step, 150 of its 335 us.  (GPU box)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
A, y0 = bench.make_problem(dev)
At = A.T.contiguous()
res = {}


def time_mm(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


res["default_us"] = time_mm(lambda: y0 @ At)
res["linear_us"] = time_mm(lambda: torch.nn.functional.linear(y0, A))
out = torch.empty_like(y0)
res["mm_out_us"] = time_mm(lambda: torch.mm(y0, At, out=out))
try:
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(30)
    tunable.set_max_tuning_iterations(20)
    t0 = time.perf_counter()
    (y0 @ At)
    torch.cuda.synchronize()
    res["tuning_s"] = time.perf_counter() - t0
    res["tunable_us"] = time_mm(lambda: y0 @ At)
    res["tunable_results"] = [list(map(str, r)) for r in tunable.get_results()][:4]
except Exception as exc:
    res["tunable_error"] = repr(exc)
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "field_tunable.json"), "w"), indent=1)
