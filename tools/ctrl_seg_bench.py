"""The one-workgroup finalize + controller kernel (norm_finalize_ctrl_kernel) on segmented states: microseconds per
launch of error_norm_partial_ctrl minus the plain partial-norm launch, for the 9-segment augmented state of cfg3's
backward solve (L = 8.49 M), a 3-segment and a 16-segment state and the plain tensor.  (run on the GPU box)"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchdiffeq_amd import _native  # noqa: E402
from torchdiffeq_amd.misc import StateLayout  # noqa: E402
from torchdiffeq_amd.tableaus import DOPRI5  # noqa: E402

dev = torch.device("cuda:0")
kern = _native.get_kernels(dev, torch.float32)
cases = {
    "plain 8.4M": [torch.Size((65536, 128))],
    "cfg3 backward (9 segments)": [torch.Size(s) for s in ((), (65536, 64), (65536, 64), (256, 64), (256,), (256, 256), (256,), (64, 256), (64,))],
    "3 segments": [torch.Size(s) for s in ((), (65536, 64), (65536, 64))],
    "16 segments": [torch.Size((32768, 16))] * 16,
}
res = {}
for name, shapes in cases.items():
    lay = StateLayout(shapes, len(shapes) > 1)
    n = lay.total
    g = torch.Generator().manual_seed(0)
    y0 = torch.randn(n, generator=g).to(dev)
    y1 = (y0 + 0.01).contiguous()
    part = (torch.randn(n, generator=g) * 1e-7).to(dev)
    k6 = (torch.randn(n, generator=g) * 1e-7).to(dev)
    plan = kern.make_plan(lay.segments(1e-7, 1e-9), n, lay.chunk, dev)
    c = _native.StepCtrl()
    c.t0, c.dt, c.safety, c.ifactor, c.dfactor, c.exponent = 0.3, 0.05, 0.9, 10.0, 0.2, 0.2
    c.min_step, c.max_step, c.time_sign = 0.0, math.inf, 1.0
    mask = 0
    for i, a in enumerate(DOPRI5.alpha):
        c.alpha[i] = float(np.float32(a))
        mask |= (1 << i) if a == 1.0 else 0
    c.alpha_is_one, c.n_times, c.n_norm_seg = mask, 6, lay.n_seg
    tn = torch.empty(16, device=dev)
    out = {}
    for key, fn in (("plain", lambda: kern.error_norm_partial(plan, part, y0, y1, [k6], [0.025], 0.05)),
                    ("ctrl", lambda: kern.error_norm_partial_ctrl(plan, part, y0, y1, [k6], [0.025], 0.05, c, tn))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        plan.expect = ()
        out[key + "_us"] = 1e3 * e0.elapsed_time(e1) / 200
    res[name] = out
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ctrl_seg_bench.json"), "w"), indent=1)
