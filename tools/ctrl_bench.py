"""Where does norm_finalize_ctrl_kernel spend its time?  (run on the GPU box)

Times `error_norm_partial` (plain finalize) against `error_norm_partial_ctrl` on the cfg2 state under input /
configuration variants that switch individual parts of the controller kernel off:
  readback=copy   results go to device memory instead of pinned host memory (no PCIe stores)
  zero error      ratio == 0 -> the pow() is skipped
  n_times=1       one stage time instead of six
Each figure is the average over back-to-back launches (HIP events), so the common partial kernel cancels in the
differences."""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_variant():
    import numpy as np
    import torch
    from torchdiffeq_amd import _native
    from torchdiffeq_amd.tableaus import DOPRI5
    n = int(os.environ.get("CB_N", 65536 * 128))
    kern = _native.get_kernels(torch.device("cuda:0"))
    g = torch.Generator().manual_seed(0)
    y0 = torch.randn(n, generator=g).cuda()
    y1 = (y0 + 0.01).contiguous()
    zero = os.environ.get("CB_ZERO") == "1"
    part = torch.zeros(n).cuda() if zero else (torch.randn(n, generator=g) * 1e-7).cuda()
    k6 = torch.zeros(n).cuda() if zero else (torch.randn(n, generator=g) * 1e-7).cuda()
    plan = kern.make_plan([(0, n, 1e-7, 1e-9)], n, _native.pick_chunk(n), torch.device("cuda:0"))
    c = _native.StepCtrl()
    c.t0, c.dt, c.safety, c.ifactor, c.dfactor, c.exponent = 0.3, 0.05, 0.9, 10.0, 0.2, 0.2
    c.min_step, c.max_step, c.time_sign = 0.0, math.inf, 1.0
    n_times = int(os.environ.get("CB_NTIMES", 6))
    mask = 0
    for i, a in enumerate(DOPRI5.alpha[:n_times]):
        c.alpha[i] = float(np.float32(a))
        mask |= (1 << i) if a == 1.0 else 0
    c.alpha_is_one, c.n_times, c.n_norm_seg = mask, n_times, 1
    tn = torch.empty(16, device="cuda")
    reps = 300
    out = {}
    for name, fn in (("plain", lambda: kern.error_norm_partial(plan, part, y0, y1, [k6], [0.025], 0.05)),
                     ("ctrl", lambda: kern.error_norm_partial_ctrl(plan, part, y0, y1, [k6], [0.025], 0.05, c, tn))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = 1e3 * e0.elapsed_time(e1) / reps
    out["ctrl_minus_plain_us"] = out["ctrl"] - out["plain"]
    print(json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("CB_CHILD") == "1":
        run_variant()
        sys.exit(0)
    results = {}
    for label, env in [("default", {}), ("readback=copy", {"TDEQ_READBACK": "copy"}), ("zero error (no pow)", {"CB_ZERO": "1"}),
                       ("n_times=1", {"CB_NTIMES": "1"}), ("zero+copy+n_times=1", {"CB_ZERO": "1", "TDEQ_READBACK": "copy",
                                                                                   "CB_NTIMES": "1"}),
                       ("small n (1 chunk)", {"CB_N": "2048"})]:
        e = dict(os.environ, CB_CHILD="1", **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        results[label] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
        print(label, results[label], flush=True)
    json.dump(results, open(os.path.join(ROOT, "gpurun_out", "ctrl_bench.json"), "w"), indent=1)
