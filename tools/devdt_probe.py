"""r06: why does a captured trial step's combine (`combine_devdt_kernel`, step size read on the device) take 5.3 us at 2^20
elements where the eager step's `stage_combine_kernel` takes 3.4-3.8 us (profiles/r06_shard_l2.json)?  The kernel, or the way
a hipGraph node is dispatched?  Four variants of the same 5-term combine over the same buffers, per-kernel durations from
torch.profiler and the wall time per launch:
    host_dt / stream,  dev_dt / stream,  host_dt / graph replay,  dev_dt / graph replay
each as a chain of 6 launches with a GEMM between them (y @ A.T, the cfg2 field — so that every combine reads what another
kernel on other XCDs has just written, as in the real step).  Prints one JSON object (-> profiles/r06_devdt_probe.json)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchdiffeq_amd import _native  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
kern = _native.get_kernels(dev, torch.float32)
res = {"unit": "us; kernel = average device duration of the combine (torch.profiler), wall = per chain of 6 combines + 6 GEMMs"}
for rows in (2048, 8192, 32768):
    n = rows * 128
    g = torch.Generator().manual_seed(0)
    A = (torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev)
    y0 = torch.randn(rows, 128, generator=g).to(dev)
    plan = kern.make_plan([(0, n, 1e-7, 1e-9)], n, 2048, dev)
    plan.ctrl_dev.copy_(torch.tensor([1.0, 0.01, 0.0, 0.01], dtype=torch.float64))
    coefs = [0.2, -0.3, 0.5, 0.1, 0.05]

    def chain(dev_dt):
        ks = [y0 @ A.T]
        for i in range(5):
            out = torch.empty_like(y0)
            if dev_dt:
                kern.stage_combine_dev(out.view(-1), None, y0.view(-1), [k.view(-1) for k in ks], coefs[:len(ks)], None, plan)
            else:
                kern.stage_combine(out.view(-1), y0.view(-1), [k.view(-1) for k in ks], coefs[:len(ks)], 0.01)
            ks.append(out @ A.T)
        return ks[-1]

    entry = {}
    for dev_dt in (False, True):
        for mode in ("stream", "graph"):
            with torch.no_grad():
                for _ in range(3):
                    chain(dev_dt)
                torch.cuda.synchronize()
                if mode == "graph":
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr):
                        chain(dev_dt)
                    run = gr.replay
                else:
                    run = lambda: chain(dev_dt)
                for _ in range(5):
                    run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    run()
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / 100
                from torch.profiler import ProfilerActivity, profile
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    for _ in range(20):
                        run()
                    torch.cuda.synchronize()
            per = {}
            for ev in prof.events():
                d = float(getattr(ev, "device_time", 0.0) or 0.0)
                if d > 0 and "tdeq::" in ev.name:
                    a = per.setdefault(ev.name.split("(")[0].replace("void tdeq::", "")[:48], [0, 0.0])
                    a[0] += 1
                    a[1] += d
            entry[("dev_dt" if dev_dt else "host_dt") + "/" + mode] = {
                "wall_us_per_chain": round(1e6 * wall, 2),
                "kernel_us": {k: round(v[1] / v[0], 2) for k, v in sorted(per.items())}}
    res[f"{rows}x128"] = entry
print(json.dumps(res, indent=1))
