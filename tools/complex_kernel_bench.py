"""Cold bandwidth of the complex norm kernels (csrc/tdeq_kernels_complex.hpp) next to their real counterparts at the same
byte count: cplx_error_norm_kernel<T, NT> on n complex elements vs error_norm_kernel<T, NT> on 2n reals (rotating buffer
sets larger than the Infinity Cache).  Usage (GPU box): python tools/complex_kernel_bench.py -> one JSON object."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchdiffeq_amd import _native  # noqa: E402

dev = torch.device("cuda:0")
N = 4 * 1024 * 1024            # complex elements (= 8 Mi reals)
SETS, LAUNCHES = 4, 24
out = {"n_complex": N, "hbm_peak_GBps": 8000.0}
for cdt, rdt, tag in ((torch.complex64, torch.float32, "c64"), (torch.complex128, torch.float64, "c128")):
    for nt in (1, 6, 9):
        ck = _native.get_kernels(dev, cdt)
        rk = _native.get_kernels(dev, rdt)
        res = {}
        for name, kern, dtype, n in (("complex", ck, cdt, N), ("real_same_bytes", rk, rdt, 2 * N)):
            sets = []
            for s in range(SETS):
                mk = lambda: (torch.randn(n, dtype=rdt, device=dev) if dtype == rdt else
                              torch.complex(torch.randn(n, dtype=rdt, device=dev), torch.randn(n, dtype=rdt, device=dev)))
                sets.append((mk(), mk(), [mk() for _ in range(nt)]))
            plan = kern.make_plan([(0, n, 1e-5, 1e-7)], n, 2048, dev)
            coefs = [0.1 * (j + 1) for j in range(nt)]
            for y0, y1, ks in sets:
                kern.error_norm(plan, y0, y1, ks, coefs, 0.05)
            kern.read_norms(plan)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(LAUNCHES):
                y0, y1, ks = sets[i % SETS]
                kern.error_norm(plan, y0, y1, ks, coefs, 0.05)
            e1.record()
            torch.cuda.synchronize()
            kern.read_norms(plan)
            us = 1e3 * e0.elapsed_time(e1) / LAUNCHES
            nbytes = (nt + 2) * n * (8 if rdt == torch.float64 else 4) * (2 if dtype != rdt else 1)
            res[name] = {"us_per_launch_incl_finalize": us, "GBps": nbytes / us / 1e3, "frac": nbytes / us / 1e3 / 8000.0}
            del sets
            torch.cuda.empty_cache()
        out[f"{tag}_error_norm_nt{nt}"] = res
print(json.dumps(out, indent=1))
