"""cfg3 backward with PyTorch's two BLAS back ends for the MLP's GEMMs: evaluation count and wall time of the
backward solve (the weight-gradient GEMM of the default hipBLASLt path is ~4x noisier in fp32 than rocBLAS's or the
CPU's — profiles/r02_field_noise.json — and the backward solve's step growth is set by exactly that noise)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchdiffeq_amd as tda  # noqa: E402
import _fullsize as fs  # noqa: E402

dev = torch.device("cuda:0")
res = {}
for lib in ("default", "hipblas"):
    if lib != "default":
        torch.backends.cuda.preferred_blas_library(lib)
    for case, rows in (("cfg3", None), ("cfg3_shard", slice(0, 8192))):
        z = fs.load(case)
        field, y0 = fs.cfg3_problem(rows)
        field, y0 = field.to(dev), y0.to(dev)
        t = torch.tensor([0.0, 1.0], device=dev)
        best = None
        for _ in range(3):
            for p in field.parameters():
                p.grad = None
            x = y0.clone().requires_grad_(True)
            y = tda.odeint_adjoint(field, x, t, rtol=1e-5, atol=1e-7, method="dopri5")
            field.nfe = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y[-1].pow(2).sum().backward()
            torch.cuda.synchronize()
            w = time.perf_counter() - t0
            best = w if best is None else min(best, w)
        idx = torch.from_numpy(z["rows"]).to(dev)
        res[f"{case}/{lib}"] = {"nfe_bwd": field.nfe, "ref_nfe_bwd": int(z["nfe_bwd"]), "bwd_ms": 1e3 * best,
                                "grad_y0_rel_err": fs.sample_rel_err(x.grad[idx], z["grad_y0_rows"], z["grad_y0_absmax"])}
print(json.dumps(res, indent=1))
