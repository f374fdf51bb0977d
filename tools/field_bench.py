"""Which plain-PyTorch spelling of the cfg2 vector field  f(t, y) = y A^T  (65536 x 128 x 128, fp32) does
PyTorch-ROCm run fastest?  (run on the GPU box)   The field is user code — a GEMM dispatched by PyTorch to
hipBLASLt / rocBLAS, not one of this package's kernels — but bench.py has to spell it somehow."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    A, y0 = bench.make_problem(dev)
    At = A.T.contiguous()
    ys = [torch.randn_like(y0) for _ in range(8)]      # rotate inputs: 8 x 33.5 MB > Infinity Cache
    variants = {
        "y @ At (At = A.T.contiguous())": lambda y: y @ At,
        "torch.mm(y, At)": lambda y: torch.mm(y, At),
        "y @ A.T (strided view)": lambda y: y @ A.T,
        "F.linear(y, A)": lambda y: F.linear(y, A),
        "torch.einsum('bi,oi->bo')": lambda y: torch.einsum("bi,oi->bo", y, A),
        "torch.addmm(zero_bias, y, At)": None,
    }
    zb = torch.zeros(128, device=dev)
    variants["torch.addmm(zero_bias, y, At)"] = lambda y: torch.addmm(zb, y, At)
    ref = (ys[0].double() @ A.double().T)
    out = {}
    libs = ["default"]
    if hasattr(torch.backends.cuda, "preferred_blas_library"):
        libs = ["hipblaslt", "hipblas"]
    for lib in libs:
        if lib != "default":
            try:
                torch.backends.cuda.preferred_blas_library(lib)
            except Exception as exc:
                out[lib] = repr(exc)
                continue
        for name, fn in variants.items():
            try:
                for i in range(5):
                    r = fn(ys[i % 8])
                err = float((fn(ys[0]).double() - ref).abs().max() / ref.abs().max())
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 200
                e0.record()
                for i in range(reps):
                    r = fn(ys[i % 8])
                e1.record()
                torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / reps
                out[f"{lib}: {name}"] = {"us": us, "rel_err_vs_fp64": err}
                print(f"{lib:10s} {name:36s} {us:7.2f} us   rel-err vs fp64 {err:.2e}", flush=True)
            except Exception as exc:
                out[f"{lib}: {name}"] = repr(exc)
                print(lib, name, "failed:", repr(exc)[:200], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "field_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
