"""How noisy is cfg3's MLP (and its VJP) in fp32 on this GPU vs the CPU?  Relative RMS distance of the fp32
evaluation from the fp64 one, for the forward value and for the parameter / input VJPs, per BLAS backend."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _fullsize as fs  # noqa: E402

res = {}


def probe(device, tag):
    field, y0 = fs.cfg3_problem(slice(0, 8192))
    net32 = field.net.to(device)
    import copy
    net64 = copy.deepcopy(net32).double()
    y = y0.to(device)
    a = torch.randn(8192, 64, generator=torch.Generator().manual_seed(3)).to(device)

    def run(net, y, a):
        y = y.detach().requires_grad_(True)
        f = net(y)
        g = torch.autograd.grad(f, (y,) + tuple(net.parameters()), a)
        return f.detach(), g
    f32, g32 = run(net32, y, a)
    f64, g64 = run(net64, y.double(), a.double())
    rms = lambda x, r: float(((x.double() - r).pow(2).mean() / r.pow(2).mean()).sqrt())
    res[tag] = {"forward": rms(f32, f64), "vjp_y": rms(g32[0], g64[0]),
                "vjp_params": [rms(p, q) for p, q in zip(g32[1:], g64[1:])]}


probe("cpu", "cpu_fp32")
if torch.cuda.is_available():
    probe("cuda", "gpu_fp32_default_blas")
    try:
        torch.backends.cuda.preferred_blas_library("hipblas")
        probe("cuda", "gpu_fp32_rocblas")
    except Exception as exc:
        res["gpu_fp32_rocblas"] = repr(exc)
print(json.dumps(res, indent=1))
