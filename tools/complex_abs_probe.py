"""How ATen rounds |z| and z / real on this device — the measurement the complex variants of the norm kernels
(tdeq_error_norm*, tdeq_init_norms with TDEQ_C64 / TDEQ_C128) are written from.  Compares torch.abs(z) and z / tol with
candidate formulas built from separately rounded real operations, on 2^20 random values per dtype.
Usage: python tools/complex_abs_probe.py [cuda|cpu]  -> one JSON object."""
import json
import sys

import torch


def main(device):
    out = {"device": device, "torch": torch.__version__}
    g = torch.Generator().manual_seed(0)
    n = 1 << 20
    for cdt, rdt in ((torch.complex64, torch.float32), (torch.complex128, torch.float64)):
        mag = 10.0 ** (torch.rand(n, generator=g, dtype=torch.float64) * 12 - 6)
        re = (torch.randn(n, generator=g, dtype=torch.float64) * mag).to(rdt).to(device)
        im = (torch.randn(n, generator=g, dtype=torch.float64) * 10.0 ** (torch.rand(n, generator=g, dtype=torch.float64) * 12 - 6)).to(rdt).to(device)
        z = torch.complex(re, im)
        ref = torch.abs(z)
        wide = torch.float64 if rdt == torch.float32 else torch.float64
        cands = {
            "sqrt(re*re+im*im) in T": torch.sqrt(re * re + im * im),
            "hypot(re,im)": torch.hypot(re, im),
            "sqrt in fp64 rounded to T": torch.sqrt(re.to(wide) ** 2 + im.to(wide) ** 2).to(rdt),
            "scaled: m*sqrt(1+(s/m)^2)": torch.maximum(re.abs(), im.abs()) * torch.sqrt(
                1 + (torch.minimum(re.abs(), im.abs()) / torch.maximum(re.abs(), im.abs())) ** 2),
        }
        res = {k: int((v != ref).sum()) for k, v in cands.items()}
        tol = (torch.rand(n, generator=g, dtype=torch.float64) * 3 + 1e-3).to(rdt).to(device)
        q = z / tol
        inv = 1 / tol
        dv = {"(re/tol, im/tol)": int(((re / tol != q.real) | (im / tol != q.imag)).sum()),
              "(re*(1/tol), im*(1/tol))": int(((re * inv != q.real) | (im * inv != q.imag)).sum())}
        # scalar multiple: complex tensor * real Python number
        c = 0.3721
        p = z * c
        dv["z*c == (re*c, im*c)"] = int(((re * c != p.real) | (im * c != p.imag)).sum())
        # isfinite of a complex value
        out[str(cdt)] = {"abs_mismatches_of_%d" % n: res, "div_mismatches": dv}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ("cuda" if torch.cuda.is_available() else "cpu"))
