"""Build libtdeq_hip.so for gfx950:  python -m torchdiffeq_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is built in-tree (next to this file) so that it
travels with the source snapshot; it is git-ignored."""
from __future__ import annotations

import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SOURCES = [os.path.join(_PKG, "csrc", "tdeq_abi.hip")]
HEADERS = [os.path.join(_PKG, "csrc", h) for h in ("tdeq_kernels.hpp", "tdeq_kernels_complex.hpp", "tdeq_kernels_lp.hpp",
                                                   "tdeq_abi_lp.hpp")] + [
           os.path.join(_ROOT, "include", "tdeq_hip.h")]
OUTPUT = os.path.join(_PKG, "libtdeq_hip.so")

# -ffp-contract=off: every product and sum is rounded separately, like the reference's eager ops.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_PKG, "csrc")]


def needs_build() -> bool:
    if not os.path.exists(OUTPUT):
        return True
    out_m = os.path.getmtime(OUTPUT)
    return any(os.path.getmtime(p) > out_m for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUTPUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + SOURCES + ["-o", OUTPUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUTPUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
