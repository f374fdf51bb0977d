"""The C-ABI is usable from plain C: tests/c_client/abi_client.c includes only include/tdeq_hip.h, links
libtdeq_hip.so and the HIP runtime, and checks two entry points element by element against host arithmetic.
CPU container: compile + link only (hipcc cross-compiles); GPU box: build and run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_client", "abi_client.c")


def _build(out_path):
    from torchdiffeq_amd import build as tbuild
    lib = tbuild.build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    libdir = os.path.dirname(lib)
    # -x c: the client is C, not HIP C++ — the header must be consumable by a C compiler
    subprocess.check_call([hipcc, "-x", "c", "-O1", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", SRC, "-o", out_path,
                           "-L" + libdir, "-ltdeq_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return out_path


def test_c_client_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path / "abi_client"))
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_client_runs_bit_exact(tmp_path):
    exe = _build(str(tmp_path / "abi_client"))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatching elements" in out.stdout
