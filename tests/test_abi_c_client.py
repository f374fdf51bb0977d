"""The C-ABI is usable from plain C: tests/c_client/abi_client.c includes only include/tdeq_hip.h, links
libtdeq_hip.so and the HIP runtime, and checks two entry points element by element against host arithmetic.
CPU container: compile + link only (hipcc cross-compiles); GPU box: build and run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_client", "abi_client.c")


PREBUILT = os.path.join(ROOT, "tests", "c_client", "abi_client.bin")      # __graft_entry__.build() leaves it there


def _build(out_path, reuse_prebuilt=False):
    from torchdiffeq_amd import build as tbuild
    lib = tbuild.build()
    if reuse_prebuilt and os.path.exists(PREBUILT) and \
            os.path.getmtime(PREBUILT) >= max(os.path.getmtime(SRC), os.path.getmtime(lib),
                                              os.path.getmtime(os.path.join(ROOT, "include", "tdeq_hip.h"))):
        return PREBUILT             # built in the container next to the library (a cold hipcc costs ~30 s on the GPU box)
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


def build_prebuilt():
    """Called by __graft_entry__.build(): compile the C client next to its source (git-ignored, travels to the GPU box)."""
    return _build(PREBUILT)


def test_c_client_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path / "abi_client"))
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_client_runs_bit_exact(tmp_path):
    exe = _build(str(tmp_path / "abi_client"), reuse_prebuilt=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatching elements" in out.stdout
