"""The reference's OWN test suite, unmodified, against this package (build container only: reads /root/reference/tests).

    PYTHONDONTWRITEBYTECODE=1 python tools/run_reference_tests.py [unittest args, e.g. -k test_adjoint or -v]

`import torchdiffeq` inside the reference's tests resolves to torchdiffeq_amd (an alias in sys.modules — nothing of
the reference package itself is imported); the test modules are loaded from where they lie, nothing is copied.  The
container has no GPU, so DEVICES = ['cpu'] there and every solve takes the host path for CPU states
(torchdiffeq_amd/_fallback.py) under the same solver / adjoint / event logic the HIP path runs.  On a box with a GPU the
same command also runs the 'cuda' half through the HIP kernels (the reference's tests are not available there)."""
import os
import sys
import unittest
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/tests"
sys.path.insert(0, ROOT)
import torchdiffeq_amd  # noqa: E402

assert "torchdiffeq" not in sys.modules
sys.modules["torchdiffeq"] = torchdiffeq_amd
sys.path.insert(0, REF_TESTS)
sys.dont_write_bytecode = True          # /root/reference is read-only
warnings.filterwarnings("ignore", category=torchdiffeq_amd.HostPathWarning)

if __name__ == "__main__":
    import api_tests, event_tests, gradient_tests, norm_tests, odeint_tests  # noqa: E401,E402
    suite = unittest.TestSuite()
    for mod in (api_tests, event_tests, gradient_tests, norm_tests, odeint_tests):
        suite.addTests(unittest.defaultTestLoader.loadTestsFromModule(mod))
    args = sys.argv[1:]
    verbosity = 2 if "-v" in args else 1
    if "-k" in args:
        pat = args[args.index("-k") + 1]
        flat = []

        def walk(s):
            for x in s:
                walk(x) if isinstance(x, unittest.TestSuite) else flat.append(x)
        walk(suite)
        suite = unittest.TestSuite([x for x in flat if pat in x.id()])
    res = unittest.TextTestRunner(verbosity=verbosity).run(suite)
    sys.exit(0 if res.wasSuccessful() else 1)
