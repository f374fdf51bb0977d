"""The reference's example scripts, unmodified, with `import torchdiffeq` resolving to this package — next to the same
scripts on the reference itself, same seeds, printed output compared (build container only: reads
/root/reference/examples; nothing is copied).

    PYTHONDONTWRITEBYTECODE=1 python tools/run_reference_examples.py [name ...]

Each case is one subprocess per library: seeds torch / numpy, installs the alias (package run only), sets sys.argv and
runs the script as __main__ from a scratch directory with a non-interactive matplotlib backend.  Numbers in the two
outputs (losses, event times, gradient-check lines) must agree to `rtol`; odenet_mnist.py needs the MNIST download and
is not run."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = "/root/reference/examples"

CASES = {     # name -> (script, argv, rtol on the printed numbers)
    "ode_demo": ("ode_demo.py", ["--niters", "40", "--test_freq", "10"], 2e-3),
    "ode_demo_adjoint": ("ode_demo.py", ["--niters", "20", "--test_freq", "10", "--adjoint"], 2e-3),
    "bouncing_ball": ("bouncing_ball.py", ["3"], 1e-4),
    "bouncing_ball_adjoint": ("bouncing_ball.py", ["3", "--adjoint"], 1e-4),
    "cnf": ("cnf.py", ["--niters", "4", "--num_samples", "64"], 2e-3),
    "cnf_adjoint": ("cnf.py", ["--niters", "4", "--num_samples", "64", "--adjoint"], 2e-3),
    "learn_physics": ("learn_physics.py", ["--num_iterations", "3"], 2e-3),
    "latent_ode": ("latent_ode.py", ["--niters", "4"], 2e-3),
}

RUNNER = r"""
import os, sys, runpy, random
import numpy as np, torch
lib, script, argv = sys.argv[1], sys.argv[2], sys.argv[3:]
torch.manual_seed(0); np.random.seed(0); random.seed(0)
torch.set_num_threads(1)
if lib == "package":
    sys.path.insert(0, %r)
    import torchdiffeq_amd
    sys.modules["torchdiffeq"] = torchdiffeq_amd
    import warnings
    warnings.filterwarnings("ignore", category=torchdiffeq_amd.HostPathWarning)
else:
    sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(script))
sys.dont_write_bytecode = True
sys.argv = [script] + argv
runpy.run_path(script, run_name="__main__")
""" % ROOT

NUM = re.compile(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?")


def numbers(text):
    """Numbers of the output, without the wall-clock fields some scripts print."""
    keep = []
    for line in text.splitlines():
        line = re.sub(r"(?i)(time|elapsed|sec)[^|,]*", "", line)
        keep += [float(x) for x in NUM.findall(line)]
    return keep


def run(lib, script, argv, cwd):
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", RUNNER, lib, os.path.join(EX, script)] + argv, cwd=cwd, env=env,
                       capture_output=True, text=True, timeout=3000)
    return r.returncode, r.stdout, r.stderr


def main():
    names = sys.argv[1:] or list(CASES)
    report, ok = {}, True
    for name in names:
        script, argv, rtol = CASES[name]
        outs = {}
        for lib in ("reference", "package"):
            with tempfile.TemporaryDirectory() as tmp:
                outs[lib] = run(lib, script, argv, tmp)
        (rc_r, out_r, err_r), (rc_p, out_p, err_p) = outs["reference"], outs["package"]
        a, b = numbers(out_r), numbers(out_p)
        worst = max((abs(x - y) / max(abs(x), 1e-6) for x, y in zip(a, b)), default=0.0)
        good = rc_r == 0 and rc_p == 0 and len(a) == len(b) and worst <= rtol
        ok &= good
        report[name] = {"argv": argv, "rc_reference": rc_r, "rc_package": rc_p, "numbers_compared": len(a),
                        "numbers_in_package_output": len(b), "worst_rel_diff": worst, "rtol": rtol, "ok": good,
                        "reference_output_tail": out_r.strip().splitlines()[-3:],
                        "package_output_tail": out_p.strip().splitlines()[-3:]}
        if rc_p != 0:
            report[name]["package_stderr_tail"] = err_p.strip().splitlines()[-6:]
        if rc_r != 0:
            report[name]["reference_stderr_tail"] = err_r.strip().splitlines()[-6:]
        print(name, json.dumps(report[name]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "reference_examples.json"), "w"), indent=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
