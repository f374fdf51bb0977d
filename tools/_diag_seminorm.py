import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torchdiffeq_amd as tda
from test_reference_suite import _NeuralF
torch.set_default_device("cuda")
dtype = torch.float32; method = "fehlberg2"; tol = 1e-6
x0 = torch.tensor([1.0, 2.0], dtype=dtype); t = torch.tensor([0.0, 1.0], dtype=torch.float64)
ode_f = _NeuralF(width=1024, oscillate=True, freq=2).to("cuda", dtype)
out = tda.odeint_adjoint(ode_f, x0, t, atol=tol, rtol=tol, method=method); fwd1 = ode_f.nfe
ode_f.nfe = 0; out.sum().backward(); d = ode_f.nfe
out = tda.odeint_adjoint(ode_f, x0, t, atol=tol, rtol=tol, method=method, adjoint_options=dict(norm="seminorm")); fwd2 = ode_f.nfe - d
ode_f.nfe = 0; out.sum().backward(); s = ode_f.nfe
from torchdiffeq_amd import _graph
print(os.environ.get("TDEQ_HIP_GRAPH"), "fwd", fwd1, fwd2, "default", d, "seminorm", s, "refused", ode_f in _graph._GraphStep._refused)
