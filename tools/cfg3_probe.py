import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import torchdiffeq_amd as tda
dev=torch.device("cuda:0")
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(), torch.nn.Linear(256, 64)).to(dev)
class F(torch.nn.Module):
    def __init__(s):
        super().__init__(); s.net=net
    def forward(s,t,y): return s.net(y)
fm=F()
g = torch.Generator().manual_seed(1)
y0 = torch.randn(65536, 64, generator=g).to(dev)
adj = torch.randn(65536, 64, generator=g).to(dev)
params=tuple(fm.parameters())
def bare():
    y=y0.detach().requires_grad_(True)
    with torch.enable_grad():
        f=fm(None,y)
        return torch.autograd.grad(f,(y,)+params,adj)
for _ in range(3): bare()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): bare()
torch.cuda.synchronize(); print("bare MLP fwd+vjp per eval: %.3f ms"%((time.perf_counter()-t)/20*1e3))
tt=torch.tensor([0.0,1.0],device=dev)
def run():
    for p in params: p.grad=None
    x=y0.clone().requires_grad_(True)
    y=tda.odeint_adjoint(fm,x,tt,rtol=1e-5,atol=1e-7,method="dopri5")
    y[-1].pow(2).sum().backward()
run(); run(); torch.cuda.synchronize()
t=time.perf_counter(); run(); torch.cuda.synchronize(); print("fwd+bwd: %.1f ms"%((time.perf_counter()-t)*1e3))
