"""Event solves (odeint_event, bouncing-ball style; plain and adjoint interface, tensor and tuple states, 8 methods) reference vs
package on the host path (build container): event time and solution must be BIT-identical, adjoint gradients too;
backprop gradients to rounding (fp64 1e-11; fp32 1e-5, reported otherwise — the start-time gradient cancels).  r04: 80 cases,
event times / solutions / adjoint gradients identical in every bit."""
import sys, torch, warnings, random
warnings.simplefilter("ignore")
sys.path.insert(0,'/root/reference'); import torchdiffeq as ref
sys.path.insert(0,'/root/repo'); import torchdiffeq_amd as tda
rng=random.Random(3)
bad=0; n=0
for case in range(80):
    dtype=rng.choice([torch.float32,torch.float64]); method=rng.choice(['dopri5','dopri8','bosh3','tsit5','rk4','euler','adaptive_heun','midpoint'])
    g=torch.Generator().manual_seed(rng.randrange(10**6))
    pos=(torch.rand(3,generator=g,dtype=torch.float64)*5+5).to(dtype); vel=torch.randn(3,generator=g,dtype=torch.float64).to(dtype)
    grav=torch.tensor(9.8,dtype=dtype)
    tup=rng.random()<0.5; adj=rng.random()<0.3
    thr=float(rng.uniform(0.5,3.0))
    opts={}
    if method in ('rk4','euler','midpoint'): opts['step_size']=rng.choice([0.01,0.05])
    else:
        opts['max_num_steps']=3000
        if rng.random()<0.3: opts['max_step']=0.1
    res=[]
    for L in (ref,tda):
        gp=torch.nn.Parameter(grav.clone())
        class B(torch.nn.Module):
            def __init__(s): super().__init__(); s.g=gp
            def forward(s,t,st):
                if tup: p,v=st; return v, -s.g*torch.ones_like(v)
                p,v=st[:3],st[3:]; return torch.cat([v, -s.g*torch.ones_like(v)])
        b=B()
        y0=(pos.clone().requires_grad_(True),vel.clone()) if tup else torch.cat([pos,vel]).requires_grad_(True)
        ev=(lambda t,st: st[0]-thr) if tup else (lambda t,st: st[:3]-thr)
        t0=torch.tensor(0.3,dtype=dtype).requires_grad_(True)
        try:
            et,sol=L.odeint_event(b,y0,t0,event_fn=ev,odeint_interface=L.odeint_adjoint if adj else L.odeint,method=method,atol=1e-6 if dtype==torch.float32 else 1e-9,rtol=1e-5 if dtype==torch.float32 else 1e-8,options=opts or None)
            s0=sol[0] if tup else sol
            (et+s0[-1].sum()).backward()
            lead=y0[0] if tup else y0
            res.append(('ok',[et.detach(),s0.detach(),lead.grad,gp.grad,t0.grad]))
        except Exception as e:
            res.append(('err',type(e).__name__+str(e)[:50]))
    a,b_=res
    n+=1
    if a[0]!=b_[0]: bad+=1; print('STATUS',case,method,a[1] if a[0]=='err' else 'ok',b_[1] if b_[0]=='err' else 'ok'); continue
    if a[0]=='err': continue
    eq=[(x is None and y is None) or (x is not None and y is not None and torch.equal(x,y)) for x,y in zip(a[1],b_[1])]
    if not all(eq):
        ds=[float((x-y).abs().max()/(x.abs().max()+1e-30)) if x is not None and y is not None else -1 for x,y in zip(a[1],b_[1])]
        if max(ds[:2])>0 or max(ds)>(1e-5 if dtype==torch.float32 else 1e-11) or adj: bad+=1; print('BITS',case,method,str(dtype)[6:],tup,adj,opts,['%.1e'%d for d in ds], flush=True)
print('cases',n,'bad',bad)
