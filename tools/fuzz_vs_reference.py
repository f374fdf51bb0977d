"""Differential fuzzing against the reference (build container only: imports /root/reference).

  PYTHONDONTWRITEBYTECODE=1 python tools/fuzz_vs_reference.py {fixed|adaptive|adjoint|backprop|event|complex|tableau|eventgrad|callbacks|hessian|vectol|brow|hostexact} [seed] [cases]
  (TDEQ_FUZZ_BACKEND=host: the product's torch-op host path for CPU states instead of the oracle test backend)

Random states (0-dim .. 3-dim, tuples), dtypes, time grids (both directions), options and methods are solved by the
reference and by torchdiffeq_amd (host logic over the CPU oracle kernels — the same code path the `dev="cpu"` tests
use) and compared: bit for bit for the explicit fixed-grid and Adams methods, to the nonlinear solver's tolerance for
the implicit RK methods, 1e-9 for the adaptive methods, 1e-9 / 1e-6 for gradients.  Test infrastructure, like oracle/.
Last runs (round 1): fixed 1050 cases, adaptive 400, adjoint 220, backprop 120, event 100 — no mismatch other than
non-converged implicit solves (both libraries warn), rounding-level iteration-count flips, dopri8's noise-level first
step (docs/LAB_NOTEBOOK.md §12) and the 0-dim fp32 state on an fp64 grid together with `perturb` (docs/LAB_NOTEBOOK.md §8)."""
import os
import random
import sys
import warnings

import torch

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/reference")
import torchdiffeq as ref  # noqa: E402
import torchdiffeq_amd as tda  # noqa: E402
from torchdiffeq_amd import _native  # noqa: E402
from oracle.kernels import OracleKernels  # noqa: E402

ok = OracleKernels()
if os.environ.get("TDEQ_FUZZ_BACKEND", "oracle") == "host":
    # r03: the package's own torch-op host path (torchdiffeq_amd/_fallback.py) instead of the test backend
    warnings.simplefilter("ignore")
else:
    _native.get_kernels = lambda d, dtype=None: ok
torch.set_num_threads(1)
mode = sys.argv[1] if len(sys.argv) > 1 else "fixed"
sys.argv = [sys.argv[0]] + sys.argv[2:]
if len(sys.argv) < 3:
    sys.argv += ["0", "100"][len(sys.argv) - 1:]

if mode == "fixed":
    seed = int(sys.argv[1]) if len(sys.argv)>1 else 0
    n_cases = int(sys.argv[2]) if len(sys.argv)>2 else 200
    rng = random.Random(seed)
    FIXED = ['euler','midpoint','heun2','heun3','rk4','explicit_adams','implicit_adams','fixed_adams']
    IMPL = ['implicit_euler','implicit_midpoint','trapezoid','radauIIA3','gl4','radauIIA5','gl6','sdirk2','trbdf2']
    bad=0
    for case in range(n_cases):
        method = rng.choice(FIXED+IMPL)
        dtype = rng.choice([torch.float32, torch.float64])
        tdtype = rng.choice([dtype, torch.float64]) if rng.random()<0.3 else dtype
        shape = rng.choice([(), (1,), (3,), (2,3), (4,1,2)])
        is_tuple = rng.random()<0.25
        rev = rng.random()<0.4
        npts = rng.choice([2,3,5,9])
        g = torch.Generator().manual_seed(rng.randrange(10**6))
        y0 = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype)
        t = torch.sort(torch.rand(npts, generator=g, dtype=torch.float64)).values.to(tdtype)
        if float((t[1:]-t[:-1]).min()) < 1e-3: continue
        if rev: t = t.flip(0)
        opts={}
        r=rng.random()
        if r<0.4: opts['step_size']=rng.choice([0.05,0.013,0.1,0.5])
        elif r<0.5: opts['grid_constructor']=lambda f,y,tt: torch.cat([tt[:1], (tt[:1]+tt[-1:])/2, tt[-1:]]) 
        if rng.random()<0.3: opts['perturb']=True
        if rng.random()<0.3: opts['interp']='cubic'
        if 'adams' in method:
            if rng.random()<0.3: opts['max_order']=rng.choice([4,5,8,12,3])
            if rng.random()<0.3 and method!='explicit_adams': opts['max_iters']=rng.choice([1,2,6])
        if method in IMPL and rng.random()<0.3: opts['max_iters']=rng.choice([1,3,50])
        lam = rng.choice([0.3,1.0,3.0])
        if is_tuple:
            y0b = torch.randn(2, generator=g, dtype=torch.float64).to(dtype)
            f = lambda t_,y_: ((1-t_*0.5)*(y_[0]*(-lam)) - y_[0]*y_[0]*y_[0]*0.01 + y_[1].sum()*0.01, -y_[1]*0.5*(1+t_))
            yy=(y0,y0b)
        else:
            f = lambda t_,y_: (1-t_*0.5)*(y_*(-lam)) - y_*y_*y_*0.01
            yy=y0
        kw={}
        if rng.random()<0.5: kw=dict(rtol=rng.choice([1e-3,1e-6]), atol=rng.choice([1e-4,1e-8]))
        res=[]
        for lib in (ref,tda):
            try:
                with warnings.catch_warnings(record=True) as w, torch.no_grad():
                    warnings.simplefilter('always')
                    out = lib.odeint(f, yy, t, method=method, options=dict(opts), **kw)
                res.append(('ok', out, len(w)))
            except Exception as e:
                res.append(('err', type(e).__name__+': '+str(e)[:80], 0))
        a,b=res
        desc=(method,str(dtype)[6:],str(tdtype)[6:],shape,is_tuple,rev,npts,{k:(v if not callable(v) else 'fn') for k,v in opts.items()},kw)
        if a[0]!=b[0]:
            bad+=1; print('STATUS MISMATCH', desc, a[0], a[1] if a[0]=='err' else '', b[0], b[1] if b[0]=='err' else ''); continue
        if a[0]=='err':
            if a[1].split(':')[0]!=b[1].split(':')[0]: bad+=1; print('ERR TYPE MISMATCH', desc, a[1], b[1])
            continue
        oa = a[1] if is_tuple else (a[1],); ob = b[1] if is_tuple else (b[1],)
        tol = (1e-5 if dtype==torch.float32 else 1e-11) if method in IMPL else 0.0
        for x,y in zip(oa,ob):
            if x.dtype!=y.dtype or x.shape!=y.shape:
                bad+=1; print('META MISMATCH', desc, x.dtype,y.dtype,x.shape,y.shape); break
            d = float((x-y).abs().max()/ (x.abs().max()+1e-30)) if x.numel() else 0.0
            nanmismatch = bool((torch.isnan(x)!=torch.isnan(y)).any())
            if (d>tol and not (d!=d)) or nanmismatch:
                bad+=1; print('VALUE MISMATCH', desc, d); break
        if a[2]!=b[2]:
            print('warn count differs', desc, a[2], b[2])
    print('done', n_cases, 'bad', bad)
elif mode == "adaptive":
    seed = int(sys.argv[1]); n_cases=int(sys.argv[2])
    rng = random.Random(seed)
    AD = ['dopri5','dopri8','tsit5','bosh3','fehlberg2','adaptive_heun']
    bad=0
    for case in range(n_cases):
        method=rng.choice(AD); dtype=torch.float64
        shape=rng.choice([(),(1,),(3,),(2,3),(17,5)])
        is_tuple=rng.random()<0.3; rev=rng.random()<0.4; npts=rng.choice([2,3,6,15])
        g=torch.Generator().manual_seed(rng.randrange(10**6))
        y0=torch.randn(shape,generator=g,dtype=torch.float64)
        t=torch.sort(torch.rand(npts,generator=g,dtype=torch.float64)*3).values
        if float((t[1:]-t[:-1]).min())<1e-3: continue
        if rev: t=t.flip(0)
        opts={}
        if rng.random()<0.2: opts['first_step']=rng.choice([0.01,0.2])
        if rng.random()<0.2: opts['max_step']=rng.choice([0.05,0.3])
        if rng.random()<0.15: opts['min_step']=rng.choice([1e-3,0.05])
        if rng.random()<0.2: opts['safety']=0.8; opts['ifactor']=5.0; opts['dfactor']=0.3
        if rng.random()<0.2:
            lo,hi=float(t.min()),float(t.max())
            opts[rng.choice(['step_t','jump_t'])]=torch.tensor([lo+(hi-lo)*0.37, lo+(hi-lo)*0.71],dtype=torch.float64)
        lam=rng.choice([0.3,1.0,3.0])
        if is_tuple:
            y0b=torch.randn(2,generator=g,dtype=torch.float64)
            f=lambda t_,y_: (torch.sin(t_*2)*y_[0]*lam - y_[0]**3*0.1 + y_[1].sum()*0.01, -y_[1]*0.5*(1+t_))
            yy=(y0,y0b)
        else:
            f=lambda t_,y_: torch.sin(t_*2)*y_*lam - y_**3*0.1
            yy=y0
        kw=dict(rtol=rng.choice([1e-4,1e-7,1e-9]),atol=rng.choice([1e-6,1e-9,1e-11]))
        res=[]
        import time as _time
        for lib in (ref,tda):
            n=[0]
            def ff(t_,y_):
                n[0]+=1; return f(t_,y_)
            _t0=_time.perf_counter()
            try:
                with warnings.catch_warnings(record=True) as w, torch.no_grad():
                    warnings.simplefilter('always')
                    out=lib.odeint(ff,yy,t,method=method,options=dict(opts),**kw)
                res.append(('ok',out,n[0]))
            except Exception as e:
                res.append(('err',type(e).__name__+': '+str(e)[:80],n[0]))
            if os.environ.get('FUZZ_VERBOSE'):
                print('case',case,lib.__name__,method,'nfe',n[0],'%.1fs'%(_time.perf_counter()-_t0),flush=True)
        a,b=res
        desc=(method,shape,is_tuple,rev,npts,{k:(v.tolist() if torch.is_tensor(v) else v) for k,v in opts.items()},kw)
        if a[0]!=b[0]: bad+=1; print('STATUS',desc,a[:2] if a[0]=='err' else 'ok',b[:2] if b[0]=='err' else 'ok'); continue
        if a[0]=='err':
            if a[1].split(':')[0]!=b[1].split(':')[0]: bad+=1; print('ERRTYPE',desc,a[1],b[1])
            continue
        oa=a[1] if is_tuple else (a[1],); ob=b[1] if is_tuple else (b[1],)
        for x,y in zip(oa,ob):
            d=float((x-y).abs().max()/(x.abs().max()+1e-30))
            if d>1e-9 or x.shape!=y.shape: bad+=1; print('VALUE',desc,d,a[2],b[2]); break
        else:
            if a[2]!=b[2]: print('nfe differs',desc,a[2],b[2])
    print('done',n_cases,'bad',bad)
elif mode == "adjoint":
    seed=int(sys.argv[1]); n_cases=int(sys.argv[2])
    rng=random.Random(seed)
    M=['euler','rk4','explicit_adams','implicit_adams','implicit_midpoint','gl4','radauIIA5','sdirk2','trbdf2','dopri5','bosh3']
    bad=0
    for case in range(n_cases):
        method=rng.choice(M); amethod=rng.choice([None,None,rng.choice(M)])
        tgrad=rng.random()<0.4; rev=rng.random()<0.3; is_tuple=rng.random()<0.25
        npts=rng.choice([2,3,6])
        sd=rng.randrange(10**6)
        res=[]
        for lib in (ref,tda):
            torch.manual_seed(sd)
            lin=torch.nn.Linear(3,3).double()
            y0=torch.randn(4,3,dtype=torch.float64).requires_grad_(True)
            y0b=torch.randn(2,dtype=torch.float64).requires_grad_(True)
            t=torch.sort(torch.rand(npts,dtype=torch.float64)).values
            if float((t[1:]-t[:-1]).min())<0.02: res=None; break
            if rev: t=t.flip(0)
            t=t.detach().requires_grad_(tgrad)
            class F(torch.nn.Module):
                def __init__(s): super().__init__(); s.lin=lin
                def forward(s,t_,y_):
                    if is_tuple: return (torch.tanh(s.lin(y_[0]))*torch.cos(t_)+y_[1].sum()*0.1, -y_[1]*0.5)
                    return torch.tanh(s.lin(y_))*torch.cos(t_)
            yy=(y0,y0b) if is_tuple else y0
            opts={} if method in('dopri5','bosh3') else {'step_size':0.05}
            kw={}
            if amethod is not None:
                kw['adjoint_method']=amethod
                kw['adjoint_options']={} if amethod in('dopri5','bosh3') else {'step_size':0.05}
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    out=lib.odeint_adjoint(F(),yy,t,method=method,options=opts,rtol=1e-7,atol=1e-9,**kw)
                    o=out[0] if is_tuple else out
                    loss=(o[-1]**2).sum()+ (out[1].sum() if is_tuple else 0)
                    ins=[y0,*lin.parameters()]+([y0b] if is_tuple else [])+([t] if tgrad else [])
                    g=torch.autograd.grad(loss,ins,allow_unused=True)
                res.append(('ok',[o.detach()]+[x if x is not None else torch.zeros(1,dtype=torch.float64) for x in g]))
            except Exception as e:
                res.append(('err',type(e).__name__+': '+str(e)[:100]))
        if res is None: continue
        a,b=res
        desc=(method,amethod,tgrad,rev,is_tuple,npts)
        if a[0]!=b[0]: bad+=1; print('STATUS',desc,a[1] if a[0]=='err' else 'ok',b[1] if b[0]=='err' else 'ok'); continue
        if a[0]=='err': continue
        impl = any(m in ('implicit_midpoint','gl4','radauIIA5','sdirk2','trbdf2') for m in (method,amethod) if m)
        tol = 1e-6 if impl else 1e-9
        for i,(x,y) in enumerate(zip(a[1],b[1])):
            d=float((x-y).abs().max()/(x.abs().max()+1e-12))
            if d>tol: bad+=1; print('VALUE',desc,i,d); break
    print('done',n_cases,'bad',bad)
elif mode == "backprop":
    seed=int(sys.argv[1]); n_cases=int(sys.argv[2])
    rng=random.Random(seed)
    M=['euler','rk4','explicit_adams','implicit_adams','implicit_midpoint','gl4','radauIIA5','sdirk2','trbdf2','dopri5','bosh3']
    bad=0
    for case in range(n_cases):
        method=rng.choice(M); amethod=None
        tgrad=rng.random()<0.4; rev=rng.random()<0.3; is_tuple=rng.random()<0.25
        npts=rng.choice([2,3,6])
        sd=rng.randrange(10**6)
        res=[]
        for lib in (ref,tda):
            torch.manual_seed(sd)
            lin=torch.nn.Linear(3,3).double()
            y0=torch.randn(4,3,dtype=torch.float64).requires_grad_(True)
            y0b=torch.randn(2,dtype=torch.float64).requires_grad_(True)
            t=torch.sort(torch.rand(npts,dtype=torch.float64)).values
            if float((t[1:]-t[:-1]).min())<0.02: res=None; break
            if rev: t=t.flip(0)
            t=t.detach().requires_grad_(tgrad)
            class F(torch.nn.Module):
                def __init__(s): super().__init__(); s.lin=lin
                def forward(s,t_,y_):
                    if is_tuple: return (torch.tanh(s.lin(y_[0]))*torch.cos(t_)+y_[1].sum()*0.1, -y_[1]*0.5)
                    return torch.tanh(s.lin(y_))*torch.cos(t_)
            yy=(y0,y0b) if is_tuple else y0
            opts={} if method in('dopri5','bosh3') else {'step_size':0.05}
            kw={}
            if amethod is not None:
                kw['adjoint_method']=amethod
                kw['adjoint_options']={} if amethod in('dopri5','bosh3') else {'step_size':0.05}
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    out=lib.odeint(F(),yy,t,method=method,options=opts,rtol=1e-7,atol=1e-9)
                    o=out[0] if is_tuple else out
                    loss=(o[-1]**2).sum()+ (out[1].sum() if is_tuple else 0)
                    ins=[y0,*lin.parameters()]+([y0b] if is_tuple else [])+([t] if tgrad else [])
                    g=torch.autograd.grad(loss,ins,allow_unused=True)
                res.append(('ok',[o.detach()]+[x if x is not None else torch.zeros(1,dtype=torch.float64) for x in g]))
            except Exception as e:
                res.append(('err',type(e).__name__+': '+str(e)[:100]))
        if res is None: continue
        a,b=res
        desc=(method,amethod,tgrad,rev,is_tuple,npts)
        if a[0]!=b[0]: bad+=1; print('STATUS',desc,a[1] if a[0]=='err' else 'ok',b[1] if b[0]=='err' else 'ok'); continue
        if a[0]=='err': continue
        impl = any(m in ('implicit_midpoint','gl4','radauIIA5','sdirk2','trbdf2') for m in (method,amethod) if m)
        tol = 1e-6 if impl else 1e-9
        for i,(x,y) in enumerate(zip(a[1],b[1])):
            d=float((x-y).abs().max()/(x.abs().max()+1e-12))
            if d>tol: bad+=1; print('VALUE',desc,i,d); break
    print('done',n_cases,'bad',bad)
elif mode == "event":
    rng = random.Random(int(sys.argv[1]))
    n = int(sys.argv[2])
    M=['euler','midpoint','heun3','rk4','explicit_adams','implicit_adams','implicit_euler','trapezoid','gl4','radauIIA5','sdirk2','trbdf2','dopri5','tsit5','bosh3','dopri8']
    bad=0
    for case in range(n):
        method=rng.choice(M); dtype=rng.choice([torch.float32,torch.float64]); rev=rng.random()<0.4
        w0=rng.choice([1.0,2.0,0.7]); y0=torch.tensor([rng.choice([1.0,0.5]), rng.choice([0.0,0.3])],dtype=dtype)
        t0=torch.tensor(rng.choice([0.0,0.5]),dtype=dtype)
        f=(lambda t,y: torch.stack([y[1],-y[0]*w0*w0])) if not rev else (lambda t,y: torch.stack([y[1],-y[0]*w0*w0]))
        thr=rng.choice([0.0,-0.2])
        ev=lambda t,y: y[0]-thr
        opts={} if method in('dopri5','tsit5','bosh3','dopri8') else {'step_size':rng.choice([0.01,0.03]),'interp':rng.choice(['linear','cubic'])}
        res=[]
        for lib in (ref,tda):
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    with torch.no_grad():
                        et,ys=lib.odeint_event(f,y0,t0,event_fn=ev,method=method,options=dict(opts),reverse_time=rev,atol=1e-8,rtol=1e-6)
                res.append(('ok',et,ys))
            except Exception as e:
                res.append(('err',type(e).__name__+str(e)[:80]))
        a,b=res
        desc=(method,str(dtype)[6:],rev,w0,thr,opts)
        if a[0]!=b[0]: bad+=1; print('STATUS',desc,a[1] if a[0]=='err' else 'ok', b[1] if b[0]=='err' else 'ok'); continue
        if a[0]=='err': continue
        impl=method in('implicit_euler','trapezoid','gl4','radauIIA5','sdirk2','trbdf2')
        adaptive=method in('dopri5','tsit5','bosh3','dopri8')
        tol=(1e-5 if dtype==torch.float32 else 1e-9) if (impl or adaptive) else 0.0
        dt=abs(float(a[1])-float(b[1])); dy=float((a[2]-b[2]).abs().max())
        if dt>tol or dy>tol or a[2].dtype!=b[2].dtype or a[1].dtype!=b[1].dtype: bad+=1; print('VALUE',desc,dt,dy,a[1].dtype,b[1].dtype)
    print('done',n,'bad',bad)
elif mode == "complex":
    # r03: complex states (host path; the reference supports them on the CPU): every explicit method + both Adams,
    # tuples, reverse time, adjoint gradients.  Fixed-grid results bit for bit, adaptive 1e-9 (complex128) / 2e-5.
    rng = random.Random(int(sys.argv[1]))
    n = int(sys.argv[2])
    M=['euler','midpoint','heun2','heun3','rk4','explicit_adams','implicit_adams','dopri5','dopri8','tsit5','bosh3','fehlberg2','adaptive_heun']
    bad=0
    for case in range(n):
        method=rng.choice(M); cd=rng.choice([torch.complex64,torch.complex128])
        rd=torch.float32 if cd==torch.complex64 else torch.float64
        shape=rng.choice([(),(1,),(3,),(2,3)]); is_tuple=rng.random()<0.25; rev=rng.random()<0.4
        npts=rng.choice([2,3,6]); adj=rng.random()<0.3 and method in('dopri5','rk4','bosh3','euler')
        g=torch.Generator().manual_seed(rng.randrange(10**6))
        y0=torch.randn(shape,generator=g,dtype=torch.complex128).to(cd)
        y0b=torch.randn(2,generator=g,dtype=torch.float64).to(rd)
        t=torch.sort(torch.rand(npts,generator=g,dtype=torch.float64)).values.to(rd)
        if float((t[1:]-t[:-1]).min())<1e-3: continue
        if rev: t=t.flip(0)
        w=complex(rng.choice([-0.5,0.2]),rng.choice([1.0,3.0]))
        opts={}
        if method in M[:7]:
            if rng.random()<0.5: opts['step_size']=rng.choice([0.05,0.013])
            if rng.random()<0.2: opts['perturb']=True
            if rng.random()<0.3: opts['interp']='cubic'
        else:
            if rng.random()<0.2: opts['first_step']=0.05
            if rng.random()<0.2: opts['max_step']=0.1
        class F(torch.nn.Module):
            def __init__(s):
                super().__init__(); s.w=torch.nn.Parameter(torch.tensor(w,dtype=cd))
            def forward(s,tt,y):
                if is_tuple:
                    a,b=y; return (a*s.w*(1+0.3*tt)-0.1*a*a.abs()**2 + b.sum()*0.05, -b*0.5)
                return y*s.w*(1+0.3*tt)-0.1*y*y.abs()**2
        res=[]
        for lib in (ref,tda):
            f=F(); x=y0.clone().requires_grad_(adj)
            st=(x,y0b) if is_tuple else x
            try:
                if adj:
                    out=lib.odeint_adjoint(f,st,t,method=method,options=dict(opts) or None,rtol=1e-7,atol=1e-9)
                    o=out[0] if is_tuple else out
                    o[-1].abs().pow(2).sum().backward()
                    res.append(('ok',[o.detach(),x.grad,f.w.grad]))
                else:
                    with torch.no_grad(): out=lib.odeint(f,st,t,method=method,options=dict(opts) or None,rtol=1e-7,atol=1e-9)
                    res.append(('ok',list(out) if is_tuple else [out]))
            except Exception as e:
                res.append(('err',type(e).__name__+': '+str(e)[:90]))
        a,b=res; desc=(case,method,str(cd)[6:],shape,is_tuple,rev,npts,adj,opts)
        if a[0]!=b[0]: bad+=1; print('STATUS',desc,a[1] if a[0]=='err' else 'ok',b[1] if b[0]=='err' else 'ok'); continue
        if a[0]=='err':
            if a[1].split(':')[0]!=b[1].split(':')[0]: bad+=1; print('ERRTYPE',desc,a[1],b[1])
            continue
        fixed=method in M[:7]
        tol=(0.0 if fixed and not adj else (2e-5 if cd==torch.complex64 else 1e-9))
        for i,(p,q) in enumerate(zip(a[1],b[1])):
            if p.dtype!=q.dtype or p.shape!=q.shape: bad+=1; print('TYPE',desc,i,p.dtype,q.dtype); break
            fin=torch.isfinite(p.abs())&torch.isfinite(q.abs())
            if bool((torch.isfinite(p.abs())!=torch.isfinite(q.abs())).any()): bad+=1; print('NONFINITE',desc,i); break
            d=float((p-q)[fin].abs().max()/(p[fin].abs().max()+1e-30)) if fin.any() else 0.0
            if not d<=tol: bad+=1; print('VALUE',desc,i,d); break
    print('done',n,'bad',bad)
elif mode == "tableau":
    # r03: RANDOM explicit embedded Runge-Kutta tableaus (2..9 stages, random structural zeros, FSAL or not, error row
    # leading with the solution row or not) on this package's native adaptive solver against the reference's
    # RKAdaptiveStepsizeODESolver given the same table — the kernels, the end-of-step fusion and the dense output are
    # generic in the table.  Consistency of the method is irrelevant here (both libraries run the same arithmetic);
    # evaluation counts must be equal and solutions agree to 1e-10.
    from torchdiffeq._impl import rk_common as ref_rk
    from torchdiffeq._impl.odeint import SOLVERS as REF_SOLVERS
    from torchdiffeq_amd.solvers import RKAdaptiveStepsizeODESolver
    from torchdiffeq_amd.tableaus import Tableau
    rng = random.Random(int(sys.argv[1]))
    n = int(sys.argv[2])
    bad = 0
    for case in range(n):
        S = rng.randint(2, 9)
        fsal = rng.random() < 0.5
        def rnd(): return 0.0 if rng.random() < 0.25 else rng.uniform(-0.6, 0.9)
        beta = [[rnd() for _ in range(i + 1)] for i in range(S)]
        for r in beta:                      # rows sum to their abscissa, like a real method (keeps the problem tame)
            if all(v == 0.0 for v in r): r[0] = 0.3
        alpha = [min(1.0, abs(sum(r))) for r in beta]
        if fsal:
            alpha[-1] = 1.0
            tot = sum(beta[-1]); beta[-1] = [v / tot for v in beta[-1]] if abs(tot) > 1e-3 else [1.0 / S] * S
            c_sol = list(beta[-1]) + [0.0]
        else:
            w = [abs(rnd()) + 0.05 for _ in range(S + 1)]
            c_sol = [v / sum(w) for v in w]
        c_err = [rng.uniform(-1, 1) * 1e-2 * (rng.random() < 0.8) for _ in range(S + 1)]
        if rng.random() < 0.3:
            c_err = [c * 1e-2 for c in c_sol[:-1]] + [rng.uniform(-1, 1) * 1e-3]          # leads with the solution row
        c_err[0] -= sum(c_err)               # like a real pair: the two solutions agree to first order
        mid = [0.5 * c for c in c_sol]; mid[0] += 0.125; mid[-1] -= 0.125
        order = rng.choice([2, 3, 5, 8])
        f64 = lambda v: torch.tensor(v, dtype=torch.float64)
        name = f"fuzztab{case}"
        RefCls = type("RefFuzz", (ref_rk.RKAdaptiveStepsizeODESolver,), dict(order=order, tableau=ref_rk._ButcherTableau(
            alpha=f64(alpha), beta=[f64(r) for r in beta], c_sol=f64(c_sol), c_error=f64(c_err)), mid=f64(mid)))
        Cls = type("Fuzz", (RKAdaptiveStepsizeODESolver,), dict(order=order, tableau=Tableau(
            name, order, tuple(alpha), tuple(tuple(r) for r in beta), tuple(c_sol), tuple(c_err), tuple(mid))))
        REF_SOLVERS[name] = RefCls; tda.SOLVERS[name] = Cls
        dtype = rng.choice([torch.float32, torch.float64])
        shape = rng.choice([(3,), (2, 3), (17,)])
        is_tuple = rng.random() < 0.3
        g = torch.Generator().manual_seed(rng.randrange(10 ** 6))
        y0 = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype)
        y0b = torch.randn(4, generator=g, dtype=torch.float64).to(dtype)
        t = torch.sort(torch.rand(rng.choice([2, 3, 6]), generator=g, dtype=torch.float64)).values.to(dtype)
        if float((t[1:] - t[:-1]).min()) < 1e-3: continue
        if rng.random() < 0.4: t = t.flip(0)
        kw = dict(rtol=rng.choice([1e-3, 1e-5]), atol=rng.choice([1e-4, 1e-7]))
        opts = {"max_num_steps": 3000}
        if rng.random() < 0.3: opts["first_step"] = 0.02
        res = []
        for lib in (ref, tda):
            nfe = [0]
            def fn(t_, y):
                nfe[0] += 1
                if is_tuple: return (-y[0] * (1 + 0.3 * t_) + 0.1 * torch.sin(y[0]) + y[1].sum() * 0.01, -0.5 * y[1])
                return -y * (1 + 0.3 * t_) + 0.1 * torch.sin(y)
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    with torch.no_grad():
                        out = lib.odeint(fn, (y0, y0b) if is_tuple else y0, t, method=name, options=dict(opts), **kw)
                res.append(("ok", out[0] if is_tuple else out, nfe[0]))
            except Exception as e:
                res.append(("err", type(e).__name__ + ": " + str(e)[:80], None))
        del REF_SOLVERS[name], tda.SOLVERS[name]
        a, b = res
        desc = (case, S, fsal, order, str(dtype)[6:], shape, is_tuple, kw, opts)
        if a[0] != b[0]: bad += 1; print("STATUS", desc, a[1] if a[0] == "err" else "ok", b[1] if b[0] == "err" else "ok"); continue
        if a[0] == "err": continue
        fin = torch.isfinite(a[1]) & torch.isfinite(b[1])
        d = float(((a[1] - b[1])[fin]).abs().max() / (a[1][fin].abs().max() + 1e-30)) if fin.any() else 0.0
        tol = 1e-4 if dtype == torch.float32 else 1e-10
        if bool((torch.isfinite(a[1]) != torch.isfinite(b[1])).any()) or d > tol or (a[2] != b[2] and dtype == torch.float64):
            bad += 1; print("VALUE", desc, d, a[2], b[2])
    print("done", n, "bad", bad)
elif mode == "eventgrad":
    # r03: gradients THROUGH odeint_event (event time and final state wrt y0, the START time t0, a parameter, a second
    # tuple component), adaptive and fixed-grid methods, reverse_time, odeint_adjoint as the interface.  Found: the
    # fixed-grid event solve dropped d/d t0.
    rng = random.Random(int(sys.argv[1]))
    n = int(sys.argv[2])
    bad = 0
    M=['dopri5','bosh3','tsit5','dopri8','rk4','euler','midpoint','implicit_adams']
    for case in range(n):
        method=rng.choice(M); rev=rng.random()<0.35; is_tuple=rng.random()<0.4; adj=rng.random()<0.5
        w0=rng.choice([1.0,2.0,0.7]); thr=rng.choice([0.0,-0.2,0.3])
        opts={} if method in M[:4] else {'step_size':rng.choice([0.01,0.03]),'interp':rng.choice(['linear','cubic'])}
        res=[]
        for L in (ref, tda):
            k=torch.tensor(w0*w0,dtype=torch.float64,requires_grad=True)
            y0=torch.tensor([rng.choice([1.0]) ,0.1],dtype=torch.float64,requires_grad=True)
            aux=torch.tensor([0.5],dtype=torch.float64,requires_grad=True)
            t0=torch.tensor(0.3,dtype=torch.float64,requires_grad=True)
            class F(torch.nn.Module):
                def __init__(s): super().__init__(); s.k=torch.nn.Parameter(k.detach().clone())
                def forward(s,t,st):
                    if is_tuple:
                        y,a=st; return (torch.stack([y[1],-y[0]*s.k]), -a*0.3)
                    return torch.stack([st[1],-st[0]*s.k])
            f=F()
            ev=(lambda t,st: st[0][0]-thr+0.0*st[1].sum()) if is_tuple else (lambda t,st: st[0]-thr)
            state=(y0,aux) if is_tuple else y0
            try:
                kw=dict(odeint_interface=L.odeint_adjoint) if adj else {}
                et,ys=L.odeint_event(f,state,t0,event_fn=ev,method=method,options=dict(opts),reverse_time=rev,atol=1e-9,rtol=1e-7,**kw)
                yl=ys[0][-1] if is_tuple else ys[-1]
                loss=et*2.0+(yl**2).sum()+(ys[1][-1].sum() if is_tuple else 0)
                g=torch.autograd.grad(loss,[y0,t0,f.k]+([aux] if is_tuple else []),allow_unused=True)
                res.append(('ok',[et.detach().reshape(1),yl.detach()]+[x if x is not None else torch.zeros(1,dtype=torch.float64) for x in g]))
            except Exception as e:
                res.append(('err',type(e).__name__+': '+str(e)[:90]))
        a,b=res; desc=(case,method,rev,is_tuple,adj,w0,thr,opts)
        if a[0]!=b[0]: bad+=1; print('STATUS',desc,a[1] if a[0]=='err' else 'ok',b[1] if b[0]=='err' else 'ok'); continue
        if a[0]=='err':
            if a[1].split(':')[0]!=b[1].split(':')[0]: bad+=1; print('ERRTYPE',desc,a[1],b[1])
            continue
        tol=1e-6 if method in M[:4] else 1e-9
        if adj: tol=max(tol,1e-5)
        for i,(p,q) in enumerate(zip(a[1],b[1])):
            d=float((p-q).abs().max()/(p.abs().max()+1e-12))
            if not d<=tol: bad+=1; print('VALUE',desc,i,d); break
    print('done',n,'bad',bad)
elif mode == "callbacks":
    # r03: the (t0, dt) sequences every callback sees — callback_step / accept / reject and their *_adjoint twins — for
    # all explicit and Adams methods, tuple states, both directions, step options; forward 1e-9, backward solve 1e-6.
    rng = random.Random(int(sys.argv[1]))
    n = int(sys.argv[2])
    bad = 0
    AD=['dopri5','bosh3','tsit5','fehlberg2','adaptive_heun','dopri8']; FX=['euler','midpoint','heun2','heun3','rk4','explicit_adams','implicit_adams']
    for case in range(n):
        method=rng.choice(AD+FX); rev=rng.random()<0.4; is_tuple=rng.random()<0.3; adj=rng.random()<0.4
        g=torch.Generator().manual_seed(rng.randrange(10**6))
        A=torch.randn(3,3,generator=g,dtype=torch.float64)*0.7-0.3*torch.eye(3,dtype=torch.float64)
        y0=torch.randn(4,3,generator=g,dtype=torch.float64); yb=torch.randn(2,generator=g,dtype=torch.float64)
        t=torch.sort(torch.rand(rng.choice([2,3,5]),generator=g,dtype=torch.float64)*2).values
        if float((t[1:]-t[:-1]).min())<1e-2: continue
        if rev: t=t.flip(0)
        opts={}
        if method in AD:
            if rng.random()<0.3: opts['first_step']=0.05
            if rng.random()<0.2: opts['max_step']=0.2
            if rng.random()<0.2: opts['min_step']=0.01
        else:
            if rng.random()<0.5: opts['step_size']=rng.choice([0.05,0.13])
            if rng.random()<0.2: opts['perturb']=True
        res=[]
        for L in (ref,tda):
            rec={k:[] for k in ('s','a','r','sa','aa','ra')}
            class F(torch.nn.Module):
                def __init__(s): super().__init__(); s.A=torch.nn.Parameter(A.clone())
                def forward(s,tt,y):
                    if is_tuple: return (torch.tanh(y[0]@s.A)*torch.cos(tt), -0.5*y[1])
                    return torch.tanh(y@s.A)*torch.cos(tt)
                def callback_step(s,t0,y,dt): rec['s'].append((float(t0),float(dt)))
                def callback_accept_step(s,t0,y,dt): rec['a'].append((float(t0),float(dt)))
                def callback_reject_step(s,t0,y,dt): rec['r'].append((float(t0),float(dt)))
                def callback_step_adjoint(s,t0,y,dt): rec['sa'].append((float(t0),float(dt)))
                def callback_accept_step_adjoint(s,t0,y,dt): rec['aa'].append((float(t0),float(dt)))
                def callback_reject_step_adjoint(s,t0,y,dt): rec['ra'].append((float(t0),float(dt)))
            f=F(); x=y0.clone().requires_grad_(adj)
            st=(x,yb) if is_tuple else x
            try:
                if adj:
                    out=L.odeint_adjoint(f,st,t,method=method,options=dict(opts) or None,rtol=1e-6,atol=1e-8)
                    (out[0] if is_tuple else out)[-1].pow(2).sum().backward()
                else:
                    with torch.no_grad(): out=L.odeint(f,st,t,method=method,options=dict(opts) or None,rtol=1e-6,atol=1e-8)
                res.append(('ok',rec))
            except Exception as e:
                res.append(('err',type(e).__name__+': '+str(e)[:80]))
        a,b=res; desc=(case,method,rev,is_tuple,adj,opts,len(t))
        if a[0]!=b[0]: bad+=1; print('STATUS',desc,a[1] if a[0]=='err' else 'ok',b[1] if b[0]=='err' else 'ok'); continue
        if a[0]=='err': continue
        for k in a[1]:
            p,q=a[1][k],b[1][k]
            if len(p)!=len(q): bad+=1; print('COUNT',desc,k,len(p),len(q)); break
            d=max([abs(x[0]-y[0])+abs(x[1]-y[1]) for x,y in zip(p,q)],default=0.0)
            tol=1e-9 if k in ('s','a','r') else 1e-6
            if d>tol: bad+=1; print('SEQ',desc,k,d); break
    print('done',n,'bad',bad)
elif mode == "hessian":
    # r03: SECOND-order gradients through plain odeint (create_graph=True) wrt y0, a weight matrix and the output times,
    # explicit RK fixed-grid (linear / cubic interpolation), adaptive and Adams methods, tuple states, both directions.
    # Found: the cubic Hermite interpolation node lacked the curvature of its basis.  1e-8 (dopri8 1e-6: noise-level
    # first step); low-order adaptive methods at tight tolerances accumulate 1e-7 in d2/dt2 over thousands of steps.
    rng = random.Random(int(sys.argv[1]))
    n = int(sys.argv[2])
    bad = 0
    M=['euler','midpoint','heun2','heun3','rk4','dopri5','bosh3','tsit5','fehlberg2','adaptive_heun','explicit_adams','implicit_adams','dopri8']
    for case in range(n):
        method=rng.choice(M); rev=rng.random()<0.3; tgrad=rng.random()<0.4; is_tuple=rng.random()<0.25
        g=torch.Generator().manual_seed(rng.randrange(10**6))
        W0=torch.randn(3,3,generator=g,dtype=torch.float64)*0.5; x0=torch.randn(2,3,generator=g,dtype=torch.float64); b0=torch.randn(2,generator=g,dtype=torch.float64)
        tt=torch.sort(torch.rand(3,generator=g,dtype=torch.float64)).values
        if float((tt[1:]-tt[:-1]).min())<0.05: continue
        if rev: tt=tt.flip(0)
        opts={}
        if method in M[:5]+M[10:12] and rng.random()<0.6: opts['step_size']=rng.choice([0.1,0.07])
        if method in M[:5] and rng.random()<0.3: opts['interp']='cubic'
        res=[]
        for L in (ref,tda):
            W=W0.clone().requires_grad_(True); x=x0.clone().requires_grad_(True); t=tt.clone().requires_grad_(tgrad)
            def f(t_,y):
                if is_tuple: return (torch.tanh(y[0]@W.T)*torch.cos(t_), -y[1]*y[1]*0.3)
                return torch.tanh(y@W.T)*torch.cos(t_)
            try:
                out=L.odeint(f,(x,b0) if is_tuple else x,t,method=method,options=dict(opts) or None,rtol=1e-8,atol=1e-10)
                o=out[0] if is_tuple else out
                loss=(o[-1]**2).sum()+(o[1]**3).sum()
                ins=[x,W]+([t] if tgrad else [])
                g1=torch.autograd.grad(loss,ins,create_graph=True)
                second=sum((gi**2).sum() for gi in g1)
                g2=torch.autograd.grad(second,ins,allow_unused=True)
                res.append(('ok',[gi.detach() for gi in g1]+[gi if gi is not None else torch.zeros(1,dtype=torch.float64) for gi in g2]))
            except Exception as e:
                res.append(('err',type(e).__name__+': '+str(e)[:100]))
        a,b=res; desc=(case,method,rev,tgrad,is_tuple,opts)
        if a[0]!=b[0]: bad+=1; print('STATUS',desc,a[1] if a[0]=='err' else 'ok',b[1] if b[0]=='err' else 'ok'); continue
        if a[0]=='err': continue
        for i,(p,q) in enumerate(zip(a[1],b[1])):
            d=float((p-q).abs().max()/(p.abs().max()+1e-12))
            if not d<=(1e-6 if method=='dopri8' else 1e-8): bad+=1; print('VALUE',desc,i,d); break
    print('done',n,'bad',bad)
elif mode == "vectol":
    # r03: tolerances given PER ELEMENT (tensors / lists broadcasting against the state, vector entries of tuple
    # tolerances), all adaptive methods, fp32 / fp64, tuple states, odeint_adjoint's forward solve.  Evaluation counts
    # equal, solutions / gradients 1e-9 (the reference rejects a multi-dimensional vector ENTRY of a tuple tolerance,
    # which is accepted here: reported as STATUS, not a parity defect).
    rng = random.Random(int(sys.argv[1]))
    n = int(sys.argv[2])
    bad = 0
    AD=['dopri5','bosh3','tsit5','fehlberg2','adaptive_heun','dopri8']
    for case in range(n):
        method=rng.choice(AD); dtype=rng.choice([torch.float64,torch.float64,torch.float32]); rev=rng.random()<0.3
        shape=rng.choice([(3,),(2,3),(4,1,3)]); is_tuple=rng.random()<0.3; adj=rng.random()<0.3 and dtype==torch.float64
        g=torch.Generator().manual_seed(rng.randrange(10**6))
        y0=torch.randn(shape,generator=g,dtype=torch.float64).to(dtype); yb=torch.randn(2,generator=g,dtype=torch.float64).to(dtype)
        t=torch.sort(torch.rand(rng.choice([2,3]),generator=g,dtype=torch.float64)).values.to(dtype)
        if float((t[1:]-t[:-1]).min())<0.05: continue
        if rev: t=t.flip(0)
        def vec(lo,hi):
            e=[10**rng.uniform(lo,hi) for _ in range(3)]
            r=rng.random()
            if r<0.4: return torch.tensor(e,dtype=rng.choice([torch.float64,torch.float32]))
            if r<0.6: return e
            if r<0.8 and len(shape)>1: return torch.tensor(e,dtype=torch.float64).expand(shape).clone()
            return 10**rng.uniform(lo,hi)
        rtol=vec(-7,-3); atol=vec(-9,-5)
        if is_tuple:
            rtol=(rtol, 10**rng.uniform(-6,-3)); atol=(atol, torch.tensor([1e-7,1e-8]) if rng.random()<0.5 else 1e-8)
        res=[]
        for L in (ref,tda):
            nfe=[0]
            w=torch.tensor([1.0,3.0,0.3],dtype=dtype,requires_grad=adj)
            def f(tt,y):
                nfe[0]+=1
                if is_tuple: return (-y[0]*w*(1+0.2*tt)+0.1*torch.sin(y[0]), -0.4*y[1])
                return -y*w*(1+0.2*tt)+0.1*torch.sin(y)
            x=y0.clone().requires_grad_(adj)
            try:
                if adj:
                    out=L.odeint_adjoint(f,(x,yb) if is_tuple else x,t,method=method,rtol=rtol,atol=atol,adjoint_rtol=1e-7,adjoint_atol=1e-9,adjoint_params=(w,))
                    o=out[0] if is_tuple else out; o[-1].pow(2).sum().backward()
                    res.append(('ok',[o.detach(),x.grad,w.grad],nfe[0]))
                else:
                    with torch.no_grad(): out=L.odeint(f,(x,yb) if is_tuple else x,t,method=method,rtol=rtol,atol=atol)
                    res.append(('ok',[out[0] if is_tuple else out],nfe[0]))
            except Exception as e:
                res.append(('err',type(e).__name__+': '+str(e)[:90],0))
        a,b=res; desc=(case,method,str(dtype)[6:],shape,is_tuple,adj,rev,type(rtol).__name__,type(atol).__name__)
        if a[0]!=b[0]: bad+=1; print('STATUS',desc,a[1] if a[0]=='err' else 'ok',b[1] if b[0]=='err' else 'ok'); continue
        if a[0]=='err': continue
        tol=(1e-9 if method!='dopri8' else 1e-6) if dtype==torch.float64 else 2e-5
        if a[2]!=b[2] and dtype==torch.float64 and method!='dopri8': bad+=1; print('NFE',desc,a[2],b[2]); continue
        exact = os.environ.get("TDEQ_FUZZ_BACKEND") == "host" and not adj    # r04: the host path is the reference's arithmetic
        for i,(p,q) in enumerate(zip(a[1],b[1])):
            d=float((p-q).abs().max()/(p.abs().max()+1e-30))
            if not d<=tol or (exact and (d != 0.0 or a[2] != b[2])): bad+=1; print('VALUE' if not d<=tol else 'BITS',desc,i,d,a[2],b[2]); break
    print('done',n,'bad',bad)
elif mode == "brow":
    # r04: SURVEY.md §8(b) corners — the solver option `dtype`, states below fp32, func outputs of the wrong shape.  Runs on
    # the package's torch-op host path (forced): bf16 / fp16 solves must equal the reference bit for bit with equal
    # evaluation counts (or fail with the same assertion), fp32 / fp64 ones under a `dtype` option within the solve's
    # own tolerance with equal counts in fp64.
    warnings.simplefilter("ignore")
    import importlib
    importlib.reload(_native)           # undo the oracle substitution made above: states select their own backend
    seed, n = int(sys.argv[1]), int(sys.argv[2]); rng = random.Random(seed); bad = 0
    DT = [torch.bfloat16, torch.float16, torch.float32, torch.float64]
    for case in range(n):
        method = rng.choice(['dopri5', 'dopri8', 'bosh3', 'tsit5', 'adaptive_heun', 'fehlberg2', 'rk4', 'euler', 'midpoint'])
        sdtype = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16, torch.float32, torch.float64])
        shape = rng.choice([(3,), (2, 3), (4, 1, 2), (1,)]); is_tuple = rng.random() < 0.25; rev = rng.random() < 0.4
        g = torch.Generator().manual_seed(rng.randrange(10**6))
        y0 = torch.randn(shape, generator=g, dtype=torch.float64).to(sdtype); yb = torch.rand(2, generator=g, dtype=torch.float64).to(sdtype)
        npts = rng.choice([2, 3, 5]); t = torch.sort(torch.rand(npts, generator=g, dtype=torch.float64) * 2).values
        if float((t[1:] - t[:-1]).min()) < 0.05: continue
        t = t.to(rng.choice([torch.float32, torch.float64]))
        if rev: t = t.flip(0)
        low = sdtype in (torch.bfloat16, torch.float16)
        rtol, atol = (rng.choice([1e-2, 3e-2]), 1e-3) if low else (rng.choice([1e-4, 1e-6]), 1e-7)
        opts = {}
        fixed = method in ('rk4', 'euler', 'midpoint')
        if not fixed:
            if rng.random() < 0.6: opts['dtype'] = rng.choice(DT)
            if rng.random() < 0.2: opts['first_step'] = 0.01
            if rng.random() < 0.15: opts['max_step'] = 0.2
            if rng.random() < 0.15: opts['safety'] = 0.8
            if rng.random() < 0.15 and not rev: opts['step_t'] = torch.tensor([float(t.min()) + 0.013])
        elif rng.random() < 0.4: opts['step_size'] = 0.0625
        res = []
        for L in (ref, tda):
            nfe = [0]
            def f(tt, y):
                nfe[0] += 1
                if is_tuple: return (-y[0] * (1 + 0.2 * tt) + 0.1 * torch.sin(y[0]), -0.4 * y[1])
                return -y * (1 + 0.2 * tt) + 0.1 * torch.sin(y)
            try:
                with torch.no_grad(): out = L.odeint(f, (y0, yb) if is_tuple else y0, t, method=method, rtol=rtol, atol=atol, options=dict(opts))
                res.append(('ok', out[0] if is_tuple else out, nfe[0]))
            except Exception as e:
                res.append(('err', type(e).__name__ + ': ' + str(e)[:60], 0))
        a, b = res; desc = (case, method, str(sdtype)[6:], shape, is_tuple, rev, {k: (str(v)[6:] if k == 'dtype' else v) for k, v in opts.items()})
        if a[0] != b[0] or (a[0] == 'err' and a[1] != b[1]): bad += 1; print('STATUS', desc, a[1] if a[0] == 'err' else 'ok', '|', b[1] if b[0] == 'err' else 'ok'); continue
        if a[0] == 'err': continue
        if low:
            if a[2] != b[2] or not torch.equal(a[1], b[1]): bad += 1; print('LOWBITS', desc, a[2], b[2], float((a[1].float() - b[1].float()).abs().max()))
            continue
        if a[2] != b[2] and sdtype == torch.float64 and method != 'dopri8': bad += 1; print('NFE', desc, a[2], b[2]); continue
        d = float((a[1] - b[1]).abs().max() / (a[1].abs().max() + 1e-30))
        if not d <= (50 * rtol if sdtype == torch.float32 else (1e-6 if method == 'dopri8' else 1e-9)): bad += 1; print('VALUE', desc, d)
    print('done', n, 'bad', bad)
elif mode == "hostexact":
    # r04: the torch-op host path hands every row sum and norm to ATen exactly as the reference does (_fallback.py), so on
    # the CPU it must reproduce the reference BIT FOR BIT — every explicit method, fp32 / fp64 / complex / bf16, tuple
    # states, both directions, the solver options, odeint_adjoint and backprop gradients — with equal evaluation counts.
    warnings.simplefilter("ignore")
    import importlib
    importlib.reload(_native)
    seed, n = int(sys.argv[1]), int(sys.argv[2]); rng = random.Random(seed); bad = 0
    ADAPT = ['dopri5', 'dopri8', 'bosh3', 'tsit5', 'adaptive_heun', 'fehlberg2']; FIX = ['rk4', 'euler', 'midpoint', 'heun2', 'heun3']
    for case in range(n):
        method = rng.choice(ADAPT + ADAPT + FIX)
        dtype = rng.choice([torch.float32, torch.float64, torch.float64, torch.complex64, torch.complex128, torch.bfloat16])
        rdt = {torch.complex64: torch.float32, torch.complex128: torch.float64}.get(dtype, dtype)
        shape = rng.choice([(3,), (2, 3), (4, 1, 2), (1,), (17,)]); is_tuple = rng.random() < 0.3; rev = rng.random() < 0.4
        grad = rng.choice([None, None, 'adjoint', 'backprop']) if dtype in (torch.float32, torch.float64) else None
        g = torch.Generator().manual_seed(rng.randrange(10**6))
        mk = lambda s: (torch.complex(torch.randn(s, generator=g, dtype=torch.float64), torch.randn(s, generator=g, dtype=torch.float64))
                        if dtype.is_complex else torch.randn(s, generator=g, dtype=torch.float64)).to(dtype)
        y0, yb, w0 = mk(shape), torch.rand(2, generator=g, dtype=torch.float64).to(rdt), mk(shape) * 0.3
        npts = rng.choice([2, 3, 5]); t = torch.sort(torch.rand(npts, generator=g, dtype=torch.float64) * 2).values
        if float((t[1:] - t[:-1]).min()) < 0.05: continue
        t = t.to(rng.choice([torch.float32, torch.float64]) if rdt != torch.float64 else torch.float64)
        if rev: t = t.flip(0)
        low = dtype == torch.bfloat16
        rtol, atol = (2e-2, 1e-3) if low else ((1e-4, 1e-6) if method in ('adaptive_heun', 'fehlberg2') or rdt == torch.float32 else (1e-7, 1e-9))
        opts = {}
        if method in ADAPT:
            r = rng.random()
            if r < 0.15: opts['first_step'] = 0.01
            elif r < 0.3: opts['max_step'] = 0.2
            elif r < 0.4: opts['safety'] = 0.8
            elif r < 0.5 and not rev: opts['step_t'] = torch.tensor([float(t.min()) + 0.013])
            elif r < 0.6 and not rev: opts['jump_t'] = torch.tensor([float(t.min()) + 0.021])
            elif r < 0.66 and not rev: opts['step_t'] = torch.tensor([float(t.min()) + 0.4 * float(t.max() - t.min())])
            elif r < 0.72 and not low: opts['min_step'] = 0.3
            if rng.random() < 0.25: opts['dtype'] = rng.choice([torch.float32, torch.float64])
        else:
            if rng.random() < 0.5: opts['step_size'] = 0.0625
            if rng.random() < 0.3: opts['interp'] = 'cubic'
            if rng.random() < 0.3: opts['perturb'] = True
        t_grad = rng.random() < 0.5
        adj_kw = {}
        if grad == 'adjoint':
            r = rng.random()
            if r < 0.2: adj_kw['adjoint_options'] = dict(norm='seminorm')
            elif r < 0.35: adj_kw.update(adjoint_rtol=rtol * 10, adjoint_atol=atol * 10)
            elif r < 0.5 and method in ADAPT: adj_kw.update(adjoint_method=rng.choice(['bosh3', 'dopri5', 'rk4']), adjoint_options=dict())
            elif r < 0.6 and is_tuple: adj_kw.update(adjoint_rtol=(rtol, rtol * 3, rtol, rtol * 10), adjoint_atol=atol)
            if adj_kw.get('adjoint_method') == 'rk4': adj_kw['adjoint_options'] = dict(step_size=0.05)
        res = []
        for L in (ref, tda):
            nfe = [0]
            w = w0.clone().requires_grad_(grad is not None)
            def f(tt, y):
                nfe[0] += 1
                if is_tuple: return (-y[0] * w * (1 + 0.2 * tt) + 0.1 * torch.sin(y[0]), -0.4 * y[1] * (1 + y[0].abs().mean().to(y[1].dtype)))
                return -y * w * (1 + 0.2 * tt) + 0.1 * torch.sin(y)
            x = y0.clone().requires_grad_(grad is not None)
            tt = t.clone().requires_grad_(True) if (grad and t_grad) else t       # the output times in the graph as well
            try:
                if grad == 'adjoint':
                    out = L.odeint_adjoint(f, (x, yb) if is_tuple else x, tt, method=method, rtol=rtol, atol=atol, options=dict(opts), adjoint_params=(w,), **adj_kw)
                elif grad == 'backprop':
                    out = L.odeint(f, (x, yb) if is_tuple else x, tt, method=method, rtol=rtol, atol=atol, options=dict(opts))
                else:
                    with torch.no_grad(): out = L.odeint(f, (x, yb) if is_tuple else x, t, method=method, rtol=rtol, atol=atol, options=dict(opts))
                o = out[0] if is_tuple else out
                vals = [o.detach()] + ([out[1].detach()] if is_tuple else [])
                if grad:
                    (o[-1].pow(2).sum() + o[len(t) // 2].sum()).backward(); vals += [x.grad, w.grad] + ([tt.grad] if t_grad else [])
                res.append(('ok', vals, nfe[0]))
            except Exception as e:
                res.append(('err', type(e).__name__ + ': ' + str(e)[:60], 0))
        a, b = res; desc = (case, method, str(dtype)[6:], shape, is_tuple, rev, grad, sorted(adj_kw), {k: (str(v)[6:] if k == 'dtype' else ('t' if torch.is_tensor(v) else v)) for k, v in opts.items()})
        if a[0] != b[0] or (a[0] == 'err' and a[1] != b[1]): bad += 1; print('STATUS', desc, a[1] if a[0] == 'err' else 'ok', '|', b[1] if b[0] == 'err' else 'ok'); continue
        if a[0] == 'err': continue
        n_fwd = 2 if is_tuple else 1            # backprop gradients come from a hand-written backward (autodiff._LinearOp): same
        cmp = a[1] if grad != 'backprop' else a[1][:n_fwd]      # values to rounding, a different accumulation order of the cotangents
        real = lambda x: torch.view_as_real(x) if x.is_complex() else x
        same_bits = lambda p, q: torch.equal(real(p).isnan(), real(q).isnan()) and torch.equal(real(p).nan_to_num(), real(q).nan_to_num())   # (a NaN both sides have at the same place is the same result)
        exact = all(same_bits(p, q) for p, q in zip(cmp, b[1]))
        if exact and grad == 'backprop':
            exact = all(float((p - q).abs().max()) <= ((2e-3 if (t_grad and j == len(a[1]) - n_fwd - 1) else 1e-4) if rdt == torch.float32 else 1e-12) * float(p.abs().max() + 1e-30) for j, (p, q) in enumerate(zip(a[1][n_fwd:], b[1][n_fwd:])))       # (fp32: the time gradient is a cancelling dot product over the state — both libraries are ~1e-4 from the fp64 value)
        if a[2] != b[2] or not exact:
            bad += 1; print('BITS', desc, a[2], b[2], [float((p - q).abs().max() / (p.abs().max() + 1e-30)) for p, q in zip(a[1], b[1])])
            if os.environ.get('FUZZ_DUMP'): torch.save((a[1], b[1]), os.environ['FUZZ_DUMP'])      # the two result lists of the last mismatch
    print('done', n, 'bad', bad)
else:
    raise SystemExit("mode must be fixed | adaptive | adjoint | backprop | event | complex | tableau | eventgrad | callbacks | hessian | vectol | brow | hostexact")
