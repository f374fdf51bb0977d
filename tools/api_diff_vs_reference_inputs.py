"""Differential batch against the imported reference (build container; see api_diff_vs_reference.py): prints SAME / DIFF
per case — exception class, shapes, dtypes, values, gradients."""
import sys, torch, warnings, math
sys.path.insert(0,'/root/reference'); import torchdiffeq as ref
sys.path.insert(0,'/root/repo'); import torchdiffeq_amd as tda
torch.manual_seed(1)
A=torch.randn(3,3)*0.5
class F(torch.nn.Module):
    def __init__(s,dtype=torch.float32):
        super().__init__(); torch.manual_seed(0); s.lin=torch.nn.Linear(3,3).to(dtype)
    def forward(s,t,y): return torch.tanh(s.lin(y))*torch.cos(t)
base=torch.randn(6,3)
t=torch.linspace(0,1,4)
def run(lib,call):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        try:
            r=call(lib)
            return ('ok',r,[])
        except Exception as e:
            return (type(e).__name__, str(e)[:100],[])
def g(L, y0fn, method='dopri5', adjoint=False, tt=None, dtype=torch.float32, **kw):
    f=F(dtype)
    x=y0fn()
    fn=L.odeint_adjoint if adjoint else L.odeint
    out=fn(f,x,t if tt is None else tt,method=method,**kw)
    res=[out.detach()]
    if x.requires_grad or adjoint:
        out[-1].pow(2).sum().backward()
        res += [p.grad for p in f.parameters()]
    return tuple(res)
cases={
 'noncontig_T': lambda L: g(L, lambda: torch.randn(3,6,generator=torch.Generator().manual_seed(2)).T),
 'noncontig_slice': lambda L: g(L, lambda: torch.randn(12,3,generator=torch.Generator().manual_seed(2))[::2]),
 'expanded': lambda L: g(L, lambda: torch.randn(1,3,generator=torch.Generator().manual_seed(2)).expand(6,3)),
 'nonleaf_grad': lambda L: g(L, lambda: (torch.randn(6,3,generator=torch.Generator().manual_seed(2),requires_grad=True)*2)),
 'nonleaf_grad_adj': lambda L: g(L, lambda: (torch.randn(6,3,generator=torch.Generator().manual_seed(2),requires_grad=True)*2), adjoint=True),
 't_many': lambda L: g(L, lambda: base.clone(), tt=torch.linspace(0,1,301)),
 't_dec': lambda L: g(L, lambda: base.clone(), tt=torch.linspace(1,0,7)),
 't_dec_adj': lambda L: g(L, lambda: base.clone(), tt=torch.linspace(1,-1,7), adjoint=True),
 't_f64_y_f32_dopri5': lambda L: g(L, lambda: base.clone(), tt=torch.linspace(0,1,4,dtype=torch.float64)),
 't_f64_y_f32_rk4': lambda L: g(L, lambda: base.clone(), tt=torch.linspace(0,1,4,dtype=torch.float64), method='rk4'),
 't_f64_y_f32_adj': lambda L: g(L, lambda: base.clone(), tt=torch.linspace(0,1,4,dtype=torch.float64), adjoint=True),
 't_f32_y_f64': lambda L: g(L, lambda: base.double(), dtype=torch.float64),
 't_f32_y_f64_adj': lambda L: g(L, lambda: base.double(), dtype=torch.float64, adjoint=True),
 'dtype_opt_f32_t64': lambda L: g(L, lambda: base.clone(), tt=torch.linspace(0,1,4,dtype=torch.float64), options=dict(dtype=torch.float32)),
 'bf16': lambda L: g(L, lambda: base.bfloat16(), dtype=torch.bfloat16, rtol=1e-2, atol=1e-2),
 'bf16_rk4': lambda L: g(L, lambda: base.bfloat16(), dtype=torch.bfloat16, method='rk4', options=dict(step_size=0.25)),
 'f16_rk4': lambda L: g(L, lambda: base.half(), dtype=torch.float16, method='rk4', options=dict(step_size=0.25)),
 'rtol_tuple_plain': lambda L: tuple(o for o in L.odeint(lambda t,y:(-y[0],y[1]*0.3),(base,base[:2]),t,rtol=(1e-3,1e-6),atol=(1e-4,1e-8))),
 'rtol_tuple_wrong_len': lambda L: tuple(o for o in L.odeint(lambda t,y:(-y[0],y[1]*0.3),(base,base[:2]),t,rtol=(1e-3,),atol=1e-8)),
 'rtol_tensor_per_comp': lambda L: tuple(o for o in L.odeint(lambda t,y:(-y[0],y[1]*0.3),(base,base[:2]),t,rtol=(torch.full((6,3),1e-3),1e-6),atol=1e-8)),
 'complex': lambda L: (L.odeint(lambda t,y: 1j*y, base.to(torch.complex64), t),),
 'complex_rk4': lambda L: (L.odeint(lambda t,y: 1j*y, base.to(torch.complex64), t, method='rk4'),),
 'complex_adj': lambda L: g(L, lambda: base.to(torch.complex64), dtype=torch.complex64, adjoint=True),
 'complex_t': lambda L: (L.odeint(lambda t,y: -y, base, t.to(torch.complex64)),),
 'y_scalar_event': lambda L: L.odeint_event(lambda t,y: -torch.ones_like(y), torch.tensor(1.0), torch.tensor(0.0), event_fn=lambda t,y: y-0.5),
 'big_first_step': lambda L: g(L, lambda: base.clone(), options=dict(first_step=5.0)),
 'max_num_steps_exact': lambda L: g(L, lambda: base.clone(), options=dict(max_num_steps=6)),
 'tsit5_t_dec': lambda L: g(L, lambda: base.clone(), method='tsit5', tt=torch.linspace(1,0,4)),
 'heun_adj': lambda L: g(L, lambda: base.clone(), method='adaptive_heun', adjoint=True),
 'fehlberg2': lambda L: g(L, lambda: base.clone(), method='fehlberg2'),
 'explicit_adams': lambda L: g(L, lambda: base.clone(), method='explicit_adams', options=dict(step_size=0.05)),
 'implicit_adams_tol': lambda L: g(L, lambda: base.clone(), method='implicit_adams', rtol=1e-3, atol=1e-4, options=dict(step_size=0.05,max_iters=2)),
 'adams_max_order': lambda L: g(L, lambda: base.clone(), method='explicit_adams', options=dict(step_size=0.05,max_order=2)),
 'adams_max_order_bad': lambda L: g(L, lambda: base.clone(), method='explicit_adams', options=dict(step_size=0.05,max_order=20)),
}
nd=0
for name,call in cases.items():
    a=run(ref,call); b=run(tda,call)
    same=a[0]==b[0]; detail=''
    if same and a[0]=='ok':
        ra,rb=a[1],b[1]
        if len(ra)!=len(rb): same=False; detail='len'
        for x,y in zip(ra,rb):
            if x is None or y is None:
                if (x is None)!=(y is None): same=False; detail+=' None-mismatch'
                continue
            if x.shape!=y.shape or x.dtype!=y.dtype: same=False; detail+=f' shape/dtype {tuple(x.shape)}{x.dtype} vs {tuple(y.shape)}{y.dtype}'; continue
            xx,yy=(x.to(torch.complex128),y.to(torch.complex128)) if x.is_complex() else (x.double(),y.double())
            d=((xx-yy).abs().max()/(xx.abs().max()+1e-30)).item() if x.numel() else 0.
            tol=2e-5 if x.dtype in (torch.float32,torch.complex64) else (1e-11 if x.dtype in (torch.float64,torch.complex128) else 0.)
            if not d<=tol: same=False
            detail+=f' {d:.0e}'
    else: detail=f'REF={a[1]} || OURS={b[1]}'
    if not same: nd+=1
    print(('SAME' if same else 'DIFF'), name, a[0], b[0], str(detail)[:300])
print('non-same',nd)
