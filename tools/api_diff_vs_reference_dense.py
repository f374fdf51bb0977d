"""Differential batch against the imported reference (build container; see api_diff_vs_reference.py): prints SAME / DIFF
per case — exception class, shapes, dtypes, values, gradients."""
import sys, torch, warnings, math
sys.path.insert(0,'/root/reference'); import torchdiffeq as ref
sys.path.insert(0,'/root/repo'); import torchdiffeq_amd as tda
f=lambda t,y: torch.stack([y[...,1], -y[...,0]],-1)*(1+0.1*t)
y0=torch.tensor([[1.0,0.0],[0.5,-0.5]])
def dense(L, method='dopri5', t0=0., t1=2., qs=(0.,0.3,1.1,2.0), **kw):
    d=L.odeint_dense(f,y0,torch.tensor(t0),torch.tensor(t1),method=method,**kw)
    return tuple(d(torch.tensor(q)) for q in qs)
def run(lib,call):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        try:
            return ('ok',call(lib),[])
        except Exception as e:
            return (type(e).__name__, str(e)[:100],[])
cases={
 'dense_dopri5': lambda L: dense(L),
 'dense_dopri8': lambda L: dense(L,'dopri8'),
 'dense_tsit5': lambda L: dense(L,'tsit5'),
 'dense_bosh3': lambda L: dense(L,'bosh3'),
 'dense_heun': lambda L: dense(L,'adaptive_heun'),
 'dense_fehlberg2': lambda L: dense(L,'fehlberg2'),
 'dense_rk4': lambda L: dense(L,'rk4'),
 'dense_outside_hi': lambda L: dense(L,qs=(2.5,)),
 'dense_outside_lo': lambda L: dense(L,qs=(-0.5,)),
 'dense_tight': lambda L: dense(L,rtol=1e-10,atol=1e-12),
 'dense_opts': lambda L: dense(L,options=dict(first_step=0.01,max_num_steps=10000)),
 'dense_vecq': lambda L: (L.odeint_dense(f,y0,torch.tensor(0.),torch.tensor(2.))(torch.tensor([0.3,1.1])),),
 'dense_floatq': lambda L: (L.odeint_dense(f,y0,torch.tensor(0.),torch.tensor(2.))(0.7),),
 'dense_float_t': lambda L: (L.odeint_dense(f,y0,0.,2.)(torch.tensor(0.7)),),
 'dense_t0_eq_t1': lambda L: (L.odeint_dense(f,y0,torch.tensor(1.),torch.tensor(1.))(torch.tensor(1.0)),),
 'dense_f64_t': lambda L: (L.odeint_dense(f,y0,torch.tensor(0.,dtype=torch.float64),torch.tensor(2.,dtype=torch.float64))(torch.tensor(0.7,dtype=torch.float64)),),
 'dense_tuple': lambda L: L.odeint_dense(lambda t,y:(-y[0],y[1]),(y0,y0),torch.tensor(0.),torch.tensor(1.))(torch.tensor(0.5)),
}
nd=0
for name,call in cases.items():
    a=run(ref,call); b=run(tda,call)
    same=a[0]==b[0]; detail=''
    if same and a[0]=='ok':
        ra,rb=a[1],b[1]
        if not isinstance(ra,tuple): ra=(ra,)
        if not isinstance(rb,tuple): rb=(rb,)
        if len(ra)!=len(rb): same=False; detail='len'
        for x,y in zip(ra,rb):
            if not torch.is_tensor(x) or not torch.is_tensor(y): detail+=f' types {type(x).__name__}/{type(y).__name__}'; same = same and type(x)==type(y); continue
            if x.shape!=y.shape or x.dtype!=y.dtype: same=False; detail+=f' shape/dtype {tuple(x.shape)}{x.dtype} vs {tuple(y.shape)}{y.dtype}'; continue
            d=(x.double()-y.double()).abs().max().item() if x.numel() else 0.
            if not d<=2e-6: same=False
            detail+=f' {d:.0e}'
    else: detail=f'REF={a[1]} || OURS={b[1]}'
    if not same: nd+=1
    print(('SAME' if same else 'DIFF'), name, a[0], b[0], str(detail)[:260])
print('non-same',nd)
