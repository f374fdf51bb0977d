import sys, torch
variant = sys.argv[1]
dev = torch.device('cuda:0')
net = torch.nn.Sequential(torch.nn.Linear(8, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8)).to(dev)
params = tuple(net.parameters())
names = [n for n, _ in net.named_parameters()]
y = torch.randn(16, 8, device=dev); a = torch.randn(16, 8, device=dev)


def vjp_direct(y, a):
    with torch.enable_grad():
        yy = y.detach().requires_grad_(True)
        f = net(yy)
        g = torch.autograd.grad(f, (yy,) + params, a)
    return f.detach(), g


def vjp_proxy(y, a):
    with torch.enable_grad():
        yy = y.detach().requires_grad_(True)
        q = [p.detach().requires_grad_(True) for p in params]
        f = torch.func.functional_call(net, dict(zip(names, q)), (yy,))
        g = torch.autograd.grad(f, [yy] + q, a)
    return f.detach(), g


body = vjp_direct if variant == 'direct' else vjp_proxy


def capture_and_check(where):
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        body(y, a)
        graph.capture_begin()
        try:
            res = body(y, a)
        finally:
            graph.capture_end()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    with torch.no_grad():
        params[0].mul_(0.9)          # optimizer-style in-place update: the replay must see it
    graph.replay(); torch.cuda.synchronize()
    ref = vjp_direct(y, a)
    print(where, variant, 'ok', all(torch.allclose(p, q) for p, q in zip(res[1], ref[1])), flush=True)


class Outer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *ps):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        with torch.no_grad():
            capture_and_check('engine-thread, params are inputs of the running outer graph')
        return (g,) + (None,) * len(params)


# an ordinary training-style pass first: AccumulateGrad nodes of the params now exist (default stream)
net(y).sum().backward()
x = torch.randn(3, device=dev, requires_grad=True)
Outer.apply(x, *params).sum().backward()
print('done', flush=True)
