def odeint_adjoint(*a, **k):
    raise NotImplementedError
