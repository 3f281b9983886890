"""numba stand-in: njit/jit are identity decorators (the reference's semantics are
unchanged; numba is a semantics-neutral JIT). Supports @njit and @njit(cache=True)."""


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn
    return wrap


njit = _identity_decorator
jit = _identity_decorator
