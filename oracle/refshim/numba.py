"""Stand-in for numba when importing the reference in the build container (no numba wheel, no network):
``njit`` returns the function unchanged, ``prange`` is ``range``.  Used only by oracle/make_golden.py."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


jit = njit
prange = range
