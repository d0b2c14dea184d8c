"""
Minimal stand-in for the `numba` package, used ONLY by oracle/gen/gen_golden.py in the
build container to import the reference (/root/reference) in its NUMBA_DISABLE_JIT mode
(reference: csr/csr.py:20-43).  It makes every decorator the identity so the reference's
hot-path functions run as the plain Python they are written in.  Test infrastructure only;
never imported by the product, never needed on the GPU box.
"""
from . import config, types  # noqa: F401

prange = range


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn
    return deco


jit = njit
