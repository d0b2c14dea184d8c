"""
libcsrk's entry points as RAW FUNCTION ADDRESSES with their C signatures -- the form a nopython caller binds
(SURVEY.md section 8f-4).  The reference reaches its one native library from Numba-compiled code through cffi
function objects registered with Numba's typing (csr/kernels/mkl/_api.py:8-21) and dispatches the protocol from
jitted code in csr/_wiring.py:116-151; what Numba needs from a library for that is an address and a C prototype, no
Python state.  `address(name)` is the address of the exported symbol; `function(name)` rebuilds a callable from that
address alone with `ctypes.CFUNCTYPE` -- the object Numba's ctypes support accepts inside @njit -- and `table()`
lists every entry of include/csrk.h that way.  Handles are intptr_t and arrays plain pointers, so a jitted caller
passes `arr.ctypes.data` / `handle` integers.  Real Numba is not installable in this image:
tests/test_abi.py exercises these callables from plain Python (CPU: status codes; GPU: the reference's known answers).
"""
import ctypes as C

from .._lib import lib, SIGNATURES


def address(name):
    "address of the exported symbol `name` (an int)"
    if name not in SIGNATURES:
        raise KeyError(name)
    return C.cast(getattr(lib, name), C.c_void_p).value


def prototype(name):
    "ctypes prototype (CFUNCTYPE class) of `name`, as declared in include/csrk.h"
    res, args = SIGNATURES[name]
    return C.CFUNCTYPE(res, *args)


def function(name):
    "a callable built from the raw address and the prototype only (holds no reference to the CDLL object)"
    return prototype(name)(address(name))


def table():
    "name -> (address, restype, argtypes) for every entry point"
    return {n: (address(n),) + tuple(SIGNATURES[n]) for n in SIGNATURES}
