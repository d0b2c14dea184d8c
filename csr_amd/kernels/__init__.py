"""
Kernel registry: the host-side mirror of the reference's plugin API
(csr/kernels/__init__.py:7-123): a public dict `kernels`, a thread-local active kernel,
`get_kernel / set_kernel / use_kernel / releasing`, and the CSR_KERNEL environment variable.

The only kernel this package ships is `hip` (csr_amd.kernels.hip -> libcsrk.so, MI355X).
There is deliberately no CPU kernel to fall back to.  Out-of-tree kernel modules register
the same way they do with the reference: `csr_amd.kernels.kernels[name] = module`.
"""
import os
import threading
from contextlib import contextmanager
from importlib import import_module

kernels = {}
DEFAULT_KERNEL = 'hip'

__all__ = ['releasing', 'set_kernel', 'use_kernel', 'get_kernel', 'kernels']


class _Active(threading.local):
    def __init__(self):
        self.kern = None
        self.name = None


_active = _Active()
_default = None


@contextmanager
def releasing(h, k):
    "csr/kernels/__init__.py:36-41: release the handle when the block exits"
    try:
        yield h
    finally:
        k.release_handle(h)


def _default_kernel():
    "csr/kernels/__init__.py:100-123: explicit > CSR_KERNEL env > built-in default"
    global _default
    if _default is None:
        _default = get_kernel(os.environ.get('CSR_KERNEL', DEFAULT_KERNEL))
    return _default


def get_kernel(name=None):
    "csr/kernels/__init__.py:81-97"
    if name is None:
        return _active.kern if _active.kern is not None else _default_kernel()
    kern = kernels.get(name)
    if kern is None:
        kern = import_module(f'{__name__}.{name}')
        kernels[name] = kern
    return kern


def set_kernel(name):
    "csr/kernels/__init__.py:44-64 (thread-local)"
    if name is None:
        _active.kern, _active.name = None, None
    else:
        _active.kern, _active.name = get_kernel(name), name


@contextmanager
def use_kernel(name):
    """
    csr/kernels/__init__.py:67-78.  Unlike the reference (whose saved name is never
    updated, so it always falls back to the default on exit -- SURVEY.md section 5 quirks)
    this restores the kernel that was active on entry.
    """
    old = _active.name
    try:
        set_kernel(name)
        yield
    finally:
        set_kernel(old)
