"""
csr_amd: MI355X-native kernel backend for the lenskit/csr `csr.kernels` hot path.

Importing the package requires the built HIP library (csr_amd/libcsrk.so, built by
`python csr_amd/build.py`); there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly if libcsrk.so is missing)
from .csr import CSR  # noqa: F401
from .kernels import get_kernel, set_kernel, use_kernel, releasing  # noqa: F401

__version__ = '0.1.0'
