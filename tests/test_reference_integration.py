"""
INTEGRATION.md section 1, executed: the `hip` kernel module registers into the REFERENCE's own
registry (csr/kernels/__init__.py:7) and the reference's CSR.mult_vec / CSR.multiply dispatch into
libcsrk.  Runs only where the reference checkout exists (this container; never on the GPU box), in a
subprocess with the stand-in numba package.  Without a GPU the call must reach libcsrk and fail loudly
there (CsrkError from csrk_create), which proves the routing; with a GPU it must produce the product.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

SCRIPT = r'''
import sys, numpy as np
sys.dont_write_bytecode = True
import csr, csr.kernels
from csr import CSR
import csr_amd.kernels.hip as hip
from csr_amd._lib import CsrkError
csr.kernels.kernels['hip'] = hip
A = CSR.from_coo(np.array([0, 0, 1, 3]), np.array([1, 2, 0, 1]), np.arange(4.0))
with csr.kernels.use_kernel('hip'):
    assert csr.kernels.get_kernel() is hip
    try:
        y = A.mult_vec(np.ones(3))
        assert list(y) == [1.0, 2.0, 0.0, 3.0], y
        P = A.multiply(A.transpose())
        assert sorted(P.values.tolist()) == [1.0, 4.0, 9.0]
        print('COMPUTED')
    except CsrkError as e:
        assert 'hip' in str(e).lower()
        print('ROUTED')
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'csr')), reason='reference checkout not present')
def test_hip_kernel_registers_into_reference_registry():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1',
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'oracle', 'gen', 'numba_stub'), REF, ROOT]))
    r = subprocess.run([sys.executable, '-c', SCRIPT], env=env, capture_output=True, text=True, cwd='/tmp')
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] in ('ROUTED', 'COMPUTED')
