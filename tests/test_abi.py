"""
The C-ABI library loads and exports every symbol include/csrk.h declares; the ctypes table in
csr_amd/_lib.py covers exactly that set; and without a GPU the product fails loudly instead of
falling back to anything.  No compute on this path (CPU suite).
"""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'csrk.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'CSRK_API\s+[\w\s\*]+?\b(csrk_\w+)\s*\(', text)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 25 and 'csrk_spmv' in names and 'csrk_transpose' in names
    lib = ctypes.CDLL(os.path.join(ROOT, 'csr_amd', 'libcsrk.so'))
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/csrk.h but not exported'


def test_ctypes_table_matches_header():
    from csr_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_only_csrk_symbols_are_exported():
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', os.path.join(ROOT, 'csr_amd', 'libcsrk.so')],
                         capture_output=True, text=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if ' T ' in l]
    ours = [s for s in syms if not s.startswith('_') and not s.startswith('__hip')]
    assert ours and all(s.startswith('csrk_') for s in ours), [s for s in ours if not s.startswith('csrk_')][:5]


def test_version_and_error_string():
    from csr_amd._lib import lib
    assert lib.csrk_version() >= 1
    assert lib.csrk_free(0) == 0            # idempotent on the null handle
    assert lib.csrk_free(12345) != 0        # not a handle: error code, no crash
    assert b'invalid csrk handle' in lib.csrk_last_error()


def test_product_has_no_cpu_fallback_and_never_imports_the_oracle():
    "the product package must not reference oracle/ in any form"
    pkg = os.path.join(ROOT, 'csr_amd')
    for dp, _, fns in os.walk(pkg):
        if 'build' in dp.split(os.sep):
            continue
        for fn in fns:
            if fn.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dp, fn)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'liboracle' not in text, fn


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip('a GPU is present')
    from csr_amd import CSR
    from csr_amd._lib import CsrkError
    a = CSR.from_coo(np.array([0, 1]), np.array([1, 0]), np.array([1.0, 2.0]))
    with pytest.raises(CsrkError) as ei:
        a.mult_vec(np.ones(2))
    assert 'hip' in str(ei.value).lower()
    with pytest.raises(CsrkError):
        a.transpose()


def _build_c_consumer(tmp_path):
    "tests/c_abi/consumer.c against include/csrk.h and the shipped library, as strict C99"
    import subprocess
    exe = str(tmp_path / 'consumer')
    libdir = os.path.join(ROOT, 'csr_amd')
    r = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(ROOT, 'include'),
                        os.path.join(ROOT, 'tests', 'c_abi', 'consumer.c'), '-o', exe, '-L', libdir, '-l:libcsrk.so',
                        '-Wl,-rpath,' + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    "the boundary is a C ABI: a C99 translation unit that includes csrk.h compiles without a warning and links"
    assert os.path.exists(_build_c_consumer(tmp_path))


@pytest.mark.gpu
def test_c_consumer_runs(tmp_path):
    "the same program on the card: create / mult_vec / row extents / transpose / export / free on the reference's known answers"
    import subprocess
    r = subprocess.run([_build_c_consumer(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert 'c consumer ok' in r.stdout


def test_entry_points_are_callable_by_raw_address():
    """
    SURVEY.md section 8f-4: a nopython caller binds an address + a C signature (the cffi pattern of
    csr/kernels/mkl/_api.py:8-21).  Every exported entry is rebuilt from its raw address with CFUNCTYPE and must
    behave like the bound symbol -- checked here with calls that need no GPU (status codes, error text).
    """
    import ctypes as C
    from csr_amd import _lib
    from csr_amd.kernels import raw
    tab = raw.table()
    assert set(tab) == set(_lib.SIGNATURES) and all(addr for addr, _, _ in tab.values())
    assert len({addr for addr, _, _ in tab.values()}) == len(tab)                 # distinct symbols, distinct addresses
    assert raw.function('csrk_version')() == _lib.lib.csrk_version()
    spmv = C.CFUNCTYPE(C.c_int, C.c_ssize_t, C.c_void_p, C.c_void_p)(raw.address('csrk_spmv'))
    x, y = np.ones(2), np.zeros(2)
    assert spmv(12345, x.ctypes.data, y.ctypes.data) == _lib.lib.csrk_spmv(12345, x.ctypes.data, y.ctypes.data) == _lib.ERR_INVALID
    assert b'invalid csrk handle' in raw.function('csrk_last_error')()
    assert raw.function('csrk_free')(0) == _lib.OK and raw.function('csrk_free')(777) == _lib.ERR_INVALID
    # create goes as far as the device: without one, the raw call reports CSRK_ERR_HIP exactly like the bound symbol
    import torch
    if torch.cuda.device_count() == 0:
        rp, ci, vs = np.array([0, 1, 2], np.int32), np.array([1, 0], np.int32), np.array([1.0, 2.0])
        h = _lib.handle_t(0)
        rc = raw.function('csrk_create')(2, 2, 2, rp.ctypes.data, 0, ci.ctypes.data, vs.ctypes.data, _lib.VAL_F64, C.byref(h))
        assert rc == _lib.ERR_HIP and h.value == 0


@pytest.mark.gpu
def test_raw_address_calls_compute_the_known_answers():
    "the protocol sequence to_handle -> mult_vec -> transpose -> release through raw addresses only (tests/test_transpose.py:11-27 KAT)"
    import ctypes as C
    from csr_amd import _lib
    from csr_amd.kernels import raw
    f = {n: raw.function(n) for n in ('csrk_create', 'csrk_spmv', 'csrk_transpose', 'csrk_export', 'csrk_info', 'csrk_free')}
    rp, ci, vs = np.array([0, 2, 3, 3, 4], np.int32), np.array([1, 2, 0, 1], np.int32), np.arange(4.0)
    h, t = _lib.handle_t(0), _lib.handle_t(0)
    assert f['csrk_create'](4, 3, 4, rp.ctypes.data, 0, ci.ctypes.data, vs.ctypes.data, _lib.VAL_F64, C.byref(h)) == 0
    x, y = np.array([1.0, 10.0, 100.0]), np.empty(4)
    assert f['csrk_spmv'](h.value, x.ctypes.data, y.ctypes.data) == 0
    assert np.array_equal(y, [100.0, 2.0, 0.0, 30.0])
    assert f['csrk_transpose'](h.value, 1, C.byref(t)) == 0
    trp, tci, tvs = np.empty(4, np.int32), np.empty(4, np.int32), np.empty(4)
    assert f['csrk_export'](t.value, trp.ctypes.data, tci.ctypes.data, tvs.ctypes.data) == 0
    assert list(trp) == [0, 1, 3, 4] and list(tci) == [1, 0, 3, 0] and list(tvs) == [2.0, 0.0, 3.0, 1.0]
    assert f['csrk_free'](t.value) == 0 and f['csrk_free'](h.value) == 0
