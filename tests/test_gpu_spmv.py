"""
GPU parity: mult_vec through the C ABI (csr_amd.kernels.hip -> libcsrk.so) against the CPU
oracle and the golden vectors captured from the reference.

Tolerance (north_star: 1e-6 relative for fp64 SpMV): the HIP kernels sum long rows in a
different order than the reference's left-to-right loop, so the comparison is
|y - y_ref| <= 1e-6 * sum_j |a_ij x_j| (a relative bound that is meaningful under
cancellation, SURVEY.md section 7 hard-part 4); in practice the error is ~1e-15 of that sum
and the tests also assert the far tighter 1e-12.
"""
import numpy as np
import pytest

from conftest import Mat

pytestmark = pytest.mark.gpu

ALGOS = ['merge', 'vector', 'scalar']


@pytest.fixture(autouse=True, params=['auto', 'forced_split', 'hot', 'forced_split_hot', 'forced_split_nostream'])
def split_mode(request, monkeypatch):
    """
    Every test runs five times: with the library's own choice (small test matrices have an x that fits in L2, so the
    long-row split and the hot-column pack stay off: the merge-path tile kernel alone), with the split forced on (both
    tiers on every shape), with the hot-column pack forced on (light stream with LDS / packed / staged x), with split and
    pack both (the form the headline matrix runs in), and with the split forced on but the light stream off (what a
    handle gets when the stream's copy of the matrix does not fit in memory: the tile kernel in its cut-table form
    beside the tiers).
    """
    for k in ('CSRK_SPMV_HEAVY_SPLIT', 'CSRK_SPMV_STREAM', 'CSRK_SPMV_HOT'):
        monkeypatch.delenv(k, raising=False)
    if 'forced_split' in request.param:
        monkeypatch.setenv('CSRK_SPMV_HEAVY_SPLIT', '1')
    if 'nostream' in request.param:
        monkeypatch.setenv('CSRK_SPMV_STREAM', '0')
    if 'hot' in request.param:
        monkeypatch.setenv('CSRK_SPMV_HOT', '1')
    return request.param


def _abs_bound(m, x):
    from oracle import oracle as O
    vs = None if m.values is None else np.abs(m.values.astype(np.float64))
    return O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, vs, np.abs(np.asarray(x, dtype=np.float64)))


def _check(y, ref, bound):
    assert y.dtype == np.float64 and y.shape == ref.shape
    err = np.abs(y - ref)
    assert np.all(err <= 1e-6 * bound + 1e-300), float(np.max(err / (bound + 1e-300)))
    assert np.all(err <= 1e-12 * bound + 1e-300), float(np.max(err / (bound + 1e-300)))


def _mult_vec(m, x, algo):
    from csr_amd.kernels import hip as K
    from csr_amd import CSR
    A = CSR(m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, m.values, _cast=False)
    h = K.to_handle(A)
    try:
        K.set_spmv_algo(h, algo)
        return K.mult_vec(h, x)
    finally:
        K.release_handle(h)


@pytest.mark.parametrize('algo', ALGOS)
def test_spmv_golden(golden, algo):
    "reference outputs for the csrs() distribution: f4/f8/structure-only, empty rows, nnz=0"
    g = golden('spmv')
    for c in range(int(g['n'])):
        m = Mat(g, f'c{c}_')
        x = g[f'c{c}_x']
        y = _mult_vec(m, x, algo)
        # (float32 values times a float32 x: the reference's products are float32 -- csrk_spmv_f32x rounds them the same
        # way, so this case meets the same bar as the others)
        _check(y, g[f'c{c}_y'], _abs_bound(m, x))


@pytest.mark.parametrize('algo', ALGOS)
def test_spmv_cfg1(golden, algo):
    "BASELINE.json configs[0]: 10k x 10k, nnz = 1e5, fp64, against the reference's output"
    g = golden('cfg1_spmv')
    a = Mat(g, 'a_')
    y = _mult_vec(a, g['x'], algo)
    _check(y, g['y'], _abs_bound(a, g['x']))
    if algo == 'merge':
        # rows not cut by a tile boundary (2048-item merge tiles, 512-entry stream tiles) are summed in
        # storage order, products rounded on their own: bit-identical.  (The stream hands carries over in
        # order across up to 4 lanes = rows of up to 25 entries wherever they lie.)
        lens = np.diff(a.rowptrs)
        assert np.sum(y != g['y']) <= a.nnz // 512 + 1 + int(np.sum(lens > 25))


def test_kat_and_protocol(golden):
    "tests/test_mult_vec.py + conftest.py:33-35 of the reference: 1x1 empty warm-up, fixed KAT"
    from csr_amd import CSR
    from csr_amd.kernels import get_kernel
    K = get_kernel()
    assert K.__name__ == 'csr_amd.kernels.hip'
    m = CSR.empty(1, 1)
    h = K.to_handle(m)
    assert K.mult_vec(h, np.ones(1)).tolist() == [0.0]
    K.release_handle(h)
    K.release_handle(h)     # idempotent
    g = golden('kat')
    a = Mat(g, 'a_')
    A = CSR(a.nrows, a.ncols, a.nnz, a.rowptrs, a.colinds, a.values)
    assert np.array_equal(A.mult_vec(np.ones(3)), g['a_mv_ones'])
    with pytest.raises(AssertionError):
        A.mult_vec(np.ones(4))


def _random_csr(rng, nrows, ncols, lens, dtype=np.float64, ptr64=False, sort=False):
    rp = np.zeros(nrows + 1, dtype=np.int64 if ptr64 else np.int32)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    ci = rng.integers(0, ncols, size=nnz).astype(np.int32)
    if sort:     # ascending columns inside rows (duplicates allowed): enables the heavy-row split
        rows = np.repeat(np.arange(nrows), np.diff(rp))
        ci = ci[np.lexsort((ci, rows))]
    vs = rng.uniform(-1, 1, size=nnz).astype(dtype)

    class M:
        pass
    m = M()
    m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, m.values = nrows, ncols, nnz, rp, ci, vs
    return m


SHAPES = ['powerlaw', 'powerlaw_sorted', 'tile_edges', 'tile_edges_sorted', 'one_huge_row', 'one_huge_row_sorted',
          'adjacent_heavy_sorted', 'all_heavy_sorted', 'all_empty', 'ptr64_f32', 'ptr64_heavy_sorted']


@pytest.mark.parametrize('algo', ALGOS)
@pytest.mark.parametrize('shape', SHAPES)
def test_spmv_shapes(algo, shape):
    """
    Row-length distributions that stress tile boundaries, carries, the long-row path and -- with
    ascending columns -- the heavy-row split (rows >= 2048 entries cut out of the merge path and
    processed in column blocks): heavy rows next to each other, at the matrix ends, heavy rows
    only, wide matrices with several column blocks.
    """
    from oracle import oracle as O
    import zlib
    rng = np.random.default_rng(zlib.crc32(shape.encode()))
    srt = shape.endswith('_sorted')
    if shape.startswith('powerlaw'):
        nrows = 30000
        lens = np.minimum((rng.pareto(0.9, nrows) * 2).astype(np.int64), 60000)
        m = _random_csr(rng, nrows, 400000, lens, sort=srt)
    elif shape.startswith('tile_edges'):
        # rows of exactly 2047/2048/2049/63/64/65 entries interleaved with empties
        lens = np.array([2047, 0, 2048, 1, 2049, 0, 0, 63, 64, 65, 4096, 1, 1, 1, 6000] * 40)
        m = _random_csr(rng, len(lens), 5000, lens, sort=srt)
    elif shape.startswith('one_huge_row'):
        lens = np.zeros(5000, dtype=np.int64)
        lens[2500] = 700000
        lens[10] = 3
        m = _random_csr(rng, 5000, 1000000, lens, sort=srt)
    elif shape == 'adjacent_heavy_sorted':
        lens = rng.integers(0, 6, size=3000)
        lens[0] = 5000          # first row heavy
        lens[100:104] = [3000, 2048, 0, 9000]
        lens[-1] = 2500         # last row heavy
        m = _random_csr(rng, 3000, 300000, lens, sort=True)
    elif shape == 'all_heavy_sorted':
        m = _random_csr(rng, 40, 200000, np.full(40, 3000), sort=True)
    elif shape == 'all_empty':
        m = _random_csr(rng, 100000, 10, np.zeros(100000, dtype=np.int64))
    elif shape == 'ptr64_heavy_sorted':
        lens = rng.integers(0, 10, size=5000)
        lens[::500] = 4000
        m = _random_csr(rng, 5000, 150000, lens, dtype=np.float32, ptr64=True, sort=True)
    else:
        lens = rng.integers(0, 40, size=20000)
        m = _random_csr(rng, 20000, 3000, lens, dtype=np.float32, ptr64=True)
    x = rng.uniform(-1, 1, size=m.ncols)
    y = _mult_vec(m, x, algo)
    ref = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x)
    _check(y, ref, _abs_bound(m, x))


def test_spmv_many_heavy_rows(monkeypatch, split_mode):
    """
    More long rows than one accumulator group holds (15360): tier 0 runs as several groups.  The
    threshold is lowered so that a small matrix has that many "heavy" rows; rows of every length
    around a lane's 8 entries and a tile's 512, dense runs inside one column block, a last block
    narrower than 4096 columns.
    """
    from oracle import oracle as O
    if 'forced_split' not in split_mode:
        pytest.skip('needs the split')
    monkeypatch.setenv('CSRK_HEAVY_MIN', '64')
    monkeypatch.setenv('CSRK_TIERB_MIN', '0')
    rng = np.random.default_rng(2026)
    nrows = 19000
    lens = rng.integers(64, 130, size=nrows)
    lens[::7] = 3                       # light rows in between (16285 rows stay >= 64: two groups)
    lens[5] = 30000                     # > one column block's worth per block
    lens[9000] = 511
    lens[9001] = 512
    lens[9002] = 513
    m = _random_csr(rng, nrows, 4096 * 9 + 100, lens, sort=True)
    # a dense run: one row with every column of block 2
    s, e = int(m.rowptrs[5]), int(m.rowptrs[5]) + 4096
    m.colinds[s:e] = np.arange(2 * 4096, 3 * 4096, dtype=np.int32)
    m.colinds[int(m.rowptrs[5]):int(m.rowptrs[6])].sort()
    x = rng.uniform(-1, 1, size=m.ncols)
    y = _mult_vec(m, x, 'merge')
    ref = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x)
    _check(y, ref, _abs_bound(m, x))


def test_spmv_structure_only_and_f32_in_the_tiers():
    "values absent (every entry counts 1.0, csr/csr.py:254-262) or float32, with rows long enough for every tier"
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    lens = rng.integers(0, 30, size=6000)
    lens[[5, 900, 4000]] = [5000, 700, 2049]
    lens[100:110] = 200
    for dtype in (None, np.float32):
        m = _random_csr(rng, 6000, 40000, lens, dtype=np.float32 if dtype else np.float64, sort=True)
        if dtype is None:
            m.values = None
        x = rng.uniform(-1, 1, size=m.ncols)
        for algo in ALGOS:
            y = _mult_vec(m, x, algo)
            ref = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x)
            _check(y, ref, _abs_bound(m, x))


def test_spmv_deterministic():
    "no float atomics: repeated launches are bitwise identical"
    rng = np.random.default_rng(11)
    lens = np.minimum((rng.pareto(0.8, 20000) * 3).astype(np.int64), 100000)
    m = _random_csr(rng, 20000, 400000, lens, sort=True)
    x = rng.uniform(-1, 1, size=m.ncols)
    ys = [_mult_vec(m, x, 'merge') for _ in range(3)]
    assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])


def test_spmv_sharded_caller(golden):
    "csr/csr.py:584-590: CSR.mult_vec shards by rows when nnz > K.max_nnz (tests/test_mkl.py:76-79)"
    from csr_amd import CSR
    from csr_amd.kernels import hip as K
    g = golden('spmv')
    save = K.max_nnz
    try:
        K.max_nnz = 40
        hits = 0
        for c in range(int(g['n'])):
            if f'c{c}_y_sharded' not in g:
                continue
            hits += 1
            m = Mat(g, f'c{c}_')
            A = CSR(m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, m.values)
            y = A.mult_vec(g[f'c{c}_x'])
            assert np.all(np.abs(y - g[f'c{c}_y_sharded']) <= 1e-6 * _abs_bound(m, g[f'c{c}_x']) + 1e-300)
        assert hits > 5
    finally:
        K.max_nnz = save


def test_spmv_nonfinite_propagation():
    """
    inf / NaN in x or in the values must reach exactly the rows the reference's loop lets them reach
    (masked lanes of a tile must not turn 0 * inf into NaN for a neighbouring row).
    """
    from oracle import oracle as O
    rng = np.random.default_rng(99)
    lens = rng.integers(0, 12, size=6000)
    lens[100] = 5000          # tier 0 row
    lens[200] = 300           # tier 1 row
    lens[-1] = 3              # last row: the final, partly filled tile
    m = _random_csr(rng, 6000, 20000, lens, sort=True)
    x = rng.uniform(-1, 1, size=m.ncols)
    x[17] = np.inf
    x[4000] = -np.inf
    x[9000] = np.nan
    x[int(m.colinds[-1])] = np.inf          # the matrix's very last entry gathers an inf
    m.values[5] = np.nan
    with np.errstate(all='ignore'):
        ref = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x)
        for algo in ALGOS:
            y = _mult_vec(m, x, algo)
            assert np.array_equal(np.isnan(y), np.isnan(ref)), algo
            assert np.array_equal(np.isposinf(y), np.isposinf(ref)) and np.array_equal(np.isneginf(y), np.isneginf(ref)), algo
            fin = np.isfinite(ref)
            assert np.allclose(y[fin], ref[fin], rtol=1e-9, atol=1e-12), algo


def test_plan_stats_and_cache_trim(split_mode):
    "csrk_spmv_plan_stats reports the tiers; csrk_trim_cache returns the pool to the driver"
    if 'forced_split' not in split_mode:
        pytest.skip('x of this small matrix fits in L2: no split unless forced')
    import ctypes as C
    from csr_amd._lib import lib, check
    from csr_amd.kernels import hip as K
    from csr_amd import CSR
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 10, size=4000)
    lens[7] = 4000
    lens[9] = 500
    m = _random_csr(rng, 4000, 30000, lens, sort=True)
    h = K.to_handle(CSR(m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, m.values, _cast=False))
    try:
        K.mult_vec(h, np.ones(m.ncols))
        st = (C.c_int64 * 20)()
        check(lib.csrk_spmv_plan_stats(h.H, st, 20))
        assert st[2] == 2                      # two rows cut out of the tile path
        # tier-0 / tier-1 entries: the accumulator tier takes every cut row down to 128 entries while it has room (ACC_FLOOR)
        assert st[10] == 4500 and st[13] == 0
        assert st[3] == m.nnz - 4500
        if 'hot' in split_mode:
            # columns referenced at least twice by the tile path's rows are packed
            assert 0 < st[16] <= st[19] and 0 < st[17] <= 1_000_000
        else:
            assert st[16] == 0
    finally:
        K.release_handle(h)
    check(lib.csrk_trim_cache())


def test_lazy_split_on_second_call(split_mode):
    """
    Auto mode builds the long-row split on the SECOND launch on a handle (the reference's
    CSR.mult_vec makes a handle per call and must not pay for the plan): first result from the single
    merge path, later ones from the tiered kernels, all within tolerance of the oracle.
    """
    import ctypes as C
    from oracle import oracle as O
    from csr_amd._lib import lib, check
    from csr_amd.kernels import hip as K
    from csr_amd import CSR
    rng = np.random.default_rng(77)
    lens = rng.integers(0, 9, size=5000)
    lens[[3, 900, 4999]] = [6000, 2500, 700]
    m = _random_csr(rng, 5000, 700000, lens, sort=True)       # x = 5.6 MB > L2: split eligible
    x = rng.uniform(-1, 1, size=m.ncols)
    ref = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x)
    bound = _abs_bound(m, x)
    h = K.to_handle(CSR(m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, m.values, _cast=False))
    try:
        ys, cut = [], []
        for _ in range(3):
            ys.append(K.mult_vec(h, x))
            st = (C.c_int64 * 16)()
            check(lib.csrk_spmv_plan_stats(h.H, st, 16))
            cut.append(int(st[2]))
        for y in ys:
            _check(y, ref, bound)
        assert np.array_equal(ys[1], ys[2])
        assert cut[1] == 3 and cut[2] == 3        # rows 3, 900, 4999 are in panels from the 2nd call on
    finally:
        K.release_handle(h)


def test_concurrent_calls_on_one_handle():
    "the reference's kernels are nogil: several Python threads may multiply with the same handle"
    import threading
    from oracle import oracle as O
    from csr_amd.kernels import hip as K
    from csr_amd import CSR
    rng = np.random.default_rng(3)
    m = _random_csr(rng, 20000, 5000, rng.integers(0, 30, size=20000))
    h = K.to_handle(CSR(m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, m.values, _cast=False))
    xs = [rng.uniform(-1, 1, size=m.ncols) for _ in range(8)]
    refs = [O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x) for x in xs]
    outs = [None] * 8
    try:
        K.mult_vec(h, xs[0])

        def work(i):
            for _ in range(5):
                outs[i] = K.mult_vec(h, xs[i])
        ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    finally:
        K.release_handle(h)
    for y, r in zip(outs, refs):
        assert np.allclose(y, r, rtol=1e-12, atol=1e-12)


def test_hot_pack_is_bit_identical(monkeypatch, split_mode):
    """
    The hot-column pack only changes WHERE an x value is read from (a packed copy instead of x itself):
    products and summation order are those of the same kernel without the pack, so y is bit-identical.
    """
    if 'nostream' not in split_mode:
        monkeypatch.setenv('CSRK_SPMV_STREAM', '1')      # both runs on the light stream (eager plan)
    from csr_amd.kernels import hip as K
    from csr_amd import CSR
    rng = np.random.default_rng(4242)
    lens = rng.integers(0, 40, size=3000)
    lens[11] = 3000
    nc = 9000
    # Zipf-like column popularity so that some columns are referenced by many rows
    pop = 1.0 / np.arange(1, nc + 1)
    pop /= pop.sum()
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(nc, size=int(n), replace=False, p=pop)) for n in lens]).astype(np.int32)
    vs = rng.uniform(-1, 1, size=ci.size)
    x = rng.uniform(-1, 1, size=nc)
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('CSRK_SPMV_HOT', mode)
        h = K.to_handle(CSR(3000, nc, int(ci.size), rp, ci, vs, _cast=False))
        try:
            out[mode] = K.mult_vec(h, x)
        finally:
            K.release_handle(h)
    assert np.array_equal(out['0'], out['1'])


def test_two_part_product_equals_the_whole(split_mode):
    "csrk_spmv_device_part: part 1 (row-major path, cut rows zeroed) then part 2 (the tiers' rows) = part 3, bit for bit"
    import ctypes as C
    import torch
    from csr_amd._lib import lib, check
    from csr_amd.kernels import hip as K
    from csr_amd import CSR
    rng = np.random.default_rng(11)
    lens = rng.integers(0, 12, size=6000)
    lens[[5, 77, 3000, 5999]] = [5000, 900, 300, 2500]
    m = _random_csr(rng, 6000, 40000, lens, sort=True)
    x = rng.uniform(-1, 1, size=m.ncols)
    h = K.to_handle(CSR(m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, m.values, _cast=False))
    try:
        xd = torch.from_numpy(x).cuda()
        whole = torch.empty(m.nrows, dtype=torch.float64, device='cuda')
        for _ in range(2):      # the second call runs the planned path
            check(lib.csrk_spmv_device(h.H, xd.data_ptr(), whole.data_ptr(), None))
        n = C.c_int64(0)
        check(lib.csrk_spmv_cut_rows(h.H, None, 0, C.byref(n)))
        parts = torch.full((m.nrows,), 7.0, dtype=torch.float64, device='cuda')
        check(lib.csrk_spmv_device_part(h.H, xd.data_ptr(), parts.data_ptr(), None, 1))
        if n.value:
            rows = torch.zeros(n.value, dtype=torch.int32, device='cuda')
            check(lib.csrk_spmv_cut_rows(h.H, rows.data_ptr(), n.value, C.byref(n)))
            rl = rows.cpu().numpy()
            assert np.all(np.diff(rl) > 0) and set(rl) <= {5, 77, 3000, 5999}
            after1 = parts.cpu().numpy()
            assert np.all(after1[rl] == 0.0)                                   # the cut rows hold 0.0 after part 1
            keep = np.ones(m.nrows, dtype=bool)
            keep[rl] = False
            assert np.array_equal(after1[keep], whole.cpu().numpy()[keep])     # every other row is final
        check(lib.csrk_spmv_device_part(h.H, xd.data_ptr(), parts.data_ptr(), None, 2))
        torch.cuda.synchronize()
        assert torch.equal(parts, whole)
        with pytest.raises(Exception):
            check(lib.csrk_spmv_device_part(h.H, xd.data_ptr(), parts.data_ptr(), None, 0))
    finally:
        K.release_handle(h)


def test_handle_cache_reaches_the_plan():
    """
    The reference's caller path (csr/csr.py:580-583: a handle per product) through csr_amd.CSR: the device copy made by
    the first product is handed out again to the later ones (same csrk handle), so they run on the plan the second
    product builds; results stay identical to the oracle's and to each other.  An edit of the host arrays between two
    products is never missed (the reference re-reads them on every product, csr/csr.py:580-583): while the device copy
    is cached the arrays are write-protected, so ONE poked element raises; after `invalidate` the edit goes through and
    the next product sees it.
    """
    from oracle import oracle as O
    from csr_amd import CSR, synth
    from csr_amd.kernels import hip as K
    K.flush_handle_cache()
    w = synth.powerlaw_csr(40000, 600000, 1500000, device='cpu')
    # (arrays that own their memory: views -- torch-backed ones included -- are never cached under the write guard)
    A = CSR(40000, 600000, 1500000, w['rowptrs'].numpy().copy(), w['colinds'].numpy().copy(), w['values'].numpy().copy())
    x = synth.dense_vector(600000).numpy()

    def check(y):
        ref = O.mult_vec(A.nrows, A.ncols, A.rowptrs, A.colinds, A.values, x)
        bound = O.mult_vec(A.nrows, A.ncols, A.rowptrs, A.colinds, np.abs(A.values), np.abs(x))
        assert np.all(np.abs(y - ref) <= 1e-12 * bound + 1e-300)

    h = K.to_handle(A)
    H0 = h.H
    K.release_handle(h)
    ys = [A.mult_vec(x) for _ in range(4)]                  # product 1: plan-less kernel; 2: builds the plan; 3, 4: planned
    h = K.to_handle(A)
    assert h.H == H0                                        # still the same device copy
    K.release_handle(h)
    for y in ys:
        check(y)
    assert np.array_equal(ys[2], ys[3])
    assert any(k[0] == id(A) for k in K._cache) and not A.values.flags.writeable      # cached, and guarded
    y_old = ys[3]
    with pytest.raises(ValueError):                         # ONE element edited between two products: refused, never ignored
        A.values[777] = 3.0
    with pytest.raises(ValueError):
        A.values *= 0.5
    K.invalidate(A)                                         # announce the edit: the copy is dropped, the arrays writable again
    A.values[777] = 3.0
    y_new = A.mult_vec(x)
    check(y_new)                                            # the NEW product (the oracle reads the edited array) ...
    assert not np.array_equal(y_new, y_old)                 # ... not the old one
    K.invalidate(A)
    A.values *= 0.5
    check(A.mult_vec(x))
    A.values = A.values + 1.0                               # csr_amd.CSR's own setter invalidates by itself
    check(A.mult_vec(x))
    K.flush_handle_cache()
    assert A.values.flags.writeable


def test_live_handles_are_independent():
    """
    Two live handles on one CSR (csr/csr.py:543-567 makes them for A.multiply(A)): an in-place protocol operation on one
    -- order_columns, csr/kernels/numba/__init__.py:47-52 -- must not change what the other reads.
    """
    from csr_amd import CSR
    from csr_amd.kernels import hip as K
    K.flush_handle_cache()
    rng = np.random.default_rng(5)
    n, per = 3000, 6
    cols = np.concatenate([rng.permutation(n)[:per] for _ in range(n)]).astype(np.int32)      # unsorted rows
    A = CSR(n, n, n * per, np.arange(0, n * per + 1, per, dtype=np.int32), cols, rng.uniform(-1, 1, n * per), _cast=False)
    h1, h2 = K.to_handle(A), K.to_handle(A)
    assert h1.H != h2.H
    K.order_columns(h2)
    a1, a2 = K.from_handle(h1), K.from_handle(h2)
    assert np.array_equal(a1.colinds, A.colinds) and np.array_equal(a1.values, A.values)      # h1 untouched
    assert not np.array_equal(a2.colinds, A.colinds)
    assert all(np.all(np.diff(a2.colinds[i * per:(i + 1) * per]) > 0) for i in range(0, n, 97))
    K.release_handle(h1)
    K.release_handle(h2)
    K.flush_handle_cache()


@pytest.mark.parametrize('vals', ['f4', 'f8', 'none'])
def test_spmv_float32_vector(vals):
    """
    A float32 x (csr/kernels/numba/__init__.py:55-67 as Numba types it): with float32 values every product is a float32
    -- one rounding -- added to the float64 accumulator; with float64 or absent values x is widened.  csrk_spmv_f32x
    against the oracle's restatement of each case (orc_mult_vec_f32f32 / the float64 loops), rows up to 5000 entries,
    int64 row pointers in the float32 case.
    """
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    lens = rng.integers(0, 40, 3000)
    lens[::97] = 5000
    m = _random_csr(rng, 3000, 20000, lens, dtype=np.float32 if vals == 'f4' else np.float64, ptr64=vals == 'f4')
    if vals == 'none':
        m.values = None
    x = rng.uniform(-1, 1, size=m.ncols).astype(np.float32)
    y = _mult_vec(m, x, 'merge')
    ref = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x)
    _check(y, ref, _abs_bound(m, x))
    if vals == 'f4':
        # and it is NOT the float64 product: the two differ by ~1e-8 of the terms
        ref64 = O.mult_vec(m.nrows, m.ncols, m.rowptrs, m.colinds, m.values, x.astype(np.float64))
        assert np.max(np.abs(ref64 - ref)) > 1e-10


@pytest.mark.parametrize('algo', ['auto', 'merge', 'scalar'])
def test_spmv_float32_product_of_a_one_entry_matrix(algo):
    """
    float32 values times a float32 x on matrices of ONE entry and of none: under the merge algorithm these fall through to
    the one-lane-per-row kernel (the tile kernel's pair loads need two entries), which must round the product to float32
    like every other path (csr/kernels/numba/__init__.py:55-67 as Numba types it).  The entry is chosen so that
    the float32 and the float64 product differ.
    """
    from csr_amd import CSR
    from oracle import oracle as O
    a, b = np.float32(1.0000001), np.float32(3.0000002)
    assert np.float64(a) * np.float64(b) != np.float64(np.float32(a * b))
    one = CSR(3, 4, 1, np.array([0, 0, 1, 1], dtype=np.int32), np.array([2], dtype=np.int32), np.array([a], dtype=np.float32), _cast=False)
    x = np.array([0, 0, b, 0], dtype=np.float32)
    for _ in range(3):                                   # (first call, plan call, planned call)
        y = _mult_vec(one, x, algo)
        ref = O.mult_vec(3, 4, one.rowptrs, one.colinds, one.values, x)
        assert np.array_equal(y, ref) and y[1] == np.float64(np.float32(a * b))
    none = CSR(3, 4, 0, np.zeros(4, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32), _cast=False)
    assert np.array_equal(_mult_vec(none, x, algo), np.zeros(3))


@pytest.mark.parametrize('vals', ['f4', 'f8'])
def test_spmv_float32_vector_on_the_device_and_a_stream(vals):
    """
    csrk_spmv_f32x_device: x (float32) and y already on the card, launched on a caller's non-blocking stream -- stream-ordered,
    nothing allocated or waited for per call.  The first product widens x into the plan's buffer (plan-less kernel), the
    later ones read the float32 x directly in the copy pass, tier 0's windows and tier 1's gathers; every one of them equals
    the host entry's result bit for bit, with and without float32 products.
    """
    import ctypes as C
    import torch
    from csr_amd import CSR
    from csr_amd._lib import lib, check
    from csr_amd.kernels import hip as K
    rng = np.random.default_rng(21)
    lens = np.minimum((rng.pareto(0.8, 30000) * 4).astype(np.int64), 20000)
    m = _random_csr(rng, 30000, 600000, lens, dtype=np.float32 if vals == 'f4' else np.float64, sort=True)
    A = CSR(m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, m.values, _cast=False)
    x = rng.uniform(-1, 1, size=m.ncols).astype(np.float32)
    h = K.to_handle(A)
    try:
        want = K.mult_vec(h, x)                      # host entry (csrk_spmv_f32x): first product on the handle
        want2 = K.mult_vec(h, x)                     # ... and a planned one
        assert np.array_equal(want, want2) or np.allclose(want, want2, rtol=0, atol=1e-12 * np.abs(want).max())
        dev = torch.device('cuda', 0)
        dx = torch.from_numpy(x).to(dev)
        dy = torch.full((m.nrows,), float('nan'), dtype=torch.float64, device=dev)
        st = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        for _ in range(3):
            check(lib.csrk_spmv_f32x_device(K._live(h), dx.data_ptr(), dy.data_ptr(), C.c_void_p(st.cuda_stream)))
        st.synchronize()
        assert np.array_equal(dy.cpu().numpy(), want2)
    finally:
        K.release_handle(h)


def test_float32_matrix_keeps_float32_streams(split_mode):
    """
    A float32 matrix's plan stores float32 values in the tier-0 and light streams (6 and 8 bytes per entry instead of 10 and
    12) and widens them in the kernels: the product with a float64 vector is bit for bit the product of the same matrix
    with its values widened beforehand, and the plan is smaller.
    """
    if 'forced_split' not in split_mode:
        pytest.skip('x of this small matrix fits in L2: no split unless forced')
    import ctypes as C
    from csr_amd._lib import lib, check
    from csr_amd.kernels import hip as K
    from csr_amd import CSR
    rng = np.random.default_rng(2026)
    lens = rng.integers(0, 25, size=5000)
    lens[[3, 1500, 4100]] = [6000, 900, 2300]
    m = _random_csr(rng, 5000, 50000, lens, dtype=np.float32, sort=True)
    x = rng.uniform(-1, 1, size=m.ncols)
    ys, bytes_tier0, bytes_light = [], [], []
    for vals in (m.values, m.values.astype(np.float64)):
        h = K.to_handle(CSR(m.nrows, m.ncols, m.nnz, m.rowptrs, m.colinds, vals, _cast=False))
        try:
            K.mult_vec(h, x)
            ys.append(K.mult_vec(h, x).copy())             # (the planned product)
            st = (C.c_int64 * 34)()
            check(lib.csrk_spmv_plan_stats(h.H, st, 34))
            assert st[10] > 0                              # tier 0 holds entries
            bytes_tier0.append(st[29])
            bytes_light.append(st[31])
        finally:
            K.release_handle(h)
    assert np.array_equal(ys[0].view(np.int64), ys[1].view(np.int64))
    assert bytes_tier0[0] < 0.75 * bytes_tier0[1]
    if 'nostream' not in split_mode:
        assert 0 < bytes_light[0] < 0.8 * bytes_light[1]
