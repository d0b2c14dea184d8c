"""
Synthetic CSR workloads for bench.py and the parity tests.

torch is used here only as plumbing: it generates the matrices directly in HBM (a
10M x 10M, nnz = 2e8 matrix is 2.44 GB; generating it on the host would dominate the run).
Nothing in this module computes a product.

Power-law generator (`powerlaw_csr`), partition independent: every quantity is a pure
function of (seed, global row / global entry index), so rank r of N builds exactly the row
range it owns of the SAME matrix a single GPU would build.

* row degrees: d_i proportional to rank_i^(-alpha) for a seeded random permutation of ranks,
  clipped to [0, max_degree], rescaled (fixed point on the scale factor) and integerised
  by largest remainder so that sum(d) == nnz exactly.  alpha = 1.1 gives a mean of 20 at
  nnz/nrows = 20, a maximum of ~1e6, about half the rows with <= 1 entry and ~6 % empty.
* columns: Zipf(1.0)-popular ranks (inverse CDF of 1/r on [1, ncols+1)) mapped through a
  seeded random permutation of the column space; distinct within a row (duplicates are
  redrawn uniformly until none are left) and ascending within a row.
* values and x: U(-1, 1) from a counter-based hash (splitmix64 finaliser).
"""
import math

import torch

_M1 = -4658895280553007687      # 0xBF58476D1CE4E5B9 as int64
_M2 = -7723592293110705685      # 0x94D049BB133111EB as int64
_GOLD = -7046029254386353131    # 0x9E3779B97F4A7C15 as int64


def _lsr(z, k):
    "logical shift right on int64 tensors"
    return (z >> k) & ((1 << (64 - k)) - 1)


def mix64(z):
    "splitmix64 finaliser, wrapping int64 arithmetic"
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def hash_uniform(idx, seed, stream=0):
    "U[0,1) float64 from (global index, seed, stream): 53 random bits"
    z = mix64(idx * _GOLD + (seed * 1000003 + stream * 7919 + 12345))
    return _lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def powerlaw_degrees(nrows, nnz, alpha=1.1, max_degree=1_000_000, seed=20261003, device='cpu'):
    "int64[nrows] row degrees, sum == nnz exactly (identical on every rank)"
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ranks = torch.randperm(nrows, generator=g, device=device).to(torch.float64) + 1.0
    w = ranks.pow(-alpha)
    max_degree = float(min(max_degree, nnz))
    scale = nnz / float(w.sum())
    for _ in range(60):                      # fixed point: clipped mass is redistributed
        d = torch.clamp(w * scale, max=max_degree)
        tot = float(d.sum())
        if abs(tot - nnz) <= 0.25:
            break
        free = float((w * scale)[w * scale < max_degree].sum())
        if free <= 0:
            break
        scale *= 1.0 + (nnz - tot) / free
    d = torch.clamp(w * scale, max=max_degree)
    base = torch.floor(d)
    rem = int(nnz - int(base.sum().item()))
    deg = base.to(torch.int64)
    if rem > 0:
        frac = d - base
        top = torch.topk(frac, min(rem, nrows)).indices
        deg[top] += 1
        rem -= min(rem, nrows)
    elif rem < 0:
        frac = torch.where(deg > 0, d - base, torch.full_like(d, 2.0))
        low = torch.topk(-frac, -rem).indices
        deg[low] -= 1
        rem = 0
    assert rem == 0 and int(deg.sum().item()) == nnz, (rem, int(deg.sum().item()), nnz)
    return deg


def balanced_row_ranges(rowptr, world):
    """
    nnz-balanced contiguous row ranges: boundaries searchsorted(rowptrs, g * nnz / world) -- the
    primitive the reference's _shard_rows uses (csr/csr.py:609).  Returns world+1 row indices.
    """
    nrows = rowptr.numel() - 1
    nnz = int(rowptr[-1].item())
    tg = torch.tensor([(nnz * g) // world for g in range(1, world)], dtype=rowptr.dtype, device=rowptr.device)
    cuts = torch.searchsorted(rowptr, tg, right=False).clamp(0, nrows).tolist() if world > 1 else []
    bounds = [0] + [int(c) for c in cuts] + [nrows]
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def powerlaw_csr(nrows, ncols, nnz, alpha=1.1, max_degree=1_000_000, seed=20261003, device='cpu',
                 rank=0, world=1, values=True):
    """
    Build rows [r0, r1) of the power-law matrix owned by `rank` of `world`.

    Returns dict(nrows_total, ncols, nnz_total, row_begin, row_end, rowptrs (int32 local,
    rebased to 0; int64 if the local nnz needs it), colinds (int32), values (float64 or None)).
    """
    # a row never holds more than ncols/8 entries, so uniform redraws of duplicates converge
    deg = powerlaw_degrees(nrows, nnz, alpha, max(1, min(max_degree, ncols // 8)), seed, device)
    rowptr = torch.zeros(nrows + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    bounds = balanced_row_ranges(rowptr, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    e0, e1 = int(rowptr[r0].item()), int(rowptr[r1].item())
    n_loc = e1 - e0
    ldeg = deg[r0:r1]
    rows = torch.repeat_interleave(torch.arange(r1 - r0, device=device, dtype=torch.int64), ldeg)
    gidx = torch.arange(e0, e1, device=device, dtype=torch.int64)

    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    colperm = torch.randperm(ncols, generator=g, device=device)
    u = hash_uniform(gidx, seed, 1)
    rnk = torch.floor(torch.exp(u * math.log(ncols + 1.0))).to(torch.int64).clamp_(1, ncols) - 1
    cols = colperm[rnk]
    del u, rnk
    key = rows * ncols + cols
    del cols
    for rnd in range(64):
        key, _ = torch.sort(key)
        dup = torch.zeros(n_loc, dtype=torch.bool, device=device)
        if n_loc > 1:
            dup[1:] = key[1:] == key[:-1]
        ndup = int(dup.sum().item())
        if ndup == 0:
            break
        didx = dup.nonzero(as_tuple=True)[0]
        # redraw the duplicate slots uniformly; the slot index (post-sort) + round keeps this a
        # pure function of the seed
        fresh = torch.floor(hash_uniform(didx + e0, seed, 100 + rnd) * ncols).to(torch.int64).clamp_(0, ncols - 1)
        key[didx] = (key[didx] // ncols) * ncols + fresh
        del dup, didx, fresh
    else:
        raise RuntimeError('could not make columns distinct')
    colinds = (key % ncols).to(torch.int32)
    del key, rows
    lrp = rowptr[r0:r1 + 1] - e0
    lrp = lrp.to(torch.int32) if n_loc <= 2**31 - 1 else lrp
    vals = None
    if values:
        vals = hash_uniform(gidx, seed, 2) * 2.0 - 1.0
    return dict(nrows_total=nrows, ncols=ncols, nnz_total=nnz, row_begin=r0, row_end=r1,
                rowptrs=lrp.contiguous(), colinds=colinds.contiguous(), values=vals,
                bounds=bounds, alpha=alpha, seed=seed)


def dense_vector(n, seed=20261003, device='cpu', stream=3):
    "x ~ U(-1, 1), float64"
    idx = torch.arange(n, device=device, dtype=torch.int64)
    return hash_uniform(idx, seed, stream) * 2.0 - 1.0


def uniform_csr(nrows, ncols, nnz, seed=20261003, device='cpu'):
    "configs[0]-style matrix: nnz unique uniform coordinates, rows sorted, values N(0,1)-ish U(-1,1)"
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    total = nrows * ncols
    coords = torch.randint(0, total, (int(nnz * 1.05) + 16,), generator=g, device=device, dtype=torch.int64)
    coords = torch.unique(coords)[:nnz]
    assert coords.numel() == nnz
    rows = coords // ncols
    cols = (coords % ncols).to(torch.int32)
    rowptr = torch.zeros(nrows + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=nrows), 0)
    vals = hash_uniform(torch.arange(nnz, device=device, dtype=torch.int64), seed, 2) * 2.0 - 1.0
    return dict(nrows_total=nrows, ncols=ncols, nnz_total=nnz, row_begin=0, row_end=nrows,
                rowptrs=rowptr.to(torch.int32), colinds=cols, values=vals)


ML25M_SHAPE = (162_541, 59_047, 25_000_095)


def movielens_like(device='cpu', values=True):
    """
    BASELINE.json configs[4]: a MovieLens-25M-shaped ratings matrix -- 162 541 users x
    59 047 items, nnz 25 000 095, user activity and item popularity both power-law, values in {0.5, 1.0, .. 5.0}.
    Returns the powerlaw_csr dict plus nrows.
    """
    nr, nc, nnz = ML25M_SHAPE
    m = powerlaw_csr(nr, nc, nnz, device=device, alpha=0.9, max_degree=7000, values=values)
    if values:
        m['values'] = (torch.floor((m['values'] + 1.0) * 5.0).clamp_(0, 9) + 1.0) * 0.5
    m['nrows'] = nr
    return m
