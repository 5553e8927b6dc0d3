"""Deterministic synthetic graphs shaped like the BASELINE configs.

No dataset exists offline, so every config runs on a seeded
stand-in with the same (n, nnz, degree skew).  The generator is R-MAT with a
random vertex relabelling (so that "no reorder" really has no locality), de-duplicated,
symmetrised, self-loops added, and normalised the way the reference does it:
Â = D^-1/2 (A + I) D^-1/2 computed in fp64 then cast to fp32
(pygcn/gcnio/util/utils.py:78-90, :243-250).

All tensor work is torch, on whatever device is asked for (the GPU for the
100 M-nnz cases, the CPU for tests); results are deterministic per device type.
"""
import math

import torch

# (n, undirected edges before self-loops, rmat (a,b,c,d))
# R-MAT skew is picked so that the maximum degree lands near the real dataset's
# (Reddit: mean 492, max ≈ 21.6 k): with the classic (.57,.19,.19,.05) vertex 0 would
# be adjacent to every other vertex at this density and de-duplication would remove
# most samples, which is an easier (cache-friendlier) problem than the real graph.
SHAPES = {
    "cora":     dict(n=2485, edges=5069, abcd=(0.25, 0.25, 0.25, 0.25)),
    "reddit":   dict(n=232965, edges=57307946, abcd=(0.37, 0.25, 0.25, 0.13)),
    "products": dict(n=2449029, edges=61859140, abcd=(0.45, 0.22, 0.22, 0.11)),
    # the other graphs of the reference's run.sh sweep (GraphSAINT datasets; vertex / undirected-edge /
    # feature / class counts as published by Zeng et al., "GraphSAINT", ICLR 2020, Table 2, and for
    # pubmed by Planetoid) — shapes only, for tools/run_sweep.sh; none of them is a BASELINE config
    "pubmed":   dict(n=19717, edges=44324, abcd=(0.45, 0.22, 0.22, 0.11), nfeat=500, nclass=3),
    "flickr":   dict(n=89250, edges=899756, abcd=(0.45, 0.22, 0.22, 0.11), nfeat=500, nclass=7),
    "ppi":      dict(n=14755, edges=225270, abcd=(0.45, 0.22, 0.22, 0.11), nfeat=50, nclass=121),
    "yelp":     dict(n=716847, edges=6977410, abcd=(0.45, 0.22, 0.22, 0.11), nfeat=300, nclass=100),
    "amazon":   dict(n=1598960, edges=132169734, abcd=(0.45, 0.22, 0.22, 0.11), nfeat=200, nclass=107),
}
SHAPES["reddit"].update(nfeat=602, nclass=41)
SHAPES["products"].update(nfeat=100, nclass=47)
SHAPES["cora"].update(nfeat=1433, nclass=7)


def _rmat_pairs(n, count, abcd, gen, device):
    """`count` (u, v) samples from an R-MAT distribution over a 2^s x 2^s grid, folded to n."""
    a, b, c, _ = abcd
    scale = max(1, math.ceil(math.log2(max(n, 2))))
    u = torch.zeros(count, dtype=torch.int64, device=device)
    v = torch.zeros(count, dtype=torch.int64, device=device)
    for _lvl in range(scale):
        r = torch.rand(count, generator=gen, device=device)
        # quadrants: [0,a) -> (0,0), [a,a+b) -> (0,1), [a+b,a+b+c) -> (1,0), rest -> (1,1)
        ubit = (r >= a + b).to(torch.int64)
        vbit = (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)
        u = (u << 1) | ubit
        v = (v << 1) | vbit
    return u % n, v % n


def rmat_undirected_edges(n, edges, abcd=(0.57, 0.19, 0.19, 0.05), seed=0, device="cpu"):
    """Exactly `edges` distinct undirected edges {u<v} (as one int64 key u*n+v, sorted)."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    max_edges = n * (n - 1) // 2
    if edges > max_edges:
        raise ValueError("more edges requested than the simple graph can hold")
    keys = torch.empty(0, dtype=torch.int64, device=device)
    need = edges
    for _round in range(64):
        batch = int(need * 1.15) + 1024
        u, v = _rmat_pairs(n, batch, abcd, gen, device)
        lo, hi = torch.minimum(u, v), torch.maximum(u, v)
        ok = lo != hi
        keys = torch.unique(torch.cat([keys, lo[ok] * n + hi[ok]]))
        if keys.numel() >= edges:
            break
        need = edges - keys.numel()
    else:  # pragma: no cover
        raise RuntimeError("R-MAT generator did not reach the requested edge count")
    if keys.numel() > edges:   # drop a seeded random subset of the surplus
        perm = torch.randperm(keys.numel(), generator=gen, device=device)[:edges]
        keys = keys[perm.sort().values]
    return keys


def normalized_adjacency(n, edge_keys, relabel_seed=None):
    """CSR of Â = D^-1/2 (A+I) D^-1/2 (int32 rowptr/col, fp32 val) from undirected edge keys.

    Returns (rowptr, col, val) on the device of `edge_keys`; columns sorted within rows.
    """
    device = edge_keys.device
    u, v = edge_keys // n, edge_keys % n
    if relabel_seed is not None:
        gen = torch.Generator(device=device)
        gen.manual_seed(relabel_seed)
        perm = torch.randperm(n, generator=gen, device=device)
        u, v = perm[u], perm[v]
    loops = torch.arange(n, dtype=torch.int64, device=device)
    rows = torch.cat([u, v, loops])
    cols = torch.cat([v, u, loops])
    del u, v
    key = torch.sort(rows * n + cols).values
    del rows, cols
    rows, cols = key // n, key % n
    del key
    deg = torch.bincount(rows, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(deg, 0)
    dinv = deg.to(torch.float64).pow(-0.5)
    val = (dinv[rows] * dinv[cols]).to(torch.float32)
    return rowptr.to(torch.int32), cols.to(torch.int32), val


def make_graph(name, device="cpu", seed=1, scale=1.0):
    """(rowptr, col, val, n) of the named BASELINE-shaped graph.  `scale` < 1 shrinks n and
    the edge count together (same mean degree) for tests."""
    spec = SHAPES[name]
    n = max(16, int(spec["n"] * scale))
    edges = max(n, int(spec["edges"] * scale))
    edges = min(edges, n * (n - 1) // 2)
    keys = rmat_undirected_edges(n, edges, spec["abcd"], seed=seed, device=device)
    rowptr, col, val = normalized_adjacency(n, keys, relabel_seed=seed + 1000)
    return rowptr, col, val, n


def make_rmat(scale, edge_factor=16, abcd=(0.57, 0.19, 0.19, 0.05), device="cpu", seed=5, relabel=True):
    """Graph500-style R-MAT: n = 2^scale vertices, edge_factor*n generated directed edges,
    symmetrised, de-duplicated, self-loops added, normalised (BASELINE config 5: scale 24,
    edge factor 16, (.57,.19,.19,.05)).  Vertex labels are randomly permuted (as Graph500
    prescribes) unless relabel=False.  → (rowptr, col, val, n)"""
    device = torch.device(device)
    n = 1 << scale
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    keys = torch.empty(0, dtype=torch.int64, device=device)
    total, batch = edge_factor * n, 1 << 26
    done = 0
    while done < total:                         # in batches: 2^28 samples at scale 24
        cnt = min(batch, total - done)
        u, v = _rmat_pairs(n, cnt, abcd, gen, device)
        lo, hi = torch.minimum(u, v), torch.maximum(u, v)
        ok = lo != hi
        keys = torch.unique(torch.cat([keys, lo[ok] * n + hi[ok]]))
        done += cnt
        del u, v, lo, hi, ok
    rowptr, col, val = normalized_adjacency(n, keys, relabel_seed=(seed + 1000) if relabel else None)
    return rowptr, col, val, n


def make_rmat_row_block(n, directed_edges, world, rank, abcd=(0.57, 0.19, 0.19, 0.05), device="cpu", seed=4,
                        batch=1 << 26):
    """Row block `rank` of `world` (contiguous, equal row counts) of a graph too large to materialise
    whole on one device (BASELINE config 4, papers100M-shaped: n = 111 059 956, 1.616 G directed
    samples): R-MAT samples folded to n, vertex labels randomly permuted, symmetrised, de-duplicated,
    self-loops added, Â = D^-1/2 (A+I) D^-1/2.  The sample stream is regenerated once per row block
    (same seed → same graph) so that only one block's entries are ever resident; the degrees of ALL
    vertices (needed for the values) come out of those passes.
    → (rowptr[int32, rows+1], col[int32], val[fp32], n, row_lo, row_hi, deg[int64, n]) with GLOBAL column
    indices; deg = stored entries per row of the WHOLE graph (self-loop included), so Â = diag(u)·(A+I)·diag(u)
    with u = deg^-1/2 and the global non-zero count is deg.sum()."""
    device = torch.device(device)
    rows_per = (n + world - 1) // world
    pgen = torch.Generator(device=device)
    pgen.manual_seed(seed + 1000)
    perm = torch.randperm(n, generator=pgen, device=device)
    deg = torch.zeros(n, dtype=torch.int64, device=device)
    mine = None
    for b in range(world):
        lo, hi = b * rows_per, min(n, (b + 1) * rows_per)
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)                                  # the same sample stream for every block
        parts, done = [], 0
        while done < directed_edges:
            cnt = min(batch, directed_edges - done)
            u, v = _rmat_pairs(n, cnt, abcd, gen, device)
            u, v = perm[u], perm[v]
            ok = u != v
            u, v = u[ok], v[ok]
            for r, c in ((u, v), (v, u)):                      # symmetrise; keep the entries of this block's rows
                m = (r >= lo) & (r < hi)
                parts.append((r[m] - lo) * n + c[m])
            done += cnt
            del u, v, ok
            if len(parts) >= 16:                               # bound the backlog: fold it into one unique'd part
                parts = [torch.unique(torch.cat(parts))]
        keys = torch.unique(torch.cat(parts))
        del parts
        deg[lo:hi] = torch.bincount(keys // n, minlength=hi - lo) + 1          # + the self-loop
        if b == rank:
            mine = keys
        del keys
    lo, hi = rank * rows_per, min(n, (rank + 1) * rows_per)
    loops = torch.arange(lo, hi, dtype=torch.int64, device=device)
    keys = torch.sort(torch.cat([mine, (loops - lo) * n + loops])).values
    del mine, loops
    rows, cols = keys // n, keys % n
    del keys
    rowptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=hi - lo), 0)
    dinv = deg.to(torch.float64).pow(-0.5)
    val = (dinv[rows + lo] * dinv[cols]).to(torch.float32)
    return rowptr.to(torch.int32), cols.to(torch.int32), val, n, lo, hi, deg


def value_factor_from_degrees(deg):
    """u [n] fp32 with fl32(Â[r, c]) = u[r]·u[c] to within a few ulp: the square root of the stored diagonal
    fl32(deg^-1) — what gcn_spmm_plan_enable_slicing derives by itself when it sees a whole square matrix."""
    dinv = deg.to(torch.float64).pow(-0.5)
    return (dinv * dinv).to(torch.float32).sqrt()


def make_graph_row_block(name, world, rank, device="cpu", seed=1, scale=1.0, balance="nnz"):
    """Row block `rank` of the `world`-way contiguous, nnz-balanced row partition of make_graph(name, ...) —
    built WITHOUT materialising the whole CSR: the undirected edge keys (one int64 per edge) and the degree
    vector are the only whole-graph objects, the block's entries are selected and sorted on their own.
    → (local_rowptr[int32], col[int64, GLOBAL ids], val[fp32], n, bounds[np.int64, world+1], u[fp32, n],
       total_nnz); identical, entry for entry, to slicing make_graph's result (tests/test_graphgen.py)."""
    import numpy as np
    spec = SHAPES[name]
    n = max(16, int(spec["n"] * scale))
    edges = max(n, int(spec["edges"] * scale))
    edges = min(edges, n * (n - 1) // 2)
    keys = rmat_undirected_edges(n, edges, spec["abcd"], seed=seed, device=device)
    device = keys.device
    u, v = keys // n, keys % n
    del keys
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 1000)
    perm = torch.randperm(n, generator=gen, device=device)
    u, v = perm[u], perm[v]
    deg = torch.bincount(u, minlength=n) + torch.bincount(v, minlength=n) + 1
    rowptr_g = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr_g[1:] = torch.cumsum(deg, 0)
    total = int(rowptr_g[-1])
    if balance == "rows":
        bounds = np.array([(n * p) // world for p in range(world + 1)], dtype=np.int64)
    else:                                                     # as dist.partition_rows, on the device
        targets = torch.tensor([(total * p) // world for p in range(1, world)], dtype=torch.int64, device=device)
        cuts = torch.searchsorted(rowptr_g, targets, right=False).cpu().numpy() if world > 1 else np.zeros(0, np.int64)
        bounds = np.maximum.accumulate(np.minimum(np.concatenate([[0], cuts, [n]]).astype(np.int64), n))
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    parts = []
    for r, c in ((u, v), (v, u)):
        m = (r >= lo) & (r < hi)
        parts.append((r[m] - lo) * n + c[m])
    del u, v
    loops = torch.arange(lo, hi, dtype=torch.int64, device=device)
    parts.append((loops - lo) * n + loops)
    key = torch.sort(torch.cat(parts)).values
    del parts
    rows, cols = key // n, key % n
    del key
    local_rowptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=device)
    local_rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=hi - lo), 0)
    dinv = deg.to(torch.float64).pow(-0.5)
    val = (dinv[rows + lo] * dinv[cols]).to(torch.float32)
    return local_rowptr.to(torch.int32), cols, val, n, bounds, value_factor_from_degrees(deg), total


def random_features(n, k, seed=2, device="cpu"):
    """B ~ N(0,1) fp32 [n x k] (the reference standard-scales features, profiling_gcn.py:31-35)."""
    gen = torch.Generator(device=torch.device(device))
    gen.manual_seed(seed)
    return torch.randn((n, k), generator=gen, device=device, dtype=torch.float32)


def make_sbm(n, block=512, deg_in=200, deg_out=46, device="cpu", seed=7, relabel=True):
    """Planted-partition graph (communities of `block` vertices; every vertex draws `deg_in` partners
    inside its community and `deg_out` anywhere), symmetrised, de-duplicated, self-loops added,
    normalised, labels randomly permuted.  Unlike R-MAT it HAS the community structure the
    reference's reorderers (Rabbit, Gorder, RCM) are meant to recover.  → (rowptr, col, val, n)"""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    ids = torch.arange(n, dtype=torch.int64, device=device)
    base = (ids // block) * block
    size = torch.clamp(torch.full_like(ids, block), max=n - base)
    u_in = ids.repeat_interleave(deg_in)
    v_in = base.repeat_interleave(deg_in) + (torch.rand(n * deg_in, generator=gen, device=device)
                                              * size.repeat_interleave(deg_in)).long()
    u_out = ids.repeat_interleave(deg_out)
    v_out = torch.randint(0, n, (n * deg_out,), generator=gen, device=device, dtype=torch.int64)
    u, v = torch.cat([u_in, u_out]), torch.cat([v_in, v_out])
    lo, hi = torch.minimum(u, v), torch.maximum(u, v)
    keys = torch.unique((lo * n + hi)[lo != hi])
    rowptr, col, val = normalized_adjacency(n, keys, relabel_seed=(seed + 1000) if relabel else None)
    return rowptr, col, val, n


def community_sizes(n, communities, skew=1.0, min_size=64, seed=0):
    """`communities` heterogeneous sizes that sum to n: weights ~ rank^-skew (Zipf; skew 0 = equal sizes), every community
    at least min_size vertices.  Deterministic (numpy on the host: a few thousand numbers).  → np.int64 [communities]"""
    import numpy as np
    communities = int(max(1, min(communities, n // max(1, min_size))))
    w = np.arange(1, communities + 1, dtype=np.float64) ** -float(skew)
    np.random.default_rng(seed).shuffle(w)
    spare = n - communities * min_size
    sizes = min_size + np.floor(w / w.sum() * spare).astype(np.int64)
    sizes[: n - int(sizes.sum())] += 1                      # the rounding remainder, one vertex each
    assert int(sizes.sum()) == n
    return sizes


def make_dcsbm(n=232965, edges=57307946, communities=200, mixing=0.35, max_degree=20000, gamma=2.5, size_skew=1.0,
               device="cpu", seed=11, relabel=True, return_communities=False):
    """Degree-corrected planted-partition graph at the headline size: the stand-in for what the
    reference's datasets look like (run.sh:1-9: reddit, flickr, ppi, yelp, amazon — graphs WITH community structure, which is
    why gcn6 renumbers by default, gcn6.py:27-30,313-332).  Every vertex has a weight theta (truncated power law: exponent
    `gamma`, largest ≈ max_degree / mean degree times the mean) and lives in one of `communities` planted communities of
    heterogeneous size (community_sizes).  An edge sample picks u ~ theta over all vertices, then v ~ theta inside u's
    community with probability 1 - mixing and over all vertices otherwise; samples are de-duplicated and topped up to
    exactly `edges` distinct undirected edges, then symmetrised, + I, D^-1/2 (A+I) D^-1/2 like every other graph here,
    labels randomly permuted (relabel) so that "no reorder" has no locality.
    → (rowptr, col, val, n[, community of every vertex in the RETURNED numbering (int64)])
    The realised share of cross-community edges is a little above `mixing` where a hub saturates a small community."""
    import numpy as np
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    sizes = community_sizes(n, communities, size_skew, seed=seed)
    ncom = len(sizes)
    start_h = np.zeros(ncom + 1, np.int64)
    start_h[1:] = np.cumsum(sizes)
    start = torch.from_numpy(start_h).to(device)
    comm = torch.repeat_interleave(torch.arange(ncom, device=device), torch.from_numpy(sizes).to(device))   # planted order: by community
    # weights: theta = (1 - U)^(-1/(gamma-1)), truncated so that the largest expected degree is ~max_degree
    mean_deg = 2.0 * edges / n
    u01 = torch.rand(n, generator=gen, device=device, dtype=torch.float64)
    theta = (1.0 - u01).pow(-1.0 / (gamma - 1.0))
    cap = float(max_degree) / mean_deg * float(theta.mean())
    for _ in range(4):                                     # (truncation lowers the mean: re-aim the cap)
        t = torch.clamp(theta, max=cap)
        cap = float(max_degree) / mean_deg * float(t.mean())
    theta = torch.clamp(theta, max=cap)
    cum = torch.cumsum(theta, 0)
    total = float(cum[-1])
    cum0 = torch.cat([torch.zeros(1, dtype=torch.float64, device=device), cum])
    c_lo, c_hi = cum0[start[:-1]], cum0[start[1:]]         # theta mass in front of / up to the end of every community

    def draw(count):
        u = torch.searchsorted(cum, torch.rand(count, generator=gen, device=device, dtype=torch.float64) * total).clamp_(max=n - 1)
        inside = torch.rand(count, generator=gen, device=device) >= mixing
        cu = comm[u]
        r = torch.rand(count, generator=gen, device=device, dtype=torch.float64)
        x = torch.where(inside, c_lo[cu] + r * (c_hi[cu] - c_lo[cu]), r * total)
        v = torch.searchsorted(cum, x).clamp_(max=n - 1)
        v = torch.where(inside, torch.minimum(torch.maximum(v, start[cu]), start[cu + 1] - 1), v)   # (rounding at a community's edge)
        return u, v

    keys = torch.empty(0, dtype=torch.int64, device=device)
    need = edges
    for _round in range(64):
        batch = min(int(need * 1.2) + 1024, 1 << 26)
        u, v = draw(batch)
        lo, hi = torch.minimum(u, v), torch.maximum(u, v)
        ok = lo != hi
        keys = torch.unique(torch.cat([keys, lo[ok] * n + hi[ok]]))
        del u, v, lo, hi, ok
        if keys.numel() >= edges:
            break
        need = edges - keys.numel()
    else:  # pragma: no cover
        raise RuntimeError("DC-SBM generator did not reach the requested edge count")
    if keys.numel() > edges:
        perm = torch.randperm(keys.numel(), generator=gen, device=device)[:edges]
        keys = keys[perm.sort().values]
    rowptr, col, val = normalized_adjacency(n, keys, relabel_seed=(seed + 1000) if relabel else None)
    if not return_communities:
        return rowptr, col, val, n
    if relabel:                                            # the same permutation normalized_adjacency drew
        g2 = torch.Generator(device=device)
        g2.manual_seed(seed + 1000)
        perm = torch.randperm(n, generator=g2, device=device)
        out = torch.empty_like(comm)
        out[perm] = comm
        comm = out
    return rowptr, col, val, n, comm
