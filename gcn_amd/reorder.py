"""Host-side vertex reorderers (preprocessing), bound from libgcnspmm.so.

numpy in, numpy out.  All integer vectors are bit-exact with the reference
(order_deg.cu, order_rcm.cu, order_gorder.cu + unitheap.cu, renumber.cu); see
gcn_amd/csrc/reorder.cpp for the file:line map.  Internal functions return
``rank[old] = new``; the C-ABI renumber entry points return ``vomp[new] = old``
and rewrite the CSR in place (renumber.cu:90-95,184-189,520).
"""
import ctypes

import numpy as np

from . import _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _csr(rowptr, col):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    n = rowptr.shape[0] - 1
    return rowptr, col, n, int(col.shape[0])


def order_deg(rowptr, col, which="total", desc=True):
    """rank[old]=new by degree (total = in+out | out | in), ties by node id (order_deg.cu:8-56)."""
    rowptr, col, n, nnz = _csr(rowptr, col)
    out = np.empty(n, dtype=np.int64)
    w = {"total": 0, "out": 1, "in": 2}[which]
    _lib.check(_lib.load().gcn_order_deg(_p(rowptr), _p(col), n, nnz, w, int(bool(desc)), _p(out)),
               "gcn_order_deg")
    return out


def order_rcm(rowptr, col, directed=True):
    """Reverse Cuthill-McKee rank (order_rcm.cu:15-33)."""
    rowptr, col, n, nnz = _csr(rowptr, col)
    out = np.empty(n, dtype=np.int64)
    _lib.check(_lib.load().gcn_order_rcm(_p(rowptr), _p(col), n, nnz, int(bool(directed)), _p(out)),
               "gcn_order_rcm")
    return out


def order_gorder(rowptr, col, window=3):
    """RCM∘Gorder rank (order_gorder.cu:13-31); window 3 is what the C ABI uses (renumber.cu:176)."""
    rowptr, col, n, nnz = _csr(rowptr, col)
    out = np.empty(n, dtype=np.int64)
    _lib.check(_lib.load().gcn_order_gorder(_p(rowptr), _p(col), n, nnz, int(window), _p(out)),
               "gcn_order_gorder")
    return out


def apply_rank(rowptr, col, vals, rank):
    """CSR in the new numbering (columns sorted, values carried) and vomp[new]=old."""
    rowptr, col, n, nnz = _csr(rowptr, col)
    rowptr, col = rowptr.copy(), col.copy()
    vals = np.array(vals, dtype=np.float32, copy=True)
    rank = np.ascontiguousarray(rank, dtype=np.int64)
    vomp = np.empty(n, dtype=np.int32)
    _lib.check(_lib.load().gcn_csr_apply_rank(_p(rowptr), _p(col), _p(vals), n, nnz, _p(rank), _p(vomp)),
               "gcn_csr_apply_rank")
    return rowptr, col, vals, vomp


def _renumber(name, rowptr, col, vals, vomp=None):
    rowptr, col, n, nnz = _csr(rowptr, col)
    rowptr, col = rowptr.copy(), col.copy()
    vals = np.array(vals, dtype=np.float32, copy=True)
    vomp = np.arange(n, dtype=np.int32) if vomp is None else np.array(vomp, dtype=np.int32, copy=True)
    getattr(_lib.load(), name)(_p(rowptr), _p(col), _p(vals), _p(vomp), n, n, nnz)
    return rowptr, col, vals, vomp


def dfs(rowptr, col, vals):
    """renumber.so:dfs (renumber.cu:23-155) → (rowptr, col, vals, vomp)."""
    return _renumber("dfs", rowptr, col, vals)


def gorder(rowptr, col, vals):
    """renumber.so:gorder (renumber.cu:157-230)."""
    return _renumber("gorder", rowptr, col, vals)


def rabbit(rowptr, col, vals):
    """renumber.so:rabbit (renumber.cu:319-522)."""
    return _renumber("rabbit", rowptr, col, vals)


def perm_apply(rowptr, col, vals, vomp):
    """renumber.so:perm_apply (renumber.cu:233-318): apply a given vomp[new]=old."""
    return _renumber("perm_apply", rowptr, col, vals, vomp)


def order_deg_device(rowptr, col, which="total", desc=True):
    """order_deg on the GPU: torch tensors in (any device), rank[old]=new out on the
    same device.  The key (degree, node id) is a strict total order (order_deg.cu:8-13), so the result
    is bit-identical to the host version whatever the sort algorithm."""
    import torch
    n = rowptr.numel() - 1
    if rowptr.is_cuda:                      # the library's device kernels (reorder_device.hip)
        rp, ci = rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()
        rank = torch.empty(n, dtype=torch.int32, device=rowptr.device)
        _lib.check(_lib.load().gcn_order_deg_device(
            ctypes.c_void_p(rp.data_ptr()), ctypes.c_void_p(ci.data_ptr()), n, int(ci.numel()),
            {"total": 0, "out": 1, "in": 2}[which], int(bool(desc)), ctypes.c_void_p(rank.data_ptr()),
            ctypes.c_void_p(torch.cuda.current_stream(rowptr.device).cuda_stream)), "gcn_order_deg_device")
        return rank.to(torch.int64)
    out_deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    in_deg = torch.bincount(col.to(torch.int64), minlength=n)
    deg = {"total": out_deg + in_deg, "out": out_deg, "in": in_deg}[which]
    ids = torch.arange(n, dtype=torch.int64, device=deg.device)
    key = (deg.max() - deg if desc else deg) * n + ids          # primary: degree, secondary: id ascending
    order = torch.argsort(key)
    rank = torch.empty(n, dtype=torch.int64, device=deg.device)
    rank[order] = ids
    return rank


def order_rcm_device(rowptr, col, return_levels=False):
    """order_rcm on the GPU (reorder_device.hip): CUDA int32 CSR tensors in, int64
    rank[old]=new on the same device out — the same integers as `order_rcm(..., directed=False)`
    (and `directed=True` when the pattern is symmetric, as GCN adjacencies are)."""
    import torch
    if not rowptr.is_cuda:
        raise _lib.GcnAmdError("order_rcm_device needs CUDA/HIP tensors (the host version is order_rcm)")
    n = rowptr.numel() - 1
    rp, ci = rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()
    rank = torch.empty(n, dtype=torch.int32, device=rowptr.device)
    levels = ctypes.c_int32(0)
    _lib.check(_lib.load().gcn_order_rcm_device(
        ctypes.c_void_p(rp.data_ptr()), ctypes.c_void_p(ci.data_ptr()), n, int(ci.numel()),
        ctypes.c_void_p(rank.data_ptr()), ctypes.cast(ctypes.byref(levels), ctypes.c_void_p),
        ctypes.c_void_p(torch.cuda.current_stream(rowptr.device).cuda_stream)), "gcn_order_rcm_device")
    rank = rank.to(torch.int64)
    return (rank, int(levels.value)) if return_levels else rank


def apply_rank_device(rowptr, col, vals, rank):
    """CSR rewrite under rank[old]=new on the GPU (renumber.cu:190-217 semantics: rows and columns
    relabelled, every row's columns ascending, values carried along).
    → (rowptr', col', vals', vomp) with vomp[new]=old, all on the input device."""
    import torch
    if not rowptr.is_cuda:
        raise _lib.GcnAmdError("apply_rank_device needs CUDA/HIP tensors (the host version is apply_rank)")
    n, nnz = rowptr.numel() - 1, int(col.numel())
    rp, ci = rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()
    va, rk = vals.to(torch.float32).contiguous(), rank.to(torch.int32).contiguous()
    d = rowptr.device
    o_rp = torch.empty(n + 1, dtype=torch.int32, device=d)
    o_ci = torch.empty(nnz, dtype=torch.int32, device=d)
    o_va = torch.empty(nnz, dtype=torch.float32, device=d)
    vomp = torch.empty(n, dtype=torch.int32, device=d)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(_lib.load().gcn_csr_apply_rank_device(p(rp), p(ci), p(va), p(rk), n, nnz, p(o_rp), p(o_ci), p(o_va),
                                                     p(vomp), ctypes.c_void_p(torch.cuda.current_stream(d).cuda_stream)),
               "gcn_csr_apply_rank_device")
    return o_rp, o_ci, o_va, vomp


def order_rabbit(rowptr, col, return_communities=False):
    """The serial Rabbit of the reference (renumber.cu:319-520) without the CSR rewrite: rank[old]=new (int64; the
    inverse of the vomp the `rabbit` symbol returns) and, on request, the top-level vertex of every vertex."""
    rowptr, col, n, nnz = _csr(rowptr, col)
    vomp = np.empty(n, dtype=np.int32)
    comm = np.empty(n, dtype=np.int32) if return_communities else None
    _lib.check(_lib.load().gcn_order_rabbit(_p(rowptr), _p(col), n, nnz, _p(vomp), _p(comm) if comm is not None else None),
               "gcn_order_rabbit")
    rank = np.empty(n, dtype=np.int64)
    rank[vomp] = np.arange(n)
    return (rank, comm) if return_communities else rank


def order_rabbit_device(rowptr, col, return_communities=False, return_stats=False):
    """Rabbit on the GPU — parallel incremental aggregation (Arai et al. 2016, the algorithm renumber.cu:328-330 names;
    csrc/rabbit_device.hip).  CUDA int32 CSR of a SYMMETRIC pattern in, int64 rank[old]=new on the same device out.
    Not the serial code's integers (use `rabbit` / `order_rabbit` for those) and not bit-reproducible between runs;
    its communities reach the serial version's modularity to a few percent.
    → rank [, communities (top-level vertex per vertex)] [, dict(communities, passes, retried, left_top_level)]"""
    import torch
    if not rowptr.is_cuda:
        raise _lib.GcnAmdError("order_rabbit_device needs CUDA/HIP tensors (the host version is order_rabbit)")
    n = rowptr.numel() - 1
    rp, ci = rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()
    rank = torch.empty(n, dtype=torch.int32, device=rowptr.device)
    comm = torch.empty(n, dtype=torch.int32, device=rowptr.device) if return_communities else None
    stats = (ctypes.c_int64 * 8)()
    with torch.cuda.device(rowptr.device):
        _lib.check(_lib.load().gcn_order_rabbit_device(
            ctypes.c_void_p(rp.data_ptr()), ctypes.c_void_p(ci.data_ptr()), n, int(ci.numel()),
            ctypes.c_void_p(rank.data_ptr()), ctypes.c_void_p(comm.data_ptr()) if comm is not None else None,
            ctypes.cast(stats, ctypes.c_void_p), ctypes.c_void_p(torch.cuda.current_stream(rowptr.device).cuda_stream)),
            "gcn_order_rabbit_device")
    out = [rank.to(torch.int64)]
    if return_communities:
        out.append(comm.to(torch.int64))
    if return_stats:
        out.append(dict(communities=int(stats[0]), passes=int(stats[1]), retried=int(stats[2]), left_top_level=int(stats[3]),
                        guard_trips=[int(stats[4 + g]) for g in range(4)]))
    return out[0] if len(out) == 1 else tuple(out)


def modularity(rowptr, col, communities):
    """Newman modularity Q of a partition of an undirected graph given as a symmetric CSR pattern (self-loops ignored,
    unit weights — what Rabbit maximises, renumber.cu:454-470): Σ_c [ in_c / 2m − (deg_c / 2m)² ] with in_c the stored
    entries inside community c and deg_c the sum of its vertices' degrees.  torch tensors on any device → float"""
    import torch
    n = rowptr.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n, device=col.device), (rowptr[1:] - rowptr[:-1]).long())
    cols = col.long()
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    c = communities.to(device=col.device, dtype=torch.int64)
    two_m = float(rows.numel())
    if two_m == 0:
        return 0.0
    cr = c[rows]
    inside = torch.bincount(cr[cr == c[cols]], minlength=n).double()
    deg = torch.bincount(cr, minlength=n).double()
    return float((inside / two_m - (deg / two_m) ** 2).sum())


def order_communities_device(rowptr, col, max_rounds=24, min_merge_frac=0.002, return_levels=False):
    """A GPU community ordering in the spirit of Rabbit (incremental modularity aggregation,
    renumber.cu:319-522 / Arai et al. 2016) — SURVEY §8f.4's "parallel Rabbit", opt-in: it is NOT the
    reference's serial merge order and does not reproduce its integers (use `rabbit` for those).

    Rounds of parallel aggregation on the device: every community u picks the neighbouring community v
    with the largest modularity gain  w_uv − deg_u·deg_v / 2m  (> 0, ties to the smaller id) and merges
    into it when (deg_u, u) < (deg_v, v) — smaller into larger — and v is not itself merging this round
    (stars, no chains); the graph is contracted (sort + reduce by key) and the round repeats.  The final order keeps every community of every round contiguous (a stable sort per round,
    last round most significant) — the leaf order of the merge dendrogram.
    → rank[old]=new (int64, on the input device)."""
    import torch
    dev = rowptr.device
    n = rowptr.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), (rowptr[1:] - rowptr[:-1]).long())
    cols = col.long()
    keep = rows != cols
    u, v = rows[keep], cols[keep]
    del rows, cols, keep
    # symmetrise the pattern, unit weights, no duplicates
    key = torch.unique(torch.cat([u * n + v, v * n + u]))
    u, v = key // n, key % n
    w = torch.ones_like(u, dtype=torch.float64)
    del key
    deg = torch.bincount(u, minlength=n).double()               # community degree (sum over members)
    two_m = float(deg.sum())
    if two_m == 0:
        r = torch.arange(n, device=dev, dtype=torch.int64)
        return (r, 0) if return_levels else r
    comm = torch.arange(n, device=dev, dtype=torch.int64)       # community of every ORIGINAL vertex
    levels = []
    ncomm = n
    for _round in range(max_rounds):
        if u.numel() == 0:
            break
        gain = w - deg[u] * deg[v] / two_m
        # allowed direction: smaller (degree, id) into larger
        ok = (gain > 0) & ((deg[u] < deg[v]) | ((deg[u] == deg[v]) & (u < v)))
        if not bool(ok.any()):
            break
        gu, gv, gg = u[ok], v[ok], gain[ok]
        # per u: the largest gain, ties to the smaller v  (sort by (u, -gain, v), take the first of each u)
        order = torch.argsort(gv, stable=True)
        gu, gv, gg = gu[order], gv[order], gg[order]
        order = torch.argsort(-gg, stable=True)
        gu, gv, gg = gu[order], gv[order], gg[order]
        order = torch.argsort(gu, stable=True)
        gu, gv = gu[order], gv[order]
        first = torch.ones_like(gu, dtype=torch.bool)
        first[1:] = gu[1:] != gu[:-1]
        ident = torch.arange(ncomm, device=dev, dtype=torch.int64)
        parent = ident.clone()
        parent[gu[first]] = gv[first]
        # stars only: a community that is itself merging this round takes no members (its would-be members
        # wait for the next round, when the weights to the merged community are known) — chains of
        # merges would otherwise sweep across community borders in a single round
        moving = parent != ident
        parent = torch.where(moving & moving[parent], ident, parent)
        merged = int((parent != ident).sum())
        if merged == 0:
            break
        # compact the surviving roots to 0..ncomm'-1
        roots, newid = torch.unique(parent, return_inverse=True)
        comm = newid[comm]
        levels.append(comm.clone())
        deg = torch.zeros(roots.numel(), device=dev, dtype=torch.float64).index_add_(0, newid, deg)
        ncomm = int(roots.numel())
        # contract the edge list
        u, v = newid[u], newid[v]
        keep = u != v
        key = u[keep] * ncomm + v[keep]
        wk = w[keep]
        key, inv = torch.unique(key, return_inverse=True)
        w = torch.zeros(key.numel(), device=dev, dtype=torch.float64).index_add_(0, inv, wk)
        u, v = key // ncomm, key % ncomm
        if merged < min_merge_frac * n:
            break
    order = torch.arange(n, device=dev, dtype=torch.int64)
    for lab in levels:                                          # least significant (first round) first
        order = order[torch.argsort(lab[order], stable=True)]
    rank = torch.empty(n, device=dev, dtype=torch.int64)
    rank[order] = torch.arange(n, device=dev, dtype=torch.int64)
    return (rank, len(levels)) if return_levels else rank
