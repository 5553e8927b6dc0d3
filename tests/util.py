"""Shared helpers for the tests: the oracle bindings and small graph builders.
The oracle is test infrastructure (oracle/README.md); nothing here is product code."""
import ctypes
import os

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = ctypes.CDLL(os.path.join(ROOT, "oracle", "libspmm_oracle.so"))
    return _oracle


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def oracle_spmm(rowptr, col, val, B, fp64=True):
    """CPU oracle C = A @ B (fp64 accumulate by default), numpy arrays in/out."""
    rowptr = np.ascontiguousarray(rowptr, np.int32); col = np.ascontiguousarray(col, np.int32)
    val = np.ascontiguousarray(val, np.float32); B = np.ascontiguousarray(B, np.float32)
    m, k = len(rowptr) - 1, B.shape[1]
    C = np.empty((m, k), np.float32)
    fn = oracle().spmm_oracle_f64 if fp64 else oracle().spmm_oracle_f32
    fn(_p(rowptr), _p(col), _p(val), _p(B), _p(C), m, k)
    return C


def rel_err(C, Cref):
    """max|C - C*| / max|C*|  — the parity metric of BASELINE.md §3 (tolerance 1e-5)."""
    denom = float(np.abs(Cref).max())
    return float(np.abs(C.astype(np.float64) - Cref.astype(np.float64)).max()) / (denom if denom > 0 else 1.0)


def random_csr(m, n, nnz_target, seed, empty_rows=0.0, long_rows=(), sorted_cols=True):
    """Random CSR with optional empty rows and a few very long rows (hub rows)."""
    rng = np.random.default_rng(seed)
    lens = rng.poisson(max(nnz_target / max(m, 1), 0.01), m).astype(np.int64)
    if empty_rows > 0:
        lens[rng.random(m) < empty_rows] = 0
    for r, L in long_rows:
        lens[r] = L
    lens = np.minimum(lens, n)
    rowptr = np.zeros(m + 1, np.int64); rowptr[1:] = np.cumsum(lens)
    col = np.empty(rowptr[-1], np.int32)
    for r in range(m):
        c = rng.choice(n, size=lens[r], replace=False) if lens[r] <= n // 2 else rng.permutation(n)[:lens[r]]
        col[rowptr[r]:rowptr[r + 1]] = np.sort(c) if sorted_cols else c
    val = (rng.standard_normal(rowptr[-1]) * 0.5).astype(np.float32)
    return rowptr.astype(np.int32), col, val


def sym_norm_graph(n, e, seed):
    """Â = D^-1/2 (A+I) D^-1/2 for a random undirected graph (fp64 → fp32 like utils.py:78-90)."""
    rng = np.random.default_rng(seed)
    u, v = rng.integers(0, n, e), rng.integers(0, n, e)
    A = sp.coo_matrix((np.ones(e), (u, v)), shape=(n, n)); A = (A + A.T).tocsr()
    A.setdiag(0); A.eliminate_zeros(); A.data[:] = 1.0
    A = (A + sp.eye(n)).tocsr()
    d = np.asarray(A.sum(1)).ravel() ** -0.5
    A = (sp.diags(d) @ A @ sp.diags(d)).tocsr(); A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)


def sampled_rows_oracle_err(rowptr_dev, col_dev, val_dev, B_dev, C_dev, rows):
    """rel. error of C_dev[rows] against the fp64 C oracle on a compacted copy of the sampled rows: only the rows
    of B the sample references travel to the host (full-size configs: B has tens of GB).  All *_dev are torch
    tensors on one device; rows a sorted numpy int64 array.  → (rel_err, entries_checked)"""
    import torch
    dev = C_dev.device
    r = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
    rp = rowptr_dev.long()
    start, lens = rp[r], rp[r + 1] - rp[r]
    seg = torch.repeat_interleave(torch.arange(len(rows), device=dev), lens)
    first = torch.cumsum(lens, 0) - lens
    e = start[seg] + (torch.arange(int(lens.sum()), device=dev) - first[seg])
    cols = col_dev[e].long()
    uniq, inv = torch.unique(cols, return_inverse=True)
    sub_rp = np.zeros(len(rows) + 1, np.int32)
    sub_rp[1:] = np.cumsum(lens.cpu().numpy())
    Cref = oracle_spmm(sub_rp, inv.to(torch.int32).cpu().numpy(), val_dev[e].cpu().numpy(), B_dev[uniq].cpu().numpy())
    return rel_err(C_dev[r].cpu().numpy(), Cref), int(e.numel())
