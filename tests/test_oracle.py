"""The oracle itself, pinned: oracle/spmm_oracle.c against scipy fp64 and against the Python
reference's recorded torch.spmm outputs; oracle/reorder_oracle.py against the golden integer
vectors produced by the reference's own compiled code (tests/golden/, oracle/make_golden.py)."""
import glob
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from util import GOLDEN, ROOT, oracle_spmm, random_csr, rel_err

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import reorder_oracle as ro  # noqa: E402

REORDER_CASES = sorted(glob.glob(os.path.join(GOLDEN, "reorder_*.npz")))


def test_spmm_oracle_matches_scipy_fp64():
    m, n, k = 800, 1000, 37
    rowptr, col, val = random_csr(m, n, 20000, seed=1, empty_rows=0.1, long_rows=[(3, 900)])
    B = np.random.default_rng(0).standard_normal((n, k)).astype(np.float32)
    A = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(m, n))
    ref = (A @ B.astype(np.float64))
    assert rel_err(oracle_spmm(rowptr, col, val, B, fp64=True), ref) <= 1e-7
    assert rel_err(oracle_spmm(rowptr, col, val, B, fp64=False), ref) <= 1e-5


@pytest.mark.parametrize("name", ["gcn1_tiny", "gcn1_cora_shaped"])
def test_spmm_oracle_matches_python_reference_golden(name):
    """golden = outputs of torch.spmm(adj, support) at pygcn/gcn1.py:53 run by the reference"""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = int(g["n"])
    A = sp.coo_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(n, n)).tocsr()
    A.sort_indices()
    for sup, agg in (("support1", "agg1"), ("support2", "agg2")):
        C = oracle_spmm(A.indptr, A.indices, A.data, g[sup])
        assert rel_err(C, g[agg]) <= 1e-5


def test_golden_adjacency_is_the_reference_normalisation():
    """Â in the fixture = D^-1/2 (A+I) D^-1/2 (utils.py:78-90): symmetric, unit self-loop mass"""
    g = np.load(os.path.join(GOLDEN, "gcn1_cora_shaped.npz"))
    n = int(g["n"])
    A = sp.coo_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(n, n)).tocsr()
    assert A.nnz == 12623 and abs(A - A.T).max() < 1e-7
    deg = np.diff(A.indptr)
    assert np.allclose(A.diagonal(), 1.0 / deg, rtol=1e-6)


@pytest.mark.parametrize("path", REORDER_CASES, ids=[os.path.basename(p)[8:-4] for p in REORDER_CASES])
def test_reorder_oracle_bit_exact_vs_reference_golden(path):
    g = np.load(path)
    rp, ci, va = g["rowptr"], g["col"], g["val"]
    for which in ("total", "out", "in"):
        for desc in (False, True):
            got = ro.order_deg(rp, ci, which, desc)
            assert np.array_equal(got, g[f"deg_{which}_{'desc' if desc else 'asc'}"])
    assert np.array_equal(ro.order_rcm(rp, ci, True), g["rcm_directed"])
    assert np.array_equal(ro.order_rcm(rp, ci, False), g["rcm_undirected"])
    for w in (1, 3, 5):
        assert np.array_equal(ro.complete_gorder(rp, ci, w), g[f"gorder_w{w}"]), f"gorder window {w}"
    for fn in ("dfs", "gorder", "rabbit"):
        out = getattr(ro, fn)(rp, ci, va)
        for key, arr in zip(("rowptr", "col", "val", "vomp"), out):
            assert np.array_equal(arr, g[f"{fn}_{key}"]), f"{fn} {key}"
    out = ro.perm_apply(rp, ci, va, g["perm_apply_in_vomp"])
    for key, arr in zip(("rowptr", "col", "val"), out):
        assert np.array_equal(arr, g[f"perm_apply_{key}"])


def test_libstdcxx_sort_restatement_sorts_and_is_deterministic():
    rng = np.random.default_rng(0)
    for n in (0, 1, 5, 16, 17, 100, 1000):
        keys = rng.integers(0, 7, n).tolist()
        a = list(range(n))
        ro._libstdcxx_sort(a, lambda x, y: keys[x] < keys[y])
        assert sorted(a) == list(range(n))
        assert all(keys[a[i]] <= keys[a[i + 1]] for i in range(n - 1))
    # adversarial input that drives introsort into its heapsort fallback still sorts
    n = 3000
    a = list(range(n // 2)) + list(range(n // 2))[::-1]
    keys = a[:]
    idx = list(range(n))
    ro._libstdcxx_sort(idx, lambda x, y: keys[x] < keys[y])
    assert all(keys[idx[i]] <= keys[idx[i + 1]] for i in range(n - 1))
