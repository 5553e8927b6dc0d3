"""Device reorderers (gcn_amd/csrc/reorder_device.hip): the GPU versions of order_deg,
order_rcm and the CSR rewrite must return the SAME integers as the host versions (which are pinned
bit-exactly to the reference, tests/test_reorder.py) — including the golden vectors recorded from the
reference's own compiled code."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from gcn_amd import graphgen, reorder
from util import GOLDEN, sym_norm_graph

pytestmark = pytest.mark.gpu
CASES = sorted(glob.glob(os.path.join(GOLDEN, "reorder_*.npz")))


def _d(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _graphs():
    rng = np.random.default_rng(3)
    out = {}
    # many components: isolated vertices (self-loop only and none at all), pairs, small cliques, one big part
    n = 4000
    A = sp.random(n, n, density=0.0008, random_state=5, format="csr")
    A = ((A + A.T) != 0).astype(np.float32).tolil()
    A[:600, :] = 0; A[:, :600] = 0                       # 600 isolated vertices ...
    for i in range(0, 300):
        A[i, i] = 1                                      # ... half of them with a self-loop
    for i in range(300, 400, 2):
        A[i, i + 1] = A[i + 1, i] = 1                    # pairs
    A = A.tocsr(); A.sort_indices()
    out["components"] = A
    # high diameter: a path with a few chords, vertex labels shuffled
    n = 3000
    p = rng.permutation(n)
    r = np.concatenate([p[:-1], p[1:], p[::100][:-1], p[::100][1:]])
    c = np.concatenate([p[1:], p[:-1], p[::100][1:], p[::100][:-1]])
    P = sp.coo_matrix((np.ones(len(r), np.float32), (r, c)), shape=(n, n)).tocsr(); P.data[:] = 1; P.sort_indices()
    out["path"] = P
    # star + ring (hub with every vertex adjacent)
    n = 2000
    r = np.concatenate([np.zeros(n - 1, int), np.arange(1, n), np.arange(1, n - 1), np.arange(2, n)])
    c = np.concatenate([np.arange(1, n), np.zeros(n - 1, int), np.arange(2, n), np.arange(1, n - 1)])
    S = sp.coo_matrix((np.ones(len(r), np.float32), (r, c)), shape=(n, n)).tocsr(); S.data[:] = 1; S.sort_indices()
    out["star_ring"] = S
    # GCN-style normalised adjacency with self-loops
    rp, ci, va = sym_norm_graph(5000, 40000, seed=8)
    out["gcn"] = sp.csr_matrix((va, ci, rp), shape=(5000, 5000))
    # asymmetric pattern: the device version symmetrises (= host directed=False)
    D = sp.random(1500, 1500, density=0.004, random_state=9, format="csr"); D.data[:] = 1; D.sort_indices()
    out["asymmetric"] = D.astype(np.float32)
    return out


@pytest.mark.parametrize("name", ["components", "path", "star_ring", "gcn", "asymmetric"])
def test_order_rcm_device_equals_host(name):
    A = _graphs()[name]
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    want = reorder.order_rcm(rp, ci, directed=False)
    got, levels = reorder.order_rcm_device(_d(rp), _d(ci), return_levels=True)
    assert np.array_equal(got.cpu().numpy(), want)
    assert levels >= 1
    if name != "asymmetric":                              # symmetric pattern: the directed variant agrees too
        assert np.array_equal(got.cpu().numpy(), reorder.order_rcm(rp, ci, directed=True))
    if name == "path":
        assert levels > 20                                # really went level by level through a deep BFS


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_device_orderings_vs_reference_golden(path):
    g = np.load(path)
    rp, ci = g["rowptr"], g["col"]
    assert np.array_equal(reorder.order_rcm_device(_d(rp), _d(ci)).cpu().numpy(), g["rcm_undirected"])
    for which in ("total", "out", "in"):
        for desc in (False, True):
            got = reorder.order_deg_device(_d(rp), _d(ci), which, desc).cpu().numpy()
            assert np.array_equal(got, g[f"deg_{which}_{'desc' if desc else 'asc'}"])


def test_apply_rank_device_equals_host_and_rejects_non_permutations():
    rp, ci, va = sym_norm_graph(6000, 60000, seed=4)
    rank = reorder.order_rcm(rp, ci, directed=True)
    w_rp, w_ci, w_va, w_vomp = reorder.apply_rank(rp, ci, va, rank)
    g_rp, g_ci, g_va, g_vomp = reorder.apply_rank_device(_d(rp), _d(ci), _d(va), _d(rank))
    assert np.array_equal(g_rp.cpu().numpy(), w_rp) and np.array_equal(g_ci.cpu().numpy(), w_ci)
    assert np.array_equal(g_va.cpu().numpy(), w_va) and np.array_equal(g_vomp.cpu().numpy(), w_vomp)
    bad = rank.copy(); bad[0] = bad[1]
    with pytest.raises(Exception):
        reorder.apply_rank_device(_d(rp), _d(ci), _d(va), _d(bad))


def test_device_rcm_pipeline_on_a_large_graph_matches_host():
    """products-shaped R-MAT at 1/20 scale (122 k vertices, 6 M non-zeros, thousands of components)"""
    rowptr, col, val, n = graphgen.make_graph("products", device="cuda:0", seed=3, scale=0.05)
    rank = reorder.order_rcm_device(rowptr, col)
    want = reorder.order_rcm(rowptr.cpu().numpy(), col.cpu().numpy(), directed=True)
    assert np.array_equal(rank.cpu().numpy(), want)
    rp2, ci2, va2, vomp = reorder.apply_rank_device(rowptr, col, val, rank)
    h = reorder.apply_rank(rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy(), want)
    assert np.array_equal(rp2.cpu().numpy(), h[0]) and np.array_equal(ci2.cpu().numpy(), h[1])
    assert np.array_equal(va2.cpu().numpy(), h[2]) and np.array_equal(vomp.cpu().numpy(), h[3])
