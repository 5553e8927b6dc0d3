"""Product reorderers (C++ in libgcnspmm.so, bound through the C ABI): bit-exact against
(i) the golden vectors recorded from the reference's compiled code, (ii) the reference build
itself (oracle/_ref, present in the build container) on random graphs, (iii) the independent
Python oracle; plus the invariants the reference asserts (renumber.cu:120-149,283-313)."""
import ctypes
import glob
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import gcn_amd
from gcn_amd import reorder
from util import GOLDEN, ROOT, sym_norm_graph

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import reorder_oracle as ro  # noqa: E402

CASES = sorted(glob.glob(os.path.join(GOLDEN, "reorder_*.npz")))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "renumber_ref.so")
REF_ORD = os.path.join(ROOT, "oracle", "_ref", "libref_orders.so")


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_bit_exact_vs_reference_golden(path):
    g = np.load(path)
    rp, ci, va = g["rowptr"], g["col"], g["val"]
    for which in ("total", "out", "in"):
        for desc in (False, True):
            assert np.array_equal(reorder.order_deg(rp, ci, which, desc),
                                  g[f"deg_{which}_{'desc' if desc else 'asc'}"])
    assert np.array_equal(reorder.order_rcm(rp, ci, True), g["rcm_directed"])
    assert np.array_equal(reorder.order_rcm(rp, ci, False), g["rcm_undirected"])
    for w in (1, 3, 5):
        assert np.array_equal(reorder.order_gorder(rp, ci, w), g[f"gorder_w{w}"])
    for fn in ("dfs", "gorder", "rabbit"):
        out = getattr(reorder, fn)(rp, ci, va)
        for key, arr in zip(("rowptr", "col", "val", "vomp"), out):
            assert arr.dtype == g[f"{fn}_{key}"].dtype
            assert np.array_equal(arr, g[f"{fn}_{key}"]), f"{fn} {key}"
    out = reorder.perm_apply(rp, ci, va, g["perm_apply_in_vomp"])
    for key, arr in zip(("rowptr", "col", "val"), out[:3]):
        assert np.array_equal(arr, g[f"perm_apply_{key}"])
    # apply_rank(rank) == the C-ABI gorder rewrite
    rank = reorder.order_gorder(rp, ci, 3)
    rp2, ci2, va2, vomp = reorder.apply_rank(rp, ci, va, rank)
    assert np.array_equal(rp2, g["gorder_rowptr"]) and np.array_equal(ci2, g["gorder_col"])
    assert np.array_equal(va2, g["gorder_val"]) and np.array_equal(vomp, g["gorder_vomp"])


def _graphs():
    yield "sym_2k", sym_norm_graph(2000, 20000, seed=10)
    yield "sym_sparse", sym_norm_graph(3000, 4000, seed=11)
    rng = np.random.default_rng(12)                      # skewed, asymmetric, with self-loops
    n, e = 1500, 12000
    u = np.minimum(n - 1, (n * rng.random(e) ** 2.5).astype(int)); v = rng.integers(0, n, e)
    A = sp.coo_matrix((np.ones(e), (u, v)), shape=(n, n)).tocsr(); A = (A + sp.eye(n)).tocsr()
    A.data[:] = rng.random(A.nnz) + 0.1; A.sort_indices()
    yield "directed_skewed", (A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32))
    # R-MAT, products-shaped degree skew (hub rows, many equal dQ candidates): the case where Rabbit's
    # "first maximum in key order" tie rule and the equal-degree introsort permutation really matter
    from gcn_amd import graphgen
    rowptr, col, val, _ = graphgen.make_graph("products", device="cpu", seed=3, scale=0.006)
    yield "rmat_products_shaped", (rowptr.numpy(), col.numpy(), val.numpy())


@pytest.mark.skipif(not (os.path.exists(REF_SO) and os.path.exists(REF_ORD)),
                    reason="oracle/_ref not built (needs /root/reference; build container only)")
@pytest.mark.parametrize("name,graph", list(_graphs()), ids=[n for n, _ in _graphs()])
def test_bit_exact_vs_reference_build_on_random_graphs(name, graph, capfd):
    rp, ci, va = graph
    n, nnz = len(rp) - 1, len(ci)
    ref, refo = ctypes.CDLL(REF_SO), ctypes.CDLL(REF_ORD)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    for fn in ("dfs", "gorder", "rabbit"):
        a = [rp.copy(), ci.copy(), va.copy(), np.arange(n, dtype=np.int32)]
        getattr(ref, fn)(p(a[0]), p(a[1]), p(a[2]), p(a[3]), n, n, nnz)
        b = getattr(reorder, fn)(rp, ci, va)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), fn
    for which, wname in enumerate(("total", "out", "in")):
        for desc in (0, 1):
            o = np.zeros(n, np.int64)
            refo.ref_order_deg(p(rp), p(ci), n, nnz, which, desc, p(o))
            assert np.array_equal(o, reorder.order_deg(rp, ci, wname, bool(desc)))
    for d in (0, 1):
        o = np.zeros(n, np.int64)
        refo.ref_order_rcm(p(rp), p(ci), n, nnz, d, p(o))
        assert np.array_equal(o, reorder.order_rcm(rp, ci, bool(d)))
    for w in (2, 3, 7):
        o = np.zeros(n, np.int64)
        refo.ref_complete_gorder(p(rp), p(ci), n, nnz, w, p(o))
        assert np.array_equal(o, reorder.order_gorder(rp, ci, w))
    capfd.readouterr()      # swallow the reference's Info()/printf chatter


def test_product_vs_python_oracle_on_a_fresh_graph():
    rp, ci, va = sym_norm_graph(300, 1500, seed=33)
    assert np.array_equal(reorder.order_rcm(rp, ci), ro.order_rcm(rp, ci))
    assert np.array_equal(reorder.order_gorder(rp, ci, 3), ro.complete_gorder(rp, ci, 3))
    for fn in ("dfs", "gorder", "rabbit"):
        for x, y in zip(getattr(reorder, fn)(rp, ci, va), getattr(ro, fn)(rp, ci, va)):
            assert np.array_equal(x, y), fn


@pytest.mark.parametrize("fn", ["dfs", "gorder", "rabbit"])
def test_renumbering_invariants(fn):
    """what the reference asserts inline: vomp is a bijection, the rewritten CSR equals
    A[vomp][:, vomp] exactly, columns sorted ascending within each row"""
    rp, ci, va = sym_norm_graph(1200, 9000, seed=5)
    n = len(rp) - 1
    rp2, ci2, va2, vomp = getattr(reorder, fn)(rp, ci, va)
    assert sorted(vomp.tolist()) == list(range(n))
    A = sp.csr_matrix((va, ci, rp), shape=(n, n))
    B = sp.csr_matrix((va2, ci2, rp2), shape=(n, n))
    P = A[vomp][:, vomp].tocsr(); P.sort_indices()
    assert np.array_equal(P.indptr, rp2) and np.array_equal(P.indices, ci2) and np.array_equal(P.data, va2)
    for r in range(n):
        seg = ci2[rp2[r]:rp2[r + 1]]
        assert np.all(seg[1:] > seg[:-1])
    assert B.nnz == A.nnz


def test_gorder_with_isolated_vertices():
    """Vertices with no edge at all (no self-loop either): RCM ranks them last, so they sit
    above the heap range and Gorder appends them at the end (order_gorder.cu:42-43,78) — the
    case UnitHeap::ReConstruct (unitheap.cu:31-37) silently relies on.  Product == Python oracle
    (== the reference build when present)."""
    rp = np.array([0, 0, 2, 4, 6, 6, 8, 10], np.int32)      # vertices 0 and 4 are isolated
    ci = np.array([2, 3, 1, 3, 1, 2, 6, 6, 5, 5], np.int32)
    n, nnz = len(rp) - 1, len(ci)
    got = reorder.order_gorder(rp, ci, 3)
    assert np.array_equal(got, ro.complete_gorder(rp, ci, 3))
    assert sorted(got.tolist()) == list(range(n))
    if os.path.exists(REF_ORD):
        o = np.zeros(n, np.int64)
        ctypes.CDLL(REF_ORD).ref_complete_gorder(ctypes.c_void_p(rp.ctypes.data), ctypes.c_void_p(ci.ctypes.data),
                                                 n, nnz, 3, ctypes.c_void_p(o.ctypes.data))
        assert np.array_equal(got, o)


def test_order_deg_device_variant_equals_host():
    """the torch (device-capable) degree ordering is bit-identical to the C++ host one"""
    import torch
    rp, ci, va = sym_norm_graph(1500, 9000, seed=8)
    rng = np.random.default_rng(1)
    ci2 = ci.copy(); ci2[rng.integers(0, len(ci), 500)] = rng.integers(0, 1500, 500)   # make in != out degrees
    for which in ("total", "out", "in"):
        for desc in (True, False):
            got = reorder.order_deg_device(torch.from_numpy(rp), torch.from_numpy(ci2), which, desc).numpy()
            assert np.array_equal(got, reorder.order_deg(rp, ci2, which, desc)), (which, desc)


def test_gpu_style_community_ordering_is_a_permutation_that_groups_planted_communities():
    """order_communities_device (opt-in, NOT the reference's integers): pure tensor code, so its logic
    is checked here on the CPU: a valid, deterministic permutation under which most neighbours in the
    new order come from the same planted community, and which rewrites the CSR consistently"""
    import torch
    from gcn_amd import graphgen
    n = 6000
    rowptr, col, val, _ = graphgen.make_sbm(n, device="cpu", seed=7)
    rank, levels = reorder.order_communities_device(rowptr, col, return_levels=True)
    assert levels >= 3 and sorted(rank.tolist()) == list(range(n))
    assert torch.equal(rank, reorder.order_communities_device(rowptr, col))
    gen = torch.Generator(); gen.manual_seed(7 + 1000)
    perm = torch.randperm(n, generator=gen)                                  # make_sbm's relabelling
    block = torch.empty(n, dtype=torch.long); block[perm] = torch.arange(n) // 512
    order = torch.argsort(rank)
    same = float((block[order][1:] == block[order][:-1]).float().mean())
    assert same > 0.7, same                                                  # random order: ~0.08
    rp2, ci2, va2, vomp = reorder.apply_rank(rowptr.numpy(), col.numpy(), val.numpy(), rank.numpy())
    A = sp.csr_matrix((val.numpy(), col.numpy(), rowptr.numpy()), shape=(n, n))
    B = sp.csr_matrix((va2, ci2, rp2), shape=(n, n))
    assert abs(B - A[vomp][:, vomp]).max() == 0
    # degenerate inputs
    e = torch.zeros(5, dtype=torch.int32)
    assert reorder.order_communities_device(torch.zeros(5, dtype=torch.int32), e[:0]).tolist() == [0, 1, 2, 3]


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_order_rabbit_without_the_rewrite_gives_the_reference_vomp_and_its_communities(path):
    """gcn_order_rabbit: the serial Rabbit as a rank (no CSR rewrite) — the inverse of the vomp recorded from the
    reference's compiled code — plus the top-level vertex of every vertex: communities are contiguous in the new
    order, their top-level vertex is a member, and the modularity helper agrees with a dense evaluation"""
    import torch
    g = np.load(path)
    rp, ci = g["rowptr"], g["col"]
    n = len(rp) - 1
    rank, comm = reorder.order_rabbit(rp, ci, return_communities=True)
    assert np.array_equal(rank[g["rabbit_vomp"]], np.arange(n))              # bit-exact with the reference's order
    assert np.array_equal(reorder.order_rabbit(rp, ci), rank)
    in_new_order = comm[g["rabbit_vomp"]]
    changes = int((np.diff(in_new_order) != 0).sum())
    assert changes == len(np.unique(comm)) - 1                               # every community one contiguous run
    assert np.array_equal(comm[np.unique(comm)], np.unique(comm))            # a top-level vertex belongs to itself
    # modularity against a dense evaluation on the symmetrised, loop-free pattern
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n)); A = ((A + A.T) > 0).astype(np.float64).tolil()
    A.setdiag(0); A = A.tocsr()
    two_m = A.sum()
    if two_m > 0:
        deg = np.asarray(A.sum(1)).ravel()
        q = 0.0
        for c in np.unique(comm):
            mem = np.flatnonzero(comm == c)
            q += A[mem][:, mem].sum() / two_m - (deg[mem].sum() / two_m) ** 2
        As = A.tocsr(); As.sort_indices()
        got = reorder.modularity(torch.from_numpy(As.indptr.astype(np.int32)), torch.from_numpy(As.indices.astype(np.int32)),
                                 torch.from_numpy(comm))
        assert abs(got - q) <= 1e-9
