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


# ---- Rabbit on the device: parallel incremental aggregation (csrc/rabbit_device.hip; renumber.cu:328-330 / Arai'16) ----
# No bit parity with the serial host Rabbit is expected (the header says so): what is bounded is the QUALITY.
def _sym_loopfree(A):
    A = ((A + A.T) != 0).astype(np.float32).tocsr()
    A.setdiag(0); A.eliminate_zeros(); A.sort_indices()
    return A


def test_rabbit_device_on_small_graphs_is_a_permutation_with_sound_communities():
    """structure-only checks on graphs with known answers: two cliques joined by a bridge -> two communities; a graph
    without edges / isolated vertices -> everyone top-level; every returned community is contiguous in the order and
    named after one of its members; stats add up"""
    C = np.ones((12, 12)); A = sp.block_diag([C, C]).tolil(); A[11, 12] = A[12, 11] = 1
    A = _sym_loopfree(A.tocsr())
    rank, comm, stats = reorder.order_rabbit_device(_d(A.indptr.astype(np.int32)), _d(A.indices.astype(np.int32)),
                                                    return_communities=True, return_stats=True)
    rank, comm = rank.cpu().numpy(), comm.cpu().numpy()
    assert sorted(rank.tolist()) == list(range(24)) and stats["communities"] == 2
    assert stats["guard_trips"] == [0, 0, 0, 0]          # every bounded loop ended by itself (a trip is an error: GCN_ERR_INTERNAL)
    assert len(set(comm[:12])) == 1 and len(set(comm[12:])) == 1 and comm[0] != comm[12]
    assert set(rank[:12]) in ({*range(12)}, {*range(12, 24)})                    # each clique one contiguous run
    # no edges at all, and edges among a few vertices only
    for n, edges in ((50, []), (50, [(1, 2), (2, 3), (1, 3), (10, 11)])):
        M = sp.lil_matrix((n, n)); [M.__setitem__((u, v), 1) for u, v in edges]
        M = _sym_loopfree(M.tocsr())
        rank, comm, stats = reorder.order_rabbit_device(_d(M.indptr.astype(np.int32)), _d(M.indices.astype(np.int32)),
                                                        return_communities=True, return_stats=True)
        assert sorted(rank.cpu().tolist()) == list(range(n))
        c = comm.cpu().numpy()
        assert stats["communities"] == len(np.unique(c)) == n - (3 if edges else 0)
        if edges:
            assert c[1] == c[2] == c[3] and c[10] == c[11] and c[0] == 0
    # the golden graphs: permutation, contiguous communities named after a member, modularity not below the singletons'
    for path in CASES:
        g = np.load(path)
        A = _sym_loopfree(sp.csr_matrix((np.ones(len(g["col"])), g["col"], g["rowptr"])))
        n = A.shape[0]
        rp, ci = _d(A.indptr.astype(np.int32)), _d(A.indices.astype(np.int32))
        rank, comm = reorder.order_rabbit_device(rp, ci, return_communities=True)
        r, c = rank.cpu().numpy(), comm.cpu().numpy()
        assert sorted(r.tolist()) == list(range(n))
        order = np.argsort(r)
        assert int((np.diff(c[order]) != 0).sum()) == len(np.unique(c)) - 1     # one contiguous run per community
        assert np.array_equal(c[np.unique(c)], np.unique(c))
        if A.nnz:
            q = reorder.modularity(rp, ci, comm)
            q0 = reorder.modularity(rp, ci, torch.arange(n))
            assert q >= q0 - 1e-12, (path, q, q0)


@pytest.mark.parametrize("n", [60000])
def test_rabbit_device_matches_the_serial_rabbits_quality_at_a_fraction_of_its_time(n):
    """planted-partition graph (communities of 512, labels shuffled): the parallel Rabbit reaches the serial
    (reference-exact) Rabbit's modularity to 2 %, exposes the communities to the row-panel kernels (window coverage
    >= 0.6: enough to turn the MFMA panels on), finds about as many communities as were planted, and takes less than a
    tenth of the host time (measured: 17 ms against 2.8 s).  Twice: runs differ in which merges race, not in quality."""
    import time
    import gcn_amd
    dev = torch.device("cuda:0")
    rowptr, col, val, n = graphgen.make_sbm(n, device=dev, seed=7)
    planted = -(-n // 512)
    reorder.order_rabbit_device(rowptr, col)                                     # (first call: allocations, code load)
    t0 = time.perf_counter()
    rank_h, comm_h = reorder.order_rabbit(rowptr.cpu().numpy(), col.cpu().numpy(), return_communities=True)
    t_host = time.perf_counter() - t0
    q_host = reorder.modularity(rowptr, col, torch.from_numpy(comm_h))
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rank, comm, stats = reorder.order_rabbit_device(rowptr, col, return_communities=True, return_stats=True)
        torch.cuda.synchronize()
        t_dev = time.perf_counter() - t0
        assert torch.equal(torch.sort(rank).values, torch.arange(n, device=dev))
        q_dev = reorder.modularity(rowptr, col, comm)
        assert q_dev >= 0.98 * q_host, (q_dev, q_host)
        assert 0.8 * planted <= stats["communities"] <= 1.3 * planted, stats
        assert stats["guard_trips"] == [0, 0, 0, 0], stats
        print(f"rabbit n={n}: device {t_dev * 1e3:.1f} ms, host {t_host * 1e3:.1f} ms ({t_host / max(t_dev, 1e-9):.0f} x)")   # a figure, not a bar
        rp, ci, va, _ = reorder.apply_rank_device(rowptr, col, val, rank)
        adj = gcn_amd.CsrAdjacency(rp, ci, va, (n, n), symmetric=True, panels="auto")
        assert adj.panel_coverage >= 0.6 and adj.panel_rows > 0 and adj.dense_panels > 0, adj.panel_coverage
    # and the product on the renumbered matrix is the product (P·Â·Pᵀ)(P·B) = P·(Â·B)
    B = graphgen.random_features(n, 64, seed=3, device=dev)
    base = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True).matmul_raw(B)
    inv = torch.empty_like(rank); inv[rank] = torch.arange(n, device=dev)
    got = adj.matmul_raw(B[inv])
    assert float((got - base[inv]).abs().max() / base.abs().max()) <= 1e-5
