"""The synthetic stand-ins for the BASELINE graphs: deterministic, symmetric, normalised the
way the reference normalises (utils.py:78-90), with the requested size."""
import numpy as np
import scipy.sparse as sp
import torch

from gcn_amd import graphgen


def test_small_reddit_shaped_graph_is_valid_and_deterministic():
    rp, ci, va, n = graphgen.make_graph("reddit", device="cpu", seed=1, scale=0.01)
    rp2, ci2, va2, _ = graphgen.make_graph("reddit", device="cpu", seed=1, scale=0.01)
    assert torch.equal(rp, rp2) and torch.equal(ci, ci2) and torch.equal(va, va2)
    assert rp.dtype == torch.int32 and ci.dtype == torch.int32 and va.dtype == torch.float32
    nnz = int(ci.numel())
    edges = max(n, int(graphgen.SHAPES["reddit"]["edges"] * 0.01))
    assert nnz == 2 * edges + n                      # symmetrised + one self-loop per vertex
    A = sp.csr_matrix((va.numpy(), ci.numpy(), rp.numpy()), shape=(n, n))
    assert abs(A - A.T).max() == 0
    assert A.has_sorted_indices
    deg = np.diff(rp.numpy())
    # Â = D^-1/2 (A+I) D^-1/2 in fp64, cast to fp32
    d = deg.astype(np.float64) ** -0.5
    rows = np.repeat(np.arange(n), deg)
    assert np.array_equal(va.numpy(), (d[rows] * d[ci.numpy()]).astype(np.float32))
    assert np.all(A.diagonal() > 0)


def test_cora_shape_matches_survey_numbers():
    rp, ci, va, n = graphgen.make_graph("cora", device="cpu", seed=0)
    assert n == 2485 and int(ci.numel()) == 12623     # nnz = 2*5069 + 2485


def test_row_blocks_of_a_graph_too_large_to_hold_whole_tile_one_symmetric_normalised_matrix():
    """make_rmat_row_block (BASELINE config 4 stand-in): the row blocks, generated independently from
    the same seed, stack to one symmetric Â = D^-1/2 (A+I) D^-1/2 with sorted, distinct columns."""
    import numpy as np
    import scipy.sparse as sp
    n, samples, world = 3000, 40000, 4
    mats, seen = [], 0
    for r in range(world):
        rp, ci, va, nn, lo, hi, deg = graphgen.make_rmat_row_block(n, samples, world, r, device="cpu", seed=4, batch=1 << 13)
        assert torch.equal(deg[lo:hi], (rp[1:] - rp[:-1]).long())        # whole-graph degrees come with every block
        assert nn == n and lo == seen and rp.dtype == torch.int32 and ci.dtype == torch.int32 and va.dtype == torch.float32
        seen = hi
        rows = np.repeat(np.arange(hi - lo), np.diff(rp.numpy()))
        key = rows.astype(np.int64) * n + ci.numpy()
        assert np.all(np.diff(key) > 0)                       # sorted within rows, no duplicates
        mats.append(sp.csr_matrix((va.numpy(), ci.numpy(), rp.numpy()), shape=(hi - lo, n)))
    assert seen == n
    A = sp.vstack(mats).tocsr()
    assert abs(A - A.T).max() == 0.0 and np.all(A.diagonal() > 0)
    d = np.asarray((A != 0).sum(1)).ravel()
    coo = A.tocoo()
    assert np.abs(coo.data - 1.0 / np.sqrt(d[coo.row] * d[coo.col])).max() < 1e-7
    # deterministic
    again = graphgen.make_rmat_row_block(n, samples, world, 1, device="cpu", seed=4, batch=1 << 13)
    assert torch.equal(again[1], graphgen.make_rmat_row_block(n, samples, world, 1, device="cpu", seed=4, batch=1 << 13)[1])


def test_rank_local_row_blocks_equal_slices_of_the_whole_graph():
    """make_graph_row_block builds a rank's block of the nnz-balanced partition without the whole CSR; stacked,
    the blocks are make_graph's matrix entry for entry, the bounds are dist.partition_rows', and the value
    factor reproduces the values"""
    import numpy as np
    from gcn_amd.dist import partition_rows
    rp, ci, va, n = graphgen.make_graph("reddit", device="cpu", seed=1, scale=0.004)
    for world in (1, 3):
        want = partition_rows(rp.numpy(), world)
        for r in range(world):
            lrp, col, val, nn, bounds, u, total = graphgen.make_graph_row_block("reddit", world, r, device="cpu", seed=1, scale=0.004)
            assert nn == n and total == int(ci.numel()) and np.array_equal(bounds, want) and col.dtype == torch.int64
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            e0, e1 = int(rp[lo]), int(rp[hi])
            assert torch.equal(lrp.long(), rp[lo:hi + 1].long() - e0)
            assert torch.equal(col, ci[e0:e1].long()) and torch.equal(val, va[e0:e1])
            rows = torch.repeat_interleave(torch.arange(lo, hi), (lrp[1:] - lrp[:-1]).long())
            assert float(((u[rows] * u[col] - val).abs() / val).max()) < 4e-7


def test_dcsbm_generator_plants_communities_and_hits_the_requested_size():
    """graphgen.make_dcsbm (the structured stand-in of round 4): exact edge count, symmetric normalised adjacency with
    self-loops, skewed degrees, heterogeneous communities, a realised mixing near the one asked for, deterministic,
    and labels that really are shuffled"""
    import torch
    from gcn_amd import graphgen
    n, edges = 6000, 300000
    rp, ci, va, n2, comm = graphgen.make_dcsbm(n=n, edges=edges, communities=12, mixing=0.3, max_degree=1500, seed=3,
                                               return_communities=True)
    assert n2 == n and int(ci.numel()) == 2 * edges + n and int(rp[-1]) == 2 * edges + n
    deg = (rp[1:] - rp[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n), deg)
    A = torch.sparse_coo_tensor(torch.stack([rows, ci.long()]), va, (n, n)).to_dense()
    assert torch.equal(A, A.t()) and bool((torch.diagonal(A) > 0).all())
    u = deg.double().pow(-0.5)
    assert torch.allclose(va.double(), u[rows] * u[ci.long()], rtol=1e-6)
    assert int(deg.max()) > 5 * float(deg.float().mean())                # hubs
    sizes = torch.bincount(comm)
    assert sizes.numel() == 12 and int(sizes.max()) > 4 * int(sizes.min())
    off = rows != ci.long()
    cross = float((comm[rows[off]] != comm[ci.long()[off]]).float().mean())
    assert 0.25 <= cross <= 0.45, cross
    # labels shuffled: consecutive vertices are in the same community no more often than chance
    same_next = float((comm[:-1] == comm[1:]).float().mean())
    assert same_next < 2.0 * float((sizes.double() / n).pow(2).sum())
    again = graphgen.make_dcsbm(n=n, edges=edges, communities=12, mixing=0.3, max_degree=1500, seed=3)
    assert torch.equal(again[1], ci) and torch.equal(again[0], rp)
    planted = graphgen.make_dcsbm(n=n, edges=edges, communities=12, mixing=0.3, max_degree=1500, seed=3, relabel=False,
                                  return_communities=True)[4]
    assert bool((planted[1:] >= planted[:-1]).all())                     # un-shuffled: community by community
