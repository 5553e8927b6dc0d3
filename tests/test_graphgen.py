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
