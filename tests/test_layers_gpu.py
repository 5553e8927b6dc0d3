"""The callers of the op (GraphConvolution / GraphConvolution2 / GCN, mirrored from gcn6.py) on
the GPU: forward parity with the Python reference's recorded outputs, invariance under the
reorderers, fused epilogue, and a short training run."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import gcn_amd
from util import GOLDEN, rel_err

pytestmark = pytest.mark.gpu


def _golden_problem():
    g = np.load(os.path.join(GOLDEN, "gcn1_cora_shaped.npz"))
    n = int(g["n"])
    Ahat = sp.coo_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(n, n)).tocsr()
    raw = Ahat.copy(); raw.data[:] = 1.0; raw.setdiag(0); raw.eliminate_zeros()
    X = sp.coo_matrix((g["x_val"], (g["x_row"], g["x_col"])), shape=(n, int(g["nfeat"]))).tocsr()
    return g, n, raw, X


def _load_weights(model, g):
    with torch.no_grad():
        model.gc1.weight.copy_(torch.from_numpy(g["w1"])); model.gc1.bias.copy_(torch.from_numpy(g["b1"]))
        model.gc2.weight.copy_(torch.from_numpy(g["w2"])); model.gc2.bias.copy_(torch.from_numpy(g["b2"]))


@pytest.mark.parametrize("order", [None, "dfs", "gorder", "rabbit", "rcm", "deg", "communities"])
@pytest.mark.parametrize("fused", [False, True])
def test_forward_matches_the_python_reference_output(order, fused):
    """gcn1's recorded log-softmax output (both layers A(XW), dataset 'pubmed' selects that order,
    gcn6.py:215-216) is reproduced for every vertex order, in the ORIGINAL numbering"""
    g, n, raw, X = _golden_problem()
    model = gcn_amd.GCN(int(g["nfeat"]), int(g["nhid"]), int(g["ncls"]), dataset="pubmed", device="cuda:0",
                        order=order, fuse_epilogue=fused).to("cuda:0")
    _load_weights(model, g)
    model.prepare(X, raw, np.zeros(n, dtype=np.int64))
    out = model.predict().cpu().numpy()
    assert rel_err(out, g["out"]) <= 1e-5
    if order:
        assert sorted(model.vo_mp.cpu().tolist()) == list(range(n))


def test_training_reduces_the_loss_and_timers_report():
    rng = np.random.default_rng(0)
    n, f, c = 3000, 64, 4
    labels = rng.integers(0, c, n)
    # planted partition: edges mostly inside a class, features carry a weak class signal
    u = rng.integers(0, n, 40000); same = rng.random(40000) < 0.85
    v = np.where(same, rng.permutation(n)[u % n], rng.integers(0, n, 40000))
    cls_members = [np.flatnonzero(labels == k) for k in range(c)]
    v = np.where(same, np.array([rng.choice(cls_members[labels[a]]) for a in u]), v)
    A = sp.coo_matrix((np.ones(len(u)), (u, v)), shape=(n, n)); A = ((A + A.T) > 0).astype(np.float32).tocsr()
    A.setdiag(0); A.eliminate_zeros()
    X = rng.standard_normal((n, f)).astype(np.float32) + 0.5 * np.eye(c)[labels] @ rng.standard_normal((c, f))
    idx_train = rng.choice(n, 600, replace=False)
    torch.manual_seed(15)
    model = gcn_amd.GCN(f, 16, c, dataset="synthetic", device="cuda:0", order="gorder", dropout=0.5).to("cuda:0")
    losses = model.fit(X, A, labels, idx_train, train_iters=60)
    assert losses[-1] < 0.6 * losses[0]
    idx_test = np.setdiff1d(np.arange(n), idx_train)[:1000]
    assert float(model.test(idx_test, labels)) > 0.6
    rep = model.timing_report()
    assert "layer1 xw" in rep and "layer2 xw" in rep and model.gc1.timers.c.af.n_calls >= 60
    assert model.gc1.timers.c.af.avms() > 0          # HIP-event timer on the SpMM interval


def test_layer_order_auto_runs_the_spmm_at_the_narrower_width_with_the_same_result():
    """(ÂX)W and Â(XW) are the same function; 'auto' picks the one whose SpMM is narrower (§8f.1)"""
    g, n, raw, X = _golden_problem()
    nfeat, nhid, ncls = int(g["nfeat"]), int(g["nhid"]), int(g["ncls"])
    outs = {}
    for lo in ("reference", "auto"):
        model = gcn_amd.GCN(nfeat, nhid, ncls, dataset="reddit", device="cuda:0", order=None, layer_order=lo).to("cuda:0")
        _load_weights(model, g)
        model.prepare(X, raw, np.zeros(n, dtype=np.int64))
        outs[lo] = (type(model.gc2).__name__, model.predict().cpu().numpy())
    assert outs["reference"][0] == "GraphConvolution2"                     # gcn6.py:214-218 for 'reddit'
    assert outs["auto"][0] == ("GraphConvolution" if ncls <= nhid else "GraphConvolution2")
    assert rel_err(outs["auto"][1], outs["reference"][1]) <= 1e-5
    assert rel_err(outs["auto"][1], g["out"]) <= 1e-5
    with pytest.raises(ValueError):
        gcn_amd.GCN(nfeat, nhid, ncls, device="cuda:0", layer_order="fastest")
