"""Data formats and preprocessing either side of the hot path."""
import json
import os

import numpy as np
import scipy.sparse as sp
import torch

from gcn_amd import io as gio, preprocess, reorder, timers
from util import GOLDEN


def test_edgelist_roundtrip_and_vertex_count(tmp_path):
    p = tmp_path / "g.txt"
    edges = np.array([[0, 3], [3, 0], [2, 2], [7, 1]])
    gio.write_edgelist(p, edges)
    assert open(p).read() == "0 3\n3 0\n2 2\n7 1\n"
    n, e = gio.read_edgelist(p)
    assert n == 8 and np.array_equal(e, edges)           # n = max id + 1 (edgelist.cu:17-21)
    rp, ci, va = gio.edgelist_to_csr(n, e)
    assert np.array_equal(gio.csr_to_edgelist(rp, ci), [[0, 3], [2, 2], [3, 0], [7, 1]])
    # an ordering computed from the file-loaded graph is a permutation of its vertices
    assert sorted(reorder.order_rcm(rp, ci).tolist()) == list(range(n))


def test_order_file_roundtrip_skips_invalid_ranks(tmp_path):
    p = tmp_path / "ord.txt"
    gio.write_order(p, [2, 0, 9, 1], n=4)                 # 9 >= n is skipped (inout.cu:30-33)
    assert open(p).read() == "2\n0\n1\n"
    assert np.array_equal(gio.read_order(p), [2, 0, 1])


def test_graphsaint_directory(tmp_path):
    n, f = 12, 5
    rng = np.random.default_rng(0)
    A = sp.random(n, n, 0.3, random_state=1, format="csr"); A = ((A + A.T) > 0).astype(np.float32).tocsr()
    tr = sp.csr_matrix(A.toarray() * (np.arange(n)[:, None] < 8) * (np.arange(n)[None, :] < 8))
    sp.save_npz(tmp_path / "adj_full.npz", A); sp.save_npz(tmp_path / "adj_train.npz", tr)
    feats = rng.standard_normal((n, f)); np.save(tmp_path / "feats.npy", feats)
    json.dump({str(i): int(i % 3) for i in range(n)}, open(tmp_path / "class_map.json", "w"))
    json.dump({"tr": list(range(8)), "va": [8, 9], "te": [10, 11]}, open(tmp_path / "role.json", "w"))
    d = gio.load_graphsaint(str(tmp_path))
    train_nodes = np.array(sorted(set(tr.nonzero()[0])))
    assert np.allclose(d["features"][train_nodes].mean(0), 0, atol=1e-12)
    assert np.allclose(d["features"][train_nodes].std(0), 1, atol=1e-12)
    assert np.array_equal(d["labels"], np.arange(n) % 3) and list(d["idx_val"]) == [8, 9]


def test_normalize_adj_reproduces_the_reference_fixture():
    """Â recorded from the reference's utils.normalize_adj_tensor (golden) == our restatement"""
    g = np.load(os.path.join(GOLDEN, "gcn1_cora_shaped.npz"))
    n = int(g["n"])
    ref = sp.coo_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(n, n)).tocsr()
    raw = ref.copy(); raw.data[:] = 1.0; raw.setdiag(0); raw.eliminate_zeros()
    t = preprocess.normalize_adj_tensor(raw).coalesce()
    assert np.array_equal(t.indices().numpy(), np.vstack([g["adj_row"], g["adj_col"]]))
    assert np.array_equal(t.values().numpy(), g["adj_val"])                 # bit-exact fp32 values
    rp, ci, va, vo = preprocess.to_csr_int32(t)
    assert rp.dtype == torch.int32 and int(rp[-1]) == 12623 and torch.equal(vo, torch.arange(n, dtype=torch.int32))
    # the utils.py:82 quirk: no self-loops are added when mx[0,0] != 0
    quirk = raw.tolil(); quirk[0, 0] = 1.0
    assert preprocess.normalize_adj(quirk).tocsr().nnz == raw.nnz + 1


def test_host_timers_accumulate():
    t = timers.Timers()
    for _ in range(3):
        with t.hc.af:
            sum(range(1000))
    assert t.h.af.n_calls == 3 and t.h.af.ns() > 0 and t.h.af.avms() > 0
    assert t.c.af.n_calls == 3 and t.c.af.ms() == 0.0     # no device here: the device leg is off
    t.reset()
    assert t.h.af.n_calls == 0


def test_deeprobust_npz_file(tmp_path):
    """the .npz dataset layout of dataio.py:128-150 and its post-processing (:106-126): symmetrise,
    binarise, largest connected component, zero diagonal"""
    rng = np.random.default_rng(1)
    n = 60
    # component A: vertices 0..39 (ring + chords, some entries doubled and directed), component B: 40..54 (ring),
    # 55..59 isolated
    r = list(range(40)) + [3, 7, 7] + list(range(40, 55))
    c = [(i + 1) % 40 for i in range(40)] + [20, 30, 30] + [40 + (i + 1) % 15 for i in range(15)]
    A = sp.coo_matrix((np.ones(len(r)), (r, c)), shape=(n, n)).tocsr()       # directed, (7,30) stored twice -> 2.0
    A[5, 5] = 1.0                                                            # a self-loop to be dropped
    X = sp.random(n, 9, density=0.3, random_state=3, format="csr")
    y = rng.integers(0, 4, n)
    path = tmp_path / "toy.npz"
    np.savez(path, adj_data=A.data, adj_indices=A.indices, adj_indptr=A.indptr, adj_shape=A.shape,
             attr_data=X.data, attr_indices=X.indices, attr_indptr=X.indptr, attr_shape=X.shape, labels=y)
    adj, feats, labels = gio.load_deeprobust_npz(str(path))
    assert adj.shape == (40, 40) and feats.shape == (40, 9) and labels.shape == (40,)
    assert adj.dtype == np.float32 and abs(adj - adj.T).sum() == 0 and adj.diagonal().sum() == 0
    assert set(np.unique(adj.data)) == {1.0} and adj.nnz == 2 * (40 + 2)     # ring + two distinct chords
    assert np.array_equal(labels, y[:40]) and np.allclose(feats.toarray(), X[:40].toarray())
    # without the component filter: all vertices stay, no features -> identity
    np.savez(tmp_path / "nofeat.npz", adj_data=A.data, adj_indices=A.indices, adj_indptr=A.indptr, adj_shape=A.shape)
    adj2, feats2, labels2 = gio.load_deeprobust_npz(str(tmp_path / "nofeat.npz"), require_lcc=False)
    assert adj2.shape == (n, n) and labels2 is None and (feats2 != sp.identity(n)).nnz == 0
