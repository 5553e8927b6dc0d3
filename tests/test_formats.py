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
