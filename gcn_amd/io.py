"""On-disk formats either side of the SpMM path, so real graphs can be dropped
in when the files are supplied:

* text edge list ``u v`` per line — Edgelist(std::ifstream&) edgelist.cu:12-22, print_c :49-53
* order / rank file, one integer per line — read_order inout.cu:20-24, c_printorder :27-37
* GraphSAINT directory (adj_full.npz, adj_train.npz, feats.npy, class_map.json, role.json) —
  load_data / process_graph_data, profiling_gcn.py:22-72
"""
import json
import os

import numpy as np
import scipy.sparse as sp


def read_edgelist(path):
    """→ (n, edges[int64, e x 2]); n = 1 + largest vertex id seen (edgelist.cu:12-22)."""
    data = np.loadtxt(path, dtype=np.int64, ndmin=2) if os.path.getsize(path) else np.zeros((0, 2), np.int64)
    if data.size and data.shape[1] != 2:
        raise ValueError("edge list must have two integers per line")
    n = int(data.max()) + 1 if data.size else 1          # the reference returns max+1 with max=0 on empty input
    return n, data.reshape(-1, 2)


def write_edgelist(path, edges):
    """``u v`` per line (edgelist.cu:49-53)."""
    with open(path, "w") as f:
        for u, v in np.asarray(edges, dtype=np.int64).reshape(-1, 2):
            f.write(f"{u} {v}\n")


def edgelist_to_csr(n, edges, values=None):
    """One stored entry per listed edge, columns sorted within rows (int32 rowptr/col, fp32 val)."""
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    vals = np.ones(len(edges), np.float32) if values is None else np.asarray(values, np.float32)
    A = sp.coo_matrix((vals, (edges[:, 0], edges[:, 1])), shape=(n, n)).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)


def csr_to_edgelist(rowptr, col):
    """edgelist.cuh:16-25: one directed edge per stored entry, in CSR order."""
    rowptr = np.asarray(rowptr)
    rows = np.repeat(np.arange(len(rowptr) - 1, dtype=np.int64), np.diff(rowptr))
    return np.stack([rows, np.asarray(col, dtype=np.int64)], 1)


def read_order(path):
    """All integers of the file in order (inout.cu:20-24)."""
    with open(path) as f:
        return np.array([int(t) for t in f.read().split()], dtype=np.int64)


def write_order(path, rank, n=None):
    """One rank per line; entries >= n are skipped like c_printorder does (inout.cu:27-37)."""
    rank = np.asarray(rank, dtype=np.int64)
    n = len(rank) if n is None else n
    with open(path, "w") as f:
        for r in rank[:n]:
            if r >= n:
                continue
            f.write(f"{r}\n")


def load_graphsaint(prefix, normalize=True):
    """GraphSAINT-format dataset directory as profiling_gcn.py:22-72 reads it.
    → dict(adj, adj_train, features (standard-scaled on the training vertices), labels, idx_train,
    idx_val, idx_test)."""
    adj_full = sp.load_npz(os.path.join(prefix, "adj_full.npz"))
    adj_train = sp.load_npz(os.path.join(prefix, "adj_train.npz"))
    role = json.load(open(os.path.join(prefix, "role.json")))
    feats = np.load(os.path.join(prefix, "feats.npy"))
    class_map = {int(k): v for k, v in json.load(open(os.path.join(prefix, "class_map.json"))).items()}
    if len(class_map) != feats.shape[0]:
        raise ValueError("class_map and feats disagree on the number of vertices")
    if normalize:                                       # StandardScaler fit on training rows (:31-35)
        train_nodes = np.array(sorted(set(adj_train.nonzero()[0])))
        mu = feats[train_nodes].mean(0)
        sd = feats[train_nodes].std(0)
        sd[sd == 0] = 1.0
        feats = (feats - mu) / sd
    nv = adj_full.shape[0]
    first = next(iter(class_map.values()))
    labels = np.zeros(nv, dtype=np.int64)
    if isinstance(first, list):                         # multi-label → argmax, in file order (:57-63)
        for p, (_k, v) in enumerate(class_map.items()):
            labels[p] = int(np.argmax(v))
    else:
        for k, v in class_map.items():
            labels[k] = v
    return dict(adj=adj_full, adj_train=adj_train, features=feats, labels=labels,
                idx_train=np.array(role["tr"]), idx_val=np.array(role["va"]), idx_test=np.array(role["te"]))


def load_deeprobust_npz(path, require_lcc=True):
    """DeepRobust-style dataset file (cora.npz, citeseer.npz, … — what dataio.py:128-150 opens and
    :106-126 post-processes): CSR triplets `adj_*`, optional `attr_*`, optional `labels`.
    → (adj, features, labels): adj symmetrised, unweighted, zero diagonal, fp32 CSR, restricted to the
    largest connected component when asked (ties between equally large components go to the one scipy
    numbers last, as the reference's reversed argsort does); features fp32 CSR (identity when the file
    has none).  Loaded with numpy's pickle-free reader."""
    with np.load(path, allow_pickle=False) as z:
        adj = sp.csr_matrix((z["adj_data"], z["adj_indices"], z["adj_indptr"]), shape=tuple(z["adj_shape"]))
        if "attr_data" in z.files:
            feats = sp.csr_matrix((z["attr_data"], z["attr_indices"], z["attr_indptr"]), shape=tuple(z["attr_shape"]))
        else:
            feats = sp.identity(adj.shape[0], format="csr")
        labels = z["labels"] if "labels" in z.files else None
    feats = sp.csr_matrix(feats, dtype=np.float32)
    adj = (adj + adj.T).tolil()
    adj[adj > 1] = 1
    if require_lcc:
        _, comp = sp.csgraph.connected_components(adj)
        keep_comp = np.argsort(np.bincount(comp))[::-1][0]
        keep = np.flatnonzero(comp == keep_comp)
        adj = adj[keep][:, keep]
        feats = feats[keep]
        labels = labels[keep] if labels is not None else None
        if np.asarray(adj.sum(0)).ravel().min() <= 0:
            raise ValueError("graph contains singleton nodes")
    adj.setdiag(0)
    adj = adj.astype(np.float32).tocsr()
    adj.eliminate_zeros()
    if abs(adj - adj.T).sum() != 0:
        raise ValueError("input graph is not symmetric")
    if adj.nnz and not (adj.max() == 1 and np.all(adj.data == 1)):
        raise ValueError("graph must be unweighted")
    return adj, feats, labels
