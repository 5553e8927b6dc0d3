#!/usr/bin/env python3
"""oracle/make_golden.py — TEST INFRASTRUCTURE ONLY.

Regenerates tests/golden/*.npz from the REFERENCE ITSELF, in the build container
(needs /root/reference; never runs on the GPU box):

* reorder_<case>.npz — inputs + bit-exact outputs of the reference's reorderers,
  obtained by running oracle/_ref/renumber_ref.so (dfs / gorder / rabbit /
  perm_apply through the reference's C ABI, renumber.cu:23,157,233,319) and
  oracle/_ref/libref_orders.so (order_deg / order_rcm / complete_gorder) — both
  compiled from the reference's sources by oracle/Makefile.
* gcn1_<case>.npz — Â, X, seeded weights and the layer outputs of the Python
  reference's own 2-layer GCN on CPU (pygcn/gcn1.py:40-58,102-126 with
  utils.normalize_adj_tensor, utils.py:78-90,126-135), i.e. outputs of the
  torch.spmm call site the HIP kernel replaces (gcn1.py:53).

* gcn1_train_cora_shaped.npz — the Python reference's own TRAINING loop on the same Cora-shaped problem:
  pygcn.gcn1.GCN.fit → _train_without_val (gcn1.py:132-217; Adam lr 0.01, weight decay 5e-4, nll_loss on the
  training rows), dropout 0, fixed seed, 20 epochs: initial weights, the training rows and labels, the loss of
  every epoch (recorded by wrapping the nll_loss the loop calls) and the final eval-mode log-probabilities.

The fixtures are data (inputs and expected outputs) — no reference source text.
Usage:  make -C oracle && python oracle/make_golden.py
"""
import ctypes
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


# ----------------------------------------------------------------------------
# small graphs covering the tie-break / traversal hazards listed in SURVEY.md §8a
# ----------------------------------------------------------------------------
def _finish(A, n, rng, loops=True, normalise=True):
    A = sp.csr_matrix(A, shape=(n, n), dtype=np.float64)
    A.setdiag(0)
    A.eliminate_zeros()
    A.data[:] = 1.0
    if loops:
        A = (A + sp.eye(n)).tocsr()
    if normalise:
        d = np.asarray(A.sum(1)).ravel()
        dinv = np.where(d > 0, d ** -0.5, 0.0)
        A = sp.diags(dinv) @ A @ sp.diags(dinv)
    else:
        A.data[:] = rng.random(A.nnz) + 0.25
    A = A.tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)


def cases():
    rng = np.random.default_rng(1234)
    out = {}

    def rand_sym(n, e, seed):
        r = np.random.default_rng(seed)
        u, v = r.integers(0, n, e), r.integers(0, n, e)
        A = sp.coo_matrix((np.ones(e), (u, v)), shape=(n, n))
        return A + A.T

    out["random_sym_n60"] = _finish(rand_sym(60, 150, 1), 60, rng)
    out["random_sym_n400"] = _finish(rand_sym(400, 2400, 2), 400, rng)
    r3 = np.random.default_rng(3)
    P = sp.coo_matrix((np.ones(3000), (np.minimum(299, (300 * r3.random(3000) ** 3).astype(int)),
                                       r3.integers(0, 300, 3000))), shape=(300, 300))
    out["powerlaw_n300"] = _finish(P + P.T, 300, rng)
    # star: one hub, all degrees tie otherwise
    n = 33
    u = np.zeros(n - 1, int); v = np.arange(1, n)
    A = sp.coo_matrix((np.ones(n - 1), (u, v)), shape=(n, n)); out["star_n33"] = _finish(A + A.T, n, rng)
    # path: maximal BFS depth
    n = 40
    A = sp.coo_matrix((np.ones(n - 1), (np.arange(n - 1), np.arange(1, n))), shape=(n, n))
    out["path_n40"] = _finish(A + A.T, n, rng)
    # disconnected: three components + vertices that only have their self-loop
    n = 70
    B1 = rand_sym(25, 60, 3); B2 = rand_sym(20, 30, 4); B3 = rand_sym(15, 40, 5)
    A = sp.block_diag([B1, B2, B3, sp.csr_matrix((10, 10))]); out["disconnected_n70"] = _finish(A, n, rng)
    # ring lattice: every vertex has the same degree (pure id tie-breaks)
    n = 48
    idx = np.arange(n)
    A = sp.coo_matrix((np.ones(2 * n), (np.r_[idx, idx], np.r_[(idx + 1) % n, (idx + 2) % n])), shape=(n, n))
    out["ring_n48"] = _finish(A + A.T, n, rng)
    # asymmetric (directed) graph without self-loops, arbitrary values; every vertex has an out-edge
    n = 90
    r = np.random.default_rng(7)
    u = np.r_[np.arange(n), r.integers(0, n, 400)]; v = np.r_[(np.arange(n) * 7 + 3) % n, r.integers(0, n, 400)]
    A = sp.coo_matrix((np.ones(len(u)), (u, v)), shape=(n, n))
    out["directed_noloops_n90"] = _finish(A, n, rng, loops=False, normalise=False)
    # two cliques joined by a bridge (Rabbit merges; Gorder windows)
    n = 24
    C = np.ones((12, 12)); A = sp.block_diag([C, C]).tolil(); A[11, 12] = 1; A[12, 11] = 1
    out["two_cliques_n24"] = _finish(A, n, rng)
    return out


def gen_reorder():
    ref = ctypes.CDLL(os.path.join(HERE, "_ref", "renumber_ref.so"))
    refo = ctypes.CDLL(os.path.join(HERE, "_ref", "libref_orders.so"))
    for name, (rp, ci, va) in cases().items():
        n, nnz = len(rp) - 1, len(ci)
        rec = dict(rowptr=rp, col=ci, val=va)
        for fn in ("dfs", "gorder", "rabbit"):
            a = [rp.copy(), ci.copy(), va.copy(), np.arange(n, dtype=np.int32)]
            getattr(ref, fn)(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), n, n, nnz)
            for key, arr in zip(("rowptr", "col", "val", "vomp"), a):
                rec[f"{fn}_{key}"] = arr
        # perm_apply with the reversal permutation
        vomp = np.arange(n, dtype=np.int32)[::-1].copy()
        a = [rp.copy(), ci.copy(), va.copy(), vomp.copy()]
        ref.perm_apply(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), n, n, nnz)
        rec["perm_apply_in_vomp"] = vomp
        for key, arr in zip(("rowptr", "col", "val"), a):
            rec[f"perm_apply_{key}"] = arr
        for which, wname in enumerate(("total", "out", "in")):
            for desc in (0, 1):
                o = np.zeros(n, np.int64)
                refo.ref_order_deg(_p(rp), _p(ci), n, nnz, which, desc, _p(o))
                rec[f"deg_{wname}_{'desc' if desc else 'asc'}"] = o
        for directed in (0, 1):
            o = np.zeros(n, np.int64)
            refo.ref_order_rcm(_p(rp), _p(ci), n, nnz, directed, _p(o))
            rec[f"rcm_{'directed' if directed else 'undirected'}"] = o
        for w in (1, 3, 5):
            o = np.zeros(n, np.int64)
            refo.ref_complete_gorder(_p(rp), _p(ci), n, nnz, w, _p(o))
            rec[f"gorder_w{w}"] = o
        np.savez_compressed(os.path.join(OUT, f"reorder_{name}.npz"), **rec)
        print("wrote reorder_%s.npz (n=%d nnz=%d)" % (name, n, nnz), flush=True)


def gen_gcn1():
    import torch
    sys.path.insert(0, REF)
    from pygcn.gcn1 import GCN            # noqa: E402  (the Python reference, CPU)
    from pygcn.gcnio.util import utils    # noqa: E402

    for name, n, e, nfeat, nnz_row, nhid, ncls, seed in (
            ("cora_shaped", 2485, 5069, 1433, 18, 16, 7, 0), ("tiny", 50, 120, 40, 6, 8, 3, 1)):
        rng = np.random.default_rng(seed)
        # n vertices, e distinct undirected edges uniformly at random
        keys = set()
        while len(keys) < e:
            u, v = rng.integers(0, n, 2)
            if u != v:
                keys.add((min(u, v), max(u, v)))
        u = np.array([k[0] for k in sorted(keys)]); v = np.array([k[1] for k in sorted(keys)])
        A = sp.coo_matrix((np.ones(e), (u, v)), shape=(n, n)); A = (A + A.T).tocsr()
        rows = np.repeat(np.arange(n), nnz_row); cols = rng.integers(0, nfeat, n * nnz_row)
        X = sp.csr_matrix((np.ones(n * nnz_row), (rows, cols)), shape=(n, nfeat)); X.data[:] = 1.0
        X = utils.normalize_feature(X).tocsr()
        labels = rng.integers(0, ncls, n)

        adj_t, feat_t, lab_t = utils.to_tensor(A, X, labels, device="cpu")
        adj_norm = utils.normalize_adj_tensor(adj_t, sparse=True)       # utils.py:126-135
        torch.manual_seed(15)                                           # profiling_gcn.py:76-80
        model = GCN(nfeat=nfeat, nhid=nhid, nclass=ncls, device="cpu")
        model.eval()
        with torch.no_grad():
            support1 = torch.spmm(feat_t, model.gc1.weight)             # gcn1.py:45
            agg1 = torch.spmm(adj_norm, support1)                       # gcn1.py:53  ← the hot op
            h1, _, _ = model.gc1(feat_t, adj_norm)
            out = model.forward(feat_t, adj_norm)[0]
            hidden = torch.relu(h1)
            support2 = torch.mm(hidden, model.gc2.weight)
            agg2 = torch.spmm(adj_norm, support2)
        an = adj_norm.coalesce()
        Xc = X.tocoo()
        np.savez_compressed(
            os.path.join(OUT, f"gcn1_{name}.npz"),
            n=n, nfeat=nfeat, nhid=nhid, ncls=ncls,
            adj_row=an.indices()[0].numpy().astype(np.int32), adj_col=an.indices()[1].numpy().astype(np.int32),
            adj_val=an.values().numpy(),
            x_row=Xc.row.astype(np.int32), x_col=Xc.col.astype(np.int32), x_val=Xc.data.astype(np.float32),
            w1=model.gc1.weight.detach().numpy(), b1=model.gc1.bias.detach().numpy(),
            w2=model.gc2.weight.detach().numpy(), b2=model.gc2.bias.detach().numpy(),
            support1=support1.numpy(), agg1=agg1.numpy(), h1=h1.numpy(),
            support2=support2.numpy(), agg2=agg2.numpy(), out=out.numpy())
        print("wrote gcn1_%s.npz (n=%d nnz(Â)=%d)" % (name, n, an.values().numel()), flush=True)


def gen_gcn1_train(epochs=20):
    """forward + backward-through-the-op + Adam, composed by the reference itself"""
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, REF)
    from pygcn.gcn1 import GCN            # noqa: E402
    g = np.load(os.path.join(OUT, "gcn1_cora_shaped.npz"))
    n, nfeat, nhid, ncls = int(g["n"]), int(g["nfeat"]), int(g["nhid"]), int(g["ncls"])
    Ahat = sp.coo_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(n, n)).tocsr()
    raw = Ahat.copy(); raw.data[:] = 1.0; raw.setdiag(0); raw.eliminate_zeros()      # A itself: fit() normalises (gcn1.py:147-151)
    raw = raw.tocoo()
    adj_t = torch.sparse_coo_tensor(np.vstack([raw.row, raw.col]).astype(np.int64), torch.ones(raw.nnz), (n, n)).coalesce()
    X = torch.sparse_coo_tensor(np.vstack([g["x_row"], g["x_col"]]).astype(np.int64), torch.from_numpy(g["x_val"]),
                                (n, nfeat)).coalesce()
    rng = np.random.default_rng(42)
    # labels a 1-layer GCN can explain (argmax of Â·X·W for a random W): the loss really falls, the gradients are not noise
    Xs = sp.coo_matrix((g["x_val"], (g["x_row"], g["x_col"])), shape=(n, nfeat)).tocsr()
    labels = np.asarray((Ahat @ (Xs @ rng.standard_normal((nfeat, ncls)))).argmax(1)).ravel().astype(np.int64)
    idx_train = np.sort(rng.choice(n, 500, replace=False)).astype(np.int64)
    torch.manual_seed(15)
    model = GCN(nfeat=nfeat, nhid=nhid, nclass=ncls, dropout=0.0, lr=0.01, weight_decay=5e-4, device="cpu")
    init = {k: v.detach().clone().numpy() for k, v in (("w1", model.gc1.weight), ("b1", model.gc1.bias),
                                                        ("w2", model.gc2.weight), ("b2", model.gc2.bias))}
    losses = []
    real_nll = F.nll_loss

    def recording_nll(*a, **kw):                     # the loop's own loss values, epoch by epoch (gcn1.py:184)
        out = real_nll(*a, **kw)
        losses.append(float(out.detach()))
        return out
    F.nll_loss = recording_nll
    try:
        model.fit(X, adj_t, torch.from_numpy(labels), torch.from_numpy(idx_train), train_iters=epochs,
                  initialize=False, normalize=True)
    finally:
        F.nll_loss = real_nll
    assert len(losses) == epochs and losses[-1] < losses[0]
    np.savez_compressed(
        os.path.join(OUT, "gcn1_train_cora_shaped.npz"), epochs=epochs, lr=0.01, weight_decay=5e-4, dropout=0.0,
        labels=labels, idx_train=idx_train, losses=np.asarray(losses, np.float64),
        final_out=model.output.detach().numpy(),
        w1_final=model.gc1.weight.detach().numpy(), b1_final=model.gc1.bias.detach().numpy(),
        w2_final=model.gc2.weight.detach().numpy(), b2_final=model.gc2.bias.detach().numpy(), **init)
    print("wrote gcn1_train_cora_shaped.npz: losses %.6f -> %.6f over %d epochs" % (losses[0], losses[-1], epochs), flush=True)


if __name__ == "__main__":
    if "--train-only" in sys.argv:
        gen_gcn1_train()
        sys.exit(0)
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    os.makedirs(OUT, exist_ok=True)
    gen_reorder()
    gen_gcn1()
    gen_gcn1_train()
