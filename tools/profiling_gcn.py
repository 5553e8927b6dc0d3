#!/usr/bin/env python3
"""tools/profiling_gcn.py — the reference's driver (profiling_gcn.py:85-170, run.sh) on gcn_amd:

    python tools/profiling_gcn.py -g reddit -k 128 -i 100 [--order rabbit|gorder|dfs|rcm|deg|none] [--fuse]

Loads ./dataset/<graph>/ in GraphSAINT format when it exists (profiling_gcn.py:22-37); otherwise
(no dataset ships offline) trains on the shape-matched synthetic stand-in with random features
and labels.  Prints gcn6's per-layer xw / af / bi timing lines (gcn6.py:401-410).
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd                          # noqa: E402
from gcn_amd import graphgen, io as gio  # noqa: E402


def main():
    ap = argparse.ArgumentParser("Graph to be processed ... ")
    ap.add_argument("-g", "--graph", default="reddit")
    ap.add_argument("-k", "--hidden", type=int, default=128)
    ap.add_argument("-i", "--train-iters", dest="train_iters", type=int, default=100)
    ap.add_argument("--order", default="none", choices=["none", "dfs", "gorder", "rabbit", "rcm", "deg", "communities"])
    ap.add_argument("--fuse", action="store_true", help="bias + ReLU in the SpMM epilogue")
    ap.add_argument("--precompute-ax", dest="precompute_ax", action="store_true",
                    help="layer 1 as (AX)W with AX aggregated once (X is constant): two SpMMs per epoch instead of four")
    ap.add_argument("--hip-graph", dest="hip_graph", action="store_true",
                    help="capture the training step in a HIP graph and replay it (small graphs: launch-bound epochs)")
    ap.add_argument("--layer-order", default="reference", choices=["reference", "auto"],
                    help="layer 2 as the reference hard-codes it per dataset, or with the SpMM at the narrower width")
    ap.add_argument("--warmup-iters", type=int, default=3, help="untimed iterations before the timed ones (0: as the reference)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--nfeat", type=int, default=0, help="input width of the synthetic stand-in (0: the dataset's published one)")
    ap.add_argument("--nclass", type=int, default=0, help="classes of the synthetic stand-in (0: the dataset's published count)")
    args = ap.parse_args()
    seed = 15                                               # profiling_gcn.py:76-80
    np.random.seed(seed); torch.manual_seed(seed); torch.cuda.manual_seed(seed)

    prefix = os.path.join("dataset", args.graph)
    if os.path.isdir(prefix):
        d = gio.load_graphsaint(prefix)
        adj, features, labels, idx_train = d["adj"], d["features"], d["labels"], d["idx_train"]
        normalize = True
    else:
        shape = args.graph if args.graph in graphgen.SHAPES else "reddit"
        args.nfeat = args.nfeat or graphgen.SHAPES[shape].get("nfeat", 602)
        args.nclass = args.nclass or graphgen.SHAPES[shape].get("nclass", 41)
        rp, ci, va, n = graphgen.make_graph(shape, device="cuda:0", seed=1, scale=args.scale)
        adj = sp.csr_matrix((va.cpu().numpy(), ci.cpu().numpy(), rp.cpu().numpy()), shape=(n, n))
        normalize = False                                   # the generator already returns Â
        features = np.random.standard_normal((n, args.nfeat)).astype(np.float32)
        labels = np.random.randint(0, args.nclass, n)
        idx_train = np.random.choice(n, max(1, n // 2), replace=False)
        print(f"no ./dataset/{args.graph}: synthetic {shape}-shaped graph n={n} nnz={adj.nnz}")
    nclass = int(labels.max()) + 1
    model = gcn_amd.GCN(nfeat=features.shape[1], nhid=args.hidden, nclass=nclass, dataset=args.graph,
                        device="cuda:0", order=None if args.order == "none" else args.order,
                        fuse_epilogue=args.fuse, layer_order=args.layer_order, precompute_ax=args.precompute_ax).to("cuda:0")
    t0 = time.time()
    if args.warmup_iters > 0:
        # the reference averages its timers over every call including the first (library initialisation,
        # kernel loading: ~150 ms); a few untimed iterations first make the per-layer lines steady-state
        model.fit(features, adj, labels, idx_train, train_iters=args.warmup_iters, normalize=normalize)
        torch.cuda.synchronize()
        print(f"prepare + {args.warmup_iters} warm-up iterations: {time.time() - t0:.2f} s")
        model.reset_timing()
        t0 = time.time()
    losses = model.fit(features, adj, labels, idx_train, train_iters=args.train_iters, verbose=True,
                       normalize=normalize, initialize=args.warmup_iters == 0, reuse_prepared=args.warmup_iters > 0,
                       hip_graph=args.hip_graph)
    torch.cuda.synchronize()
    print(f"fit: {time.time() - t0:.2f} s, loss {losses[0]:.4f} -> {losses[-1]:.4f}, "
          f"slices={model.adj.num_slices} chunks={model.adj.num_chunks}x{model.adj.chunk_size}")
    print(model.timing_report())


if __name__ == "__main__":
    main()
