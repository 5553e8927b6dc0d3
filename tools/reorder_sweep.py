#!/usr/bin/env python3
"""tools/reorder_sweep.py — SpMM time under the host reorderers (BASELINE configs 3 and 5 in
miniature): none / degree-desc / RCM / Gorder(w=3) / DFS / Rabbit.  The orderings are computed by
libgcnspmm's host code (bit-exact with the reference), the CSR is rewritten, features are
permuted on the GPU (gather_rows) and the result is checked against the un-reordered result.
    python tools/reorder_sweep.py --graph products --scale 0.1 --k 256 --orders none,deg,rcm,gorder
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd                      # noqa: E402
from gcn_amd import graphgen, reorder   # noqa: E402


def timed_spmm(adj, H, iters):
    out = torch.empty((adj.m, H.shape[1]), device=H.device)
    for _ in range(3):
        adj.matmul_raw(H, out=out)
    torch.cuda.synchronize()
    adj.profile_begin(iters)
    for _ in range(iters):
        adj.matmul_raw(H, out=out)
    ms = adj.profile_end()
    return sum(ms) / len(ms), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="products")
    ap.add_argument("--scale", type=float, default=0.1)
    ap.add_argument("--rmat-scale", type=int, default=0, help="use a Graph500 R-MAT of 2^S vertices instead of --graph")
    ap.add_argument("--sbm", type=int, default=0, help="use a planted-partition graph of N vertices (communities of 512)")
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--orders", default="none,deg,rcm,gorder,dfs")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--slices", default="auto", help="auto | 0 | S : XCD-aware column slicing")
    ap.add_argument("--panels", default="auto", help="auto | 0 | 1 : LDS-staged row panels")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.sbm:
        rowptr, col, val, n = graphgen.make_sbm(args.sbm, device=dev, seed=7)
        args.graph, args.scale = f"sbm{args.sbm}", 1.0
    elif args.rmat_scale:
        rowptr, col, val, n = graphgen.make_rmat(args.rmat_scale, device=dev, seed=5)
        args.graph, args.scale = f"rmat{args.rmat_scale}", 1.0
    else:
        rowptr, col, val, n = graphgen.make_graph(args.graph, device=dev, seed=3, scale=args.scale)
    nnz, k = int(col.numel()), args.k
    H = graphgen.random_features(n, k, seed=2, device=dev)
    rp, ci, va = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    balg = nnz * (8 + 4 * k) + (n + 1) * 4 + n * k * 4
    deg = rowptr[1:] - rowptr[:-1]
    print(f"# graph {args.graph} scale={args.scale} n={n} nnz={nnz} k={k} mean_deg={nnz / n:.1f} "
          f"max_deg={int(deg.max())}", flush=True)
    print("order reorder_s kernel_ms GFLOP/s algGB/s frac_of_8TBps max_rel_err_vs_unordered", flush=True)
    base = None
    for name in args.orders.split(","):
        t0 = time.time()
        if name == "none":
            rp2, ci2, va2, vomp = rp, ci, va, np.arange(n, dtype=np.int32)
        elif name == "deg":
            rp2, ci2, va2, vomp = reorder.apply_rank(rp, ci, va, reorder.order_deg(rp, ci, "total", True))
        elif name == "rcm":
            rp2, ci2, va2, vomp = reorder.apply_rank(rp, ci, va, reorder.order_rcm(rp, ci))
        elif name == "gorder":
            rp2, ci2, va2, vomp = reorder.gorder(rp, ci, va)
        elif name == "dfs":
            rp2, ci2, va2, vomp = reorder.dfs(rp, ci, va)
        elif name == "rabbit":
            rp2, ci2, va2, vomp = reorder.rabbit(rp, ci, va)
        elif name in ("comm_gpu", "rcm_gpu"):                    # device orderings + device CSR rewrite
            rank = (reorder.order_communities_device(rowptr, col) if name == "comm_gpu"
                    else reorder.order_rcm_device(rowptr, col))
            out_d = reorder.apply_rank_device(rowptr, col, val, rank)
            torch.cuda.synchronize()
            rp2, ci2, va2, vomp = [x.cpu().numpy() for x in out_d]
        else:
            raise SystemExit(f"unknown order {name}")
        t_re = time.time() - t0
        adj = gcn_amd.CsrAdjacency(torch.from_numpy(rp2).to(dev), torch.from_numpy(ci2).to(dev),
                                   torch.from_numpy(va2).to(dev), (n, n), symmetric=True,
                                   slices=args.slices if args.slices == "auto" else int(args.slices),
                                   panels=args.panels if args.panels == "auto" else int(args.panels))
        vomp_d = torch.from_numpy(vomp).to(dev)
        Hp = gcn_amd.gather_rows(H, vomp_d)           # B[r,:] <- B[vomp[r],:]  (gcn6.py step 4)
        ms, out = timed_spmm(adj, Hp, args.iters)
        if base is None:
            base, err = out.clone(), 0.0
        else:
            err = float((out - base[vomp_d.long()]).abs().max() / base.abs().max())
        print(f"{name}[S={adj.num_slices},panelR={adj.panel_rows},cov={adj.panel_coverage:.2f}] {t_re:.2f} {ms:.4f} {2.0 * nnz * k / ms / 1e6:.1f} {balg / ms / 1e6:.1f} "
              f"{balg / ms / 1e-3 / 8e12:.4f} {err:.2e}", flush=True)
        del adj, Hp, out


if __name__ == "__main__":
    main()
