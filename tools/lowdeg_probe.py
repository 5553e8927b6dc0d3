#!/usr/bin/env python3
"""tools/lowdeg_probe.py — one vs four non-zeros per gather on LOW-degree graphs (R-MAT, edge factor 2..16):
where do short rows make the quad kernel's per-row reduction cost more than its wider loads save?"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd
from gcn_amd import graphgen

dev = torch.device("cuda:0")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for ef in (2, 4, 8, 16, 32):
    rowptr, col, val, n = graphgen.make_rmat(scale, edge_factor=ef, device=dev, seed=5)
    nnz = int(col.numel())
    for k in (64, 128):
        H = graphgen.random_features(n, k, seed=2, device=dev)
        out = torch.empty((n, k), device=dev)
        res = []
        for gw in (1, 4):
            adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True)
            adj.set_tile_cols(64)
            adj.set_gather_width(gw)
            for _ in range(3):
                adj.matmul_raw(H, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                adj.matmul_raw(H, out=out)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 10)
        print(f"scale={scale} ef={ef} n={n} nnz={nnz} mean_deg={nnz / n:.1f} k={k}: one-per-gather {res[0]:.4f} ms, "
              f"four-per-gather {res[1]:.4f} ms, ratio {res[0] / res[1]:.3f}", flush=True)
