#!/usr/bin/env python3
"""tools/ceiling_probe.py — issue-rate ceiling of the chunk kernel: a dense band matrix (every gather
hits L1/L2) timed at several k / tile widths.  entries/s per 64-column pass that does not move with
the bytes per gather means the kernel is instruction-issue bound there, not memory bound."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd
from gcn_amd import graphgen

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 240000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 368
r = torch.arange(n, device=dev, dtype=torch.int64)
col = ((r[:, None] - d // 2 + torch.arange(d, device=dev)[None, :]) % n).sort(dim=1).values.to(torch.int32).reshape(-1)
rowptr = (torch.arange(n + 1, device=dev, dtype=torch.int64) * d).to(torch.int32)
val = torch.rand(n * d, device=dev) / d
nnz = n * d
adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=False, slices=0)
print(f"band n={n} d={d} nnz={nnz} chunk_nnz={adj.chunk_nnz}")
for k, tile in ((16, 0), (32, 0), (64, 64), (128, 64), (128, 128), (256, 64), (256, 256)):
    H = graphgen.random_features(n, k, seed=2, device=dev)
    out = torch.empty((n, k), device=dev)
    if tile:
        adj.set_tile_cols(tile)
    for _ in range(3):
        adj.matmul_raw(H, out=out)
    torch.cuda.synchronize()
    passes = adj.num_passes(k)
    adj.profile_begin(5 * passes)          # one event pair per main-kernel launch (= per pass)
    for _ in range(5):
        adj.matmul_raw(H, out=out)
    ms = adj.profile_end()
    t = sum(ms) / 5
    print(f"k={k:4d} tile={tile:4d} passes={passes} ms={t:.4f} Gentries/s={nnz * passes / t / 1e6:.1f} "
          f"gathered TB/s={nnz * k * 4 / t / 1e9:.2f}", flush=True)
