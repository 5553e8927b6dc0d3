#!/usr/bin/env python3
"""tools/sweep.py — kernel-time sweep on one GPU (development aid, not the judged bench).
Generates the graph once, then times the SpMM for several feature widths / chunk sizes.
    python tools/sweep.py --graph reddit --ks 64,128,256,512 --chunks 0,128,256,512,1024
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd                      # noqa: E402
from gcn_amd import graphgen        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="reddit")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--ks", default="128")
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--slices", default="0")
    ap.add_argument("--blocks-per-cu", default="32", help="comma list")
    ap.add_argument("--gather-width", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rowptr, col, val, n = graphgen.make_graph(args.graph, device=dev, seed=1, scale=args.scale)
    nnz = int(col.numel())
    deg = (rowptr[1:] - rowptr[:-1])
    print(f"# graph {args.graph} n={n} nnz={nnz} mean_deg={nnz / n:.1f} max_deg={int(deg.max())}", flush=True)
    print("k tile slices chunk nchunks mainkernels_ms(avg) mainkernels_ms(min) spmm_ms GFLOP/s(spmm) algGB/s(spmm) frac_of_8TBps(spmm)", flush=True)
    for k in [int(x) for x in args.ks.split(",")]:
        H = graphgen.random_features(n, k, seed=2, device=dev)
        out = torch.empty((n, k), device=dev)
        for chunk, tile, S, bpc in [(int(x), int(t), int(sl), int(b)) for x in args.chunks.split(",")
                                    for t in args.tiles.split(",") for sl in args.slices.split(",")
                                    for b in str(args.blocks_per_cu).split(",")]:
            adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True, chunk_nnz=chunk)
            adj.set_tile_cols(tile)
            adj.enable_slicing(S)
            adj.set_blocks_per_cu(bpc)
            adj.set_gather_width(args.gather_width)
            for _ in range(3):
                adj.matmul_raw(H, out=out)
            torch.cuda.synchronize()
            adj.profile_begin(args.iters)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                adj.matmul_raw(H, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = adj.profile_end()
            step = e0.elapsed_time(e1) / args.iters
            avg, mn = sum(ms) / len(ms), min(ms)
            balg = nnz * (8 + 4 * k) + (n + 1) * 4 + n * k * 4
            print(f"{k} {tile} {S}/bpc{bpc} {adj.chunk_size} {adj.num_chunks} {adj.main_kernel(min(k, 64))[5:40]} {avg:.4f} {mn:.4f} {step:.4f} "
                  f"{2.0 * nnz * k / step / 1e6:.1f} {balg / step / 1e6:.1f} {balg / step / 1e-3 / 8e12:.4f}", flush=True)
            del adj
        del H, out


if __name__ == "__main__":
    main()
