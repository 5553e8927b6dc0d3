#!/usr/bin/env python3
"""tools/hub_probe.py — does a hub / cold split of the feature-row gathers pay on a skewed graph whose table is far
larger than the caches?    The graph is renumbered by degree (hubs first); for every tile width
the unsliced kernel runs with columns < H gathered by ordinary loads (they stay in L2) and the rest by streaming
loads (GCN_AMD_HUB_COLS, spmm_chunk_kernel<.., HUB>), H = 0 being today's kernel.  Development aid.
    python tools/hub_probe.py --scale 24 --k 512 --tiles 256,64 --hubs 0,2048,4096,16384"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd                                  # noqa: E402
from gcn_amd import graphgen, reorder           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="rmat")
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--k", type=int, default=512)
    ap.add_argument("--tiles", default="256,128,64")
    ap.add_argument("--hubs", default="0,2048,4096,8192,16384")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.graph == "rmat":
        rowptr, col, val, n = graphgen.make_rmat(args.scale, device=dev, seed=5)
    else:
        rowptr, col, val, n = graphgen.make_graph(args.graph, device=dev, seed=1)
    rank = reorder.order_deg_device(rowptr, col, "total", True)
    rowptr, col, val, _ = reorder.apply_rank_device(rowptr, col, val, rank)
    del rank
    torch.cuda.empty_cache()
    nnz, k = int(col.numel()), args.k
    indeg = torch.bincount(col.long(), minlength=n)
    cs = torch.cumsum(indeg, 0).double() / nnz
    print(f"# {args.graph} scale {args.scale} n={n} nnz={nnz} k={k} (degree-descending numbering)", flush=True)
    H = graphgen.random_features(n, k, seed=2, device=dev)
    out = torch.empty((n, k), device=dev)
    balg = nnz * (8 + 4 * k) + (n + 1) * 4 + n * k * 4
    print("tile hub_cols share_of_nnz spmm_ms algTB/s", flush=True)
    for tile in [int(t) for t in args.tiles.split(",")]:
        for hub in [int(h) for h in args.hubs.split(",")]:
            os.environ["GCN_AMD_HUB_COLS"] = str(hub)
            adj = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True, slices=0)
            adj.set_tile_cols(tile)
            adj.set_gather_width(1)
            for _ in range(2):
                adj.matmul_raw(H, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                adj.matmul_raw(H, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            share = float(cs[hub - 1]) if hub > 0 else 0.0
            print(f"{tile} {hub} {share:.3f} {ms:.3f} {balg / ms / 1e9:.2f}  {adj.main_kernel(k)}", flush=True)
            del adj


if __name__ == "__main__":
    main()
