#!/usr/bin/env python3
"""tools/graph_probe.py — what would capturing the per-layer launch sequence of the planes in a hipGraph buy?
  Rank 0's block of the W-way partition of the Reddit-shaped graph, compute only: 2 layers
(ping-pong buffers) captured with torch.cuda.graph and replayed, against the same layers launched eagerly.
    python tools/graph_probe.py --world 8"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd                                  # noqa: E402
from gcn_amd import graphgen                    # noqa: E402
from gcn_amd.dist import PipelinedAggregation, RowShardedAdjacency   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--layers", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lrp, lcol, lval, n, bounds, u, nnz = graphgen.make_graph_row_block("reddit", args.world, 0, device=dev, seed=1)
    shard = RowShardedAdjacency.from_row_block(lrp, lcol, lval, bounds, 0, args.world,
                                               lambda rp, ci, va, shape, slices="auto": gcn_amd.CsrAdjacency(rp, ci, va, shape, slices=slices),
                                               value_factor=u, total_nnz=nnz)
    shard.collective = False
    pipe = PipelinedAggregation(shard, args.k, dev, plane_cols=64)
    for b in pipe.src:
        b.normal_()
    for _ in range(10):
        pipe.step()
    pipe.finish()
    torch.cuda.synchronize()

    def timed(fn, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        pipe.finish()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    eager = timed(lambda: (pipe.step(), pipe.step()), args.layers // 2) / 2
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pipe.step()
        pipe.step()
        pipe.finish()
    graph = timed(g.replay, args.layers // 2) / 2
    print(f"world {args.world} prelaid {shard.prelaid}: eager {eager * 1e3:.4f} ms per layer, hipGraph replay {graph * 1e3:.4f} ms per layer", flush=True)


if __name__ == "__main__":
    main()
