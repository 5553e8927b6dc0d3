#!/usr/bin/env python3
"""tools/hub_probe.py — does a hub / cold split of the feature-row gathers pay on a skewed graph whose table is far
larger than the caches?  (it loses both ways.)  The graph is renumbered by degree
(hubs first).  `--split H,...`: A = A_hub (columns < H, compact table of H rows) + A_cold as two SpMMs through the public
API.  Without --split: the cache-policy variant — columns < H gathered by ordinary loads, the rest by streaming loads —
which needs the `HUB` instantiation of spmm_chunk_kernel and the GCN_AMD_HUB_COLS knob that lived in the tree from
9c723bb to the commit that removed them again (profiles/r03c_hub_cache_policy_probe_rmat24.log is its output); with
today's library that sweep times the plain kernel for every H.  Development aid.
    python tools/hub_probe.py --scale 24 --k 512 --split 8192,16384,65536"""
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
    ap.add_argument("--split", default="", help="comma list of H: time A = A_hub (columns < H, compact table of H rows) + A_cold "
                                                "as two SpMMs instead (the policy sweep is skipped)")
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
    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    if args.split:
        # two plans: the hub columns' entries against a compact table of H feature rows (a few MiB per 64-column tile:
        # L2-resident, sliced by the plan's own rule when it is not), the rest against the whole table
        os.environ["GCN_AMD_HUB_COLS"] = "0"
        full = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True)
        t_full = timed(lambda: full.matmul_raw(H, out=out))
        print(f"full matrix: {t_full:.3f} ms  {full.main_kernel(k)} slices {full.num_slices}", flush=True)
        del full
        rows = torch.repeat_interleave(torch.arange(n, device=dev), (rowptr[1:] - rowptr[:-1]).long())
        for Hc in [int(h) for h in args.split.split(",")]:
            m_hub = col < Hc
            parts = {}
            for name, mask, ncols in (("hub", m_hub, Hc), ("cold", ~m_hub, n)):
                cnt = torch.bincount(rows[mask], minlength=n)
                rp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
                rp[1:] = torch.cumsum(cnt, 0).to(torch.int32)
                parts[name] = gcn_amd.CsrAdjacency(rp, col[mask], val[mask], (n, ncols))
                del cnt
            Hhub = H[:Hc].contiguous()
            out2 = torch.empty_like(out)
            t_hub = timed(lambda: parts["hub"].matmul_raw(Hhub, out=out2))
            t_cold = timed(lambda: parts["cold"].matmul_raw(H, out=out))
            t_add = timed(lambda: out.add_(out2))
            share = float(cs[Hc - 1])
            print(f"H={Hc} share {share:.3f}: hub {t_hub:.3f} ms ({parts['hub'].main_kernel(k)}, slices {parts['hub'].num_slices}) "
                  f"cold {t_cold:.3f} ms  add {t_add:.3f} ms  sum {t_hub + t_cold + t_add:.3f} vs full {t_full:.3f}", flush=True)
            del parts, out2, Hhub
        return
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
