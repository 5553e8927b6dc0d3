#!/usr/bin/env python3
"""tools/renumber_sweep.py — does renumbering pay at the headline size on a graph that HAS structure?  (VERDICT r03,
next-round item 1: the measurement behind the kill criterion.)

For every mixing level a degree-corrected planted-partition graph of the Reddit-shaped size is generated with its
labels shuffled (graphgen.make_dcsbm), then the whole SpMM (k = 128 unless --k) is timed on
  * the graph as handed over ("no reorder"): the plan's automatic choice (value-free sliced pass);
  * the graph renumbered by `rabbit_device` (and by device RCM with --rcm): automatic slicing, explicit slice counts,
    no slicing, LDS / MFMA panels (automatic window coverage rule).
Times are whole SpMMs from HIP events; the renumbered results are checked against the un-renumbered one
(P·Â·Pᵀ·(P·B) = P·(Â·B), 1e-5).  The permutation passes are NOT in these times: the question is what the plan could
gain if they were folded away.  Development aid; prints a table and one JSON line per graph.

    python tools/renumber_sweep.py --mixing 0.2,0.35,0.5 --communities 200
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcn_amd                                  # noqa: E402
from gcn_amd import graphgen, reorder           # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mixing", default="0.2,0.35,0.5")
    ap.add_argument("--communities", type=int, default=200)
    ap.add_argument("--size-skew", type=float, default=1.0)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--slices", default="0,2,4,6,8,10,12")
    ap.add_argument("--rcm", action="store_true")
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--shape", default="reddit", choices=["reddit", "products"],
                    help="vertex / edge count of the graph: the Reddit-shaped headline size, or the products-shaped size of "
                         "BASELINE config 3 (n = 2 449 029, 61.9 M edges: mean degree 51, a table far larger than the caches)")
    ap.add_argument("--max-degree", type=int, default=20000)
    ap.add_argument("--tiles", default="", help="also: unsliced with these column tiles per pass (64,128,256)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n = int(graphgen.SHAPES[args.shape]["n"] * args.scale)
    edges = int(graphgen.SHAPES[args.shape]["edges"] * args.scale)
    k = args.k
    for mix in [float(x) for x in args.mixing.split(",")]:
        rowptr, col, val, n, comm = graphgen.make_dcsbm(n=n, edges=edges, communities=args.communities, mixing=mix,
                                                        size_skew=args.size_skew, device=dev, seed=args.seed,
                                                        max_degree=args.max_degree, return_communities=True)
        nnz = int(col.numel())
        deg = (rowptr[1:] - rowptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
        realised = float((comm[rows] != comm[col.long()]).float().mean())
        del rows
        print(f"# dcsbm n={n} nnz={nnz} communities={args.communities} (sizes {int(torch.bincount(comm).min())}..{int(torch.bincount(comm).max())}) "
              f"mixing={mix} realised={realised:.3f} max degree={int(deg.max())} k={k}", flush=True)
        B = graphgen.random_features(n, k, seed=2, device=dev)
        out = torch.empty((n, k), device=dev)
        res = {"graph": args.shape + "-dcsbm", "n": n, "nnz": nnz, "mixing": mix, "realised_mixing": round(realised, 4),
               "communities": args.communities, "k": k, "plans": {}}

        base = gcn_amd.CsrAdjacency(rowptr, col, val, (n, n), symmetric=True)
        t = timed(lambda: base.matmul_raw(B, out=out), args.iters)
        ref = out.clone()
        res["plans"]["as handed over / auto"] = {"ms": round(t, 4), "slices": base.num_slices, "kernel": base.main_kernel(k)}
        print(f"  as handed over, auto: {t:.3f} ms  slices={base.num_slices}  {base.main_kernel(k)}", flush=True)
        del base

        orders = [("rabbit_device", lambda: reorder.order_rabbit_device(rowptr, col, return_stats=True))]
        if args.rcm:
            orders.append(("rcm_device", lambda: (reorder.order_rcm_device(rowptr, col), {})))
        for oname, ofn in orders:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rank, stats = ofn()
            torch.cuda.synchronize()
            t_order = time.perf_counter() - t0
            rp, ci, va, vomp = reorder.apply_rank_device(rowptr, col, val, rank)
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            Bp = B[vomp.long()].contiguous()
            refp = ref[vomp.long()]
            print(f"  {oname}: ordering {t_order * 1e3:.1f} ms (+ CSR rewrite: {t_all * 1e3:.1f} ms) {stats}", flush=True)
            res["plans"][oname + " seconds"] = round(t_order, 4)
            configs = [("auto", dict())] + [(f"slices={s}", dict(slices=int(s))) for s in args.slices.split(",") if s != ""] + \
                      [("panels auto", dict(panels="auto", slices=0))] + \
                      [(f"unsliced, tile={t}", dict(slices=0, _tile=int(t))) for t in args.tiles.split(",") if t != ""]
            for cname, kw in configs:
                try:
                    kw = dict(kw)
                    tile = kw.pop("_tile", 0)
                    adj = gcn_amd.CsrAdjacency(rp, ci, va, (n, n), symmetric=True, **kw)
                    if tile:
                        adj.set_tile_cols(tile)
                    t = timed(lambda: adj.matmul_raw(Bp, out=out), args.iters)
                except gcn_amd.GcnAmdError as e:
                    print(f"    {cname}: refused ({e})", flush=True)
                    continue
                err = float((out - refp).abs().max() / refp.abs().max())
                extra = f" panels R={adj.panel_rows} coverage={adj.panel_coverage:.3f} dense={adj.dense_panels}" if "panels" in kw else ""
                print(f"    {oname} / {cname}: {t:.3f} ms  slices={adj.num_slices}  {adj.main_kernel(k)}  err {err:.1e}{extra}", flush=True)
                assert err <= 1e-5, err
                res["plans"][f"{oname} / {cname}"] = {"ms": round(t, 4), "slices": adj.num_slices, "kernel": adj.main_kernel(k),
                                                       "panel_coverage": round(adj.panel_coverage, 4) if "panels" in kw else None,
                                                       "dense_panels": adj.dense_panels if "panels" in kw else None}
                del adj
            del rp, ci, va, vomp, Bp, refp
        print(json.dumps(res), flush=True)
        del rowptr, col, val, comm, B, out, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
